#!/bin/bash
# metafast.sh -- launcher of the MI355X hot path with the reference's command line (src/stub.sh:1-44):
# strips the JVM-only options (-m/--memory <X>, -ea, -X*, -agentlib:*) and runs the native driver.
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
args=()
while [ $# -gt 0 ]; do
  case "$1" in
    -m|--memory) shift; shift;;
    -ea|-X*|-agentlib:*) shift;;
    *) args+=("$1"); shift;;
  esac
done
exec "$HERE/metafast_amd/cli/metafast" "${args[@]}"
