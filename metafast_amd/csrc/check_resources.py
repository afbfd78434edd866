#!/usr/bin/env python3
"""Reads hipcc's stderr of one compile with -Rpass-analysis=kernel-resource-usage: passes the compiler's own messages through, writes
the per-kernel table to <out>, and FAILS the build when one of the neighbour-lookup kernels (mf_nbr.h: waves that hand data to each
other through LDS between wave-level barriers, at a forced occupancy) needs scratch -- a build of k_ut_flags_part that spilled hung
on the GPU (round 4)."""
import os
import re
import sys

NO_SCRATCH = re.compile(r"k_ut_flags_part|k_cc_adjacency_part|k_dcc_adjacency_part")
# the translation units that hold those kernels: the guard must SEE them there (a renamed kernel, or a remark format that changed,
# is a failed check, not a passed one)
EXPECT = {"mf_unitig": ["k_ut_flags_part"], "mf_cc": ["k_cc_adjacency_part", "k_dcc_adjacency_part"]}


def main():
    src, out = sys.argv[1], sys.argv[2]
    lines = open(src, errors="replace").read().splitlines()
    kernels, cur, skip, bad = {}, None, 0, []
    for ln in lines:
        if "[-Rpass-analysis=kernel-resource-usage]" in ln:
            skip = 2
            m = re.search(r"remark:\s+Function Name: (\S+)", ln)
            if m:
                cur = m.group(1)
                kernels[cur] = {}
            else:
                m = re.search(r"remark:\s+([^:]+): (\S+)", ln)
                if m and cur:
                    kernels[cur][m.group(1).strip()] = m.group(2)
            continue
        if skip and (re.match(r"^\s*\d+ \| ", ln) or re.match(r"^\s*\| ", ln)):
            skip -= 1
            continue
        skip = 0
        if re.match(r"^\d+ warnings? generated", ln) or "remarks generated" in ln:
            continue
        print(ln, file=sys.stderr)
    unit = os.path.splitext(os.path.basename(src))[0]
    missing = []
    with open(out, "w") as f:
        for name, r in kernels.items():
            f.write(f"{name} vgprs={r.get('VGPRs')} scratch={r.get('ScratchSize [bytes/lane]')} occupancy={r.get('Occupancy [waves/SIMD]')} lds={r.get('LDS Size [bytes/block]')}\n")
            if NO_SCRATCH.search(name):
                sc = r.get("ScratchSize [bytes/lane]")
                if sc is None:
                    missing.append(f"{name}: no 'ScratchSize [bytes/lane]' remark was parsed")
                elif sc != "0":
                    bad.append((name, sc))
    for want in EXPECT.get(unit, []):
        if not any(want in name for name in kernels):
            missing.append(f"{unit}: no resource record of a kernel named *{want}* (renamed? remark format changed?)")
    for name, sc in bad:
        print(f"error: {name} uses {sc} bytes of scratch per lane (see check_resources.py)", file=sys.stderr)
    for m in missing:
        print(f"error: resource check cannot vouch for the no-scratch kernels -- {m}", file=sys.stderr)
    return 1 if bad or missing else 0


if __name__ == "__main__":
    sys.exit(main())
