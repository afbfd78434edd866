cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
echo "--- direct"; timeout -k 5 120 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "--- build+smoke in one process"; timeout -k 5 300 python3 __graft_entry__.py smoke 2>&1 | tail -6
