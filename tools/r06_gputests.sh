#!/bin/bash
# GPU box: the whole -m gpu suite in ONE process, then smoke()
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/${1:-r06_tests}; mkdir -p $out
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; rc=$?; tail -6 $out/tests.log
[ $rc -ne 0 ] && exit $rc
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
