#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r3full; mkdir -p $OUT
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1; rc=$?
tail -6 $OUT/pytest.txt
[ $rc -ne 0 ] && exit $rc
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
