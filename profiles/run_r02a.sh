#!/bin/bash
# Round-2 first GPU batch (run ON THE GPU BOX through gpurun from the repo root):
# whole GPU suite, the not-yet-collected no-alloc test, the bench line, the FETCH_SIZE calibration.
set -o pipefail
TAG=${1:-r02a}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests -x -v --timeout 300 --timeout-method thread -m gpu > $OUT/pytest.log 2>&1
rc=$?; tail -5 $OUT/pytest.log; echo "pytest rc=$rc"; [ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python -u -m pytest tests/wip_no_alloc.py -x -v --timeout 100 --timeout-method thread -m gpu > $OUT/noalloc.log 2>&1
rc=$?; tail -5 $OUT/noalloc.log; echo "noalloc rc=$rc"; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
rc=$?; tail -c 3000 $OUT/bench.json; tail -5 $OUT/bench.err; echo "bench rc=$rc"; [ $rc -eq 0 ] || exit $rc
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/calib -o c -- build/calib_gather > $OUT/calib.json 2> $OUT/calib.log
rc=$?; cat $OUT/calib.json; echo "calib rc=$rc"
find $OUT -name '*.csv' | head
