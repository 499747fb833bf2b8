#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root:
#   gpurun --timeout 900 -- 'bash profiles/collect.sh r01c'
# 1. rocprofv3 --kernel-trace --stats of the default bench command (per-kernel durations);
# 2. two PMC passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass) over a short run.
# Outputs land under gpurun_out/<tag>/; profiles/summarise.py turns them into the committed
# profiles/<tag>_kernel_stats.csv and profiles/<tag>_traffic.json.
set -e
TAG=${1:-r01}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# (without the default-mode and C = 1 legs and the CPU check, so that every launch in the table belongs to the timed configuration)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 bench.py --no-cpu-baseline --no-host-sync-leg --no-ct-sweep > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
echo "stats pass done"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o f -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-host-sync-leg --no-ct-sweep > $OUT/fetch.json 2> $OUT/fetch.log
echo "fetch pass done"
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o w -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-host-sync-leg --no-ct-sweep > $OUT/write.json 2> $OUT/write.log
echo "write pass done"
timeout -k 10 300 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "plain bench done"
find $OUT -name '*.csv' | head -20
bash profiles/pmc_step.sh $TAG 8 > $OUT/pmc_step.log 2>&1 && echo "pmc step done"
