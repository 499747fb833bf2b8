#!/bin/bash
# Secondary BASELINE configurations on one GPU, run on the GPU box:
#   gpurun --timeout 1100 -- 'bash profiles/run_configs_r03.sh [tag]'
O=gpurun_out/configs_r03${1:+_$1}; mkdir -p $O
timeout -k 10 120 python profiles/bench_configs.py leo_j2drag 19647 30 8 > $O/leo_j2drag.json 2> $O/leo_j2drag.err || exit 1
timeout -k 10 200 python profiles/bench_configs.py leo_default 19647 30 6 > $O/leo_default.json 2> $O/leo_default.err || exit 1
timeout -k 10 200 python profiles/bench_configs.py leo_full 19647 30 4 > $O/leo_full.json 2> $O/leo_full.err || exit 1
timeout -k 10 200 python profiles/bench_configs.py leo_j2drag 1000000 8 4 > $O/leo_j2drag_1m.json 2> $O/leo_j2drag_1m.err || exit 1
timeout -k 10 120 python profiles/bench_configs.py haumea 10000 2 8 > $O/haumea.json 2> $O/haumea.err || exit 1
if [ -z "$NOSHELLS" ]; then
timeout -k 10 200 python profiles/bench_configs.py shells 2000000 4 4 > $O/shells_2m.json 2> $O/shells_2m.err || exit 1
timeout -k 10 300 python profiles/bench_configs.py shells 10000000 2 3 > $O/shells_10m.json 2> $O/shells_10m.err || exit 1
fi
tail -n 2 $O/*.json | cut -c1-900
