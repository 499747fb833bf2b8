#!/bin/bash
# k_propagate_wave after the tape fusion: tests of the tape path, then the default dynamics at
# 19 647 x 30 for the given layouts (groups per object : lanes per operation; 0:0 = the library's choice):
#   gpurun --timeout 900 -- 'bash profiles/wave_sweep.sh "0:0 8:8 8:4 4:8 4:4"'
O=gpurun_out/wave_sweep; mkdir -p $O; rm -f $O/*.json
timeout -k 10 300 python -u -m pytest tests/test_gpu_tape.py -x -q --timeout 120 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for c in ${1:-0:0 8:8 8:4 4:8 4:4}; do
ng=${c%:*}; lpg=${c#*:}
CASCADE_WAVE_DEBUG=1 CASCADE_WAVE_NG=$ng CASCADE_WAVE_LPG=$lpg timeout -k 10 200 python profiles/bench_configs.py leo_default 19647 30 6 > $O/leo_default_ng${ng}_$lpg.json 2> $O/leo_default_ng${ng}_$lpg.err || { tail -5 $O/leo_default_ng${ng}_$lpg.err; exit 1; }
done
timeout -k 10 120 python profiles/bench_configs.py haumea 10000 2 8 > $O/haumea.json 2> $O/haumea.err || exit 1
timeout -k 10 200 python profiles/bench_configs.py leo_full 19647 30 4 > $O/leo_full.json 2> $O/leo_full.err || exit 1
for f in $O/*.json; do echo $f $(tail -n 1 $f | cut -c1-110) $(tail -n 1 $f | grep -o '"k_propagate_wave": [0-9.]*'); done
grep -h "\[wave\] tape" $O/leo_default_ng0_0.err $O/haumea.err $O/leo_full.err | sort -u
