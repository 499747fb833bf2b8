#!/bin/bash
# bash profiles/env_sweep.sh VAR "v1 v2 ..." "C1 C2 ..." : bench.py under VAR=v for every v and n_par_ct (run on the GPU box).
VAR=$1; VALS=$2; CS=${3:-"1 8"}
for V in $VALS; do
  for C in $CS; do
    env $VAR=$V timeout -k 10 200 python bench.py --n-par-ct $C --steps 10 --no-cpu-baseline --no-host-sync-leg > gpurun_out/sweep_${VAR}_${V}_C${C}.json 2> gpurun_out/sweep_${VAR}_${V}_C${C}.err || exit 1
    python - <<PY
import json
d=json.loads(open('gpurun_out/sweep_${VAR}_${V}_C${C}.json').read().strip().splitlines()[-1])
k=d['kernel_ms_per_step']
print('$VAR=$V C=$C', d['value'], d['ms_per_step'], {n:v for n,v in list(k.items())[:14]})
PY
  done
done
