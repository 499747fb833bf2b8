#!/bin/bash
# Instruction-cache counters over k_propagate_wave (default dynamics, 19 647 x 30):
#   gpurun -- 'bash profiles/pmc_wave_ic.sh'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_wave_ic; mkdir -p $OUT
ARGS="profiles/bench_configs.py leo_default 19647 30 3"
timeout -s KILL 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/a -o p -- python3 $ARGS > $OUT/a.json 2> $OUT/a.log || { tail -5 $OUT/a.log; exit 1; }
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/pmc_wave_ic/a/**/p_counter_collection.csv',recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0].split('::')[-1][:28]
    agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
for kk in agg:
    if kk.startswith('k_propagate_wave'): print(kk,{c:round(v) for c,v in sorted(agg[kk].items())})
PY
