#!/bin/bash
# PMC passes over a short bench run (C = 1), to see what bounds k_bp_tiles / k_np_prefilter / k_bp_dual
# (run on the GPU box: gpurun -- 'bash profiles/pmc_tiles.sh r02d'); the summary is printed and
# written to gpurun_out/<tag>_pmc_tiles.txt (committed as profiles/<tag>_pmc_tiles.txt).
TAG=${1:-r02d}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_tiles; mkdir -p $OUT
ARGS="bench.py --n-par-ct 1 --steps 2 --warmup 2 --no-cpu-baseline --no-host-sync-leg"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/sq -o p -- python3 $ARGS > $OUT/sq.json 2> $OUT/sq.log || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d $OUT/sq2 -o p -- python3 $ARGS > $OUT/sq2.json 2> $OUT/sq2.log || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $OUT/tc -o p -- python3 $ARGS > $OUT/tc.json 2> $OUT/tc.log || exit 1
python3 - $TAG <<'PY' | tee gpurun_out/${TAG}_pmc_tiles.txt
import csv,glob,collections,sys,json
print("# rocprofv3 --pmc over `python bench.py --n-par-ct 1 --steps 2 --warmup 2` (1M objects, C = 1); sums over the listed number of dispatches")
b=json.loads(open('gpurun_out/pmc_tiles/sq.json').read().strip().splitlines()[-1])
print("# units of one superstep:", json.dumps(b['roofline']['units']))
for sub in ('sq','sq2','tc'):
    f=glob.glob('gpurun_out/pmc_tiles/%s/**/p_counter_collection.csv'%sub,recursive=True)[0]
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    seen=set()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].split('::')[-1][:28]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
        if (r['Dispatch_Id']) not in seen: seen.add(r['Dispatch_Id']); cnt[k]+=1
    for kk in sorted(agg):
        if kk.startswith(('k_bp_tiles','k_np_prefilter','k_bp_dual')):
            print(sub,kk,"dispatches",cnt[kk],{c:round(v) for c,v in sorted(agg[kk].items())})
PY
