#!/bin/bash
# PMC passes over a short bench run, to see what bounds k_bp_dual (run on the GPU box).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_bp; mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/sq -o p -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-host-sync-leg > $OUT/sq.json 2> $OUT/sq.log || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $OUT/tc -o p -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-host-sync-leg > $OUT/tc.json 2> $OUT/tc.log || exit 1
python3 - <<'PY'
import csv,glob,collections
for sub in ('sq','tc'):
    f=glob.glob('gpurun_out/pmc_bp/%s/**/p_counter_collection.csv'%sub,recursive=True)[0]
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    seen=set()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].split('::')[-1][:28]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
        if (r['Dispatch_Id']) not in seen: seen.add(r['Dispatch_Id']); cnt[k]+=1
    for k in ('k_bp_dual','k_np_prefilter_flat<true>','k_rs_scatter','k_bvh_refit','k_propagate<20, false, false, 0>'):
        for kk in agg:
            if kk.startswith(k.split('<')[0]) and (k in kk or '<' not in k):
                print(sub,kk,cnt[kk],{c:round(v) for c,v in agg[kk].items()})
PY
