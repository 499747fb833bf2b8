#!/bin/bash
# PMC passes over a short bench run of bench.py's own configuration (1M objects, C = 8 unless given):
#   gpurun -- 'bash profiles/pmc_step.sh r03a [n_par_ct]'
# Writes gpurun_out/<tag>_pmc_step.txt (per-kernel counter sums) and gpurun_out/<tag>_valu.json (VALU
# wave-instructions per launch of every kernel, keyed by the run's workload string: what bench.py's
# roofline.valu_issue quotes). Copy both to profiles/.
TAG=${1:-r03a}; CC=${2:-8}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_step; mkdir -p $OUT
ARGS="bench.py --n-par-ct $CC --steps 2 --warmup 2 --no-cpu-baseline --no-host-sync-leg --no-ct-sweep"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/sq -o p -- python3 $ARGS > $OUT/sq.json 2> $OUT/sq.log || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d $OUT/sq2 -o p -- python3 $ARGS > $OUT/sq2.json 2> $OUT/sq2.log || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $OUT/tc -o p -- python3 $ARGS > $OUT/tc.json 2> $OUT/tc.log || exit 1
python3 - $TAG $CC <<'PY' | tee gpurun_out/${TAG}_pmc_step.txt
import csv,glob,collections,sys,json
tag, cc = sys.argv[1], sys.argv[2]
b=json.loads(open('gpurun_out/pmc_step/sq.json').read().strip().splitlines()[-1])
print("# rocprofv3 --pmc over `python bench.py --n-par-ct %s --steps 2 --warmup 2` (%s); sums over the listed number of dispatches" % (cc, b['config']['workload']))
print("# units of one superstep:", json.dumps(b['roofline']['units']))
valu = {}
for sub in ('sq','sq2','tc'):
    f=glob.glob('gpurun_out/pmc_step/%s/**/p_counter_collection.csv'%sub,recursive=True)[0]
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    seen=set()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].split('::')[-1].split('<')[0][:28]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
        if (r['Dispatch_Id']) not in seen: seen.add(r['Dispatch_Id']); cnt[k]+=1
    for kk in sorted(agg):
        if kk.startswith(('k_bp_tiles','k_np_prefilter','k_bp_dual','k_propagate','k_chunk_boxes','k_bvh','k_rs_','k_gather','k_np_')):
            print(sub,kk,"dispatches",cnt[kk],{c:round(v) for c,v in sorted(agg[kk].items())})
            if sub == 'sq':
                valu[kk] = {"valu_per_launch": agg[kk]['SQ_INSTS_VALU'] / cnt[kk], "salu_per_launch": agg[kk]['SQ_INSTS_SALU'] / cnt[kk], "lds_per_launch": agg[kk]['SQ_INSTS_LDS'] / cnt[kk], "dispatches": cnt[kk]}
json.dump({"workload": b['config']['workload'], "source": "profiles/pmc_step.sh: SQ_INSTS_* summed over the dispatches of a 2 + 2 + 3-superstep run, per dispatch", "kernels": valu}, open('gpurun_out/%s_valu.json' % tag, 'w'), indent=1)
PY
