#!/bin/bash
# PMC passes over a short bench run, summarised for the propagation kernels
# (run on the GPU box: gpurun -- 'bash profiles/pmc_prop.sh <tag> <n_par_ct>'); the summary is
# printed and written to gpurun_out/<tag>_pmc_prop.txt (committed as profiles/<tag>_pmc_prop.txt).
TAG=${1:-r02h}
C=${2:-8}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_prop; mkdir -p $OUT
ARGS="bench.py --n-par-ct $C --steps 2 --warmup 2 --no-cpu-baseline --no-host-sync-leg"
run_pass() {
    name=$1; shift
    timeout -s KILL 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o p -- python3 $ARGS > $OUT/$name.json 2> $OUT/$name.log
    rc=$?
    if [ $rc -eq 137 ] || [ $rc -eq 124 ]; then echo "pass $name killed"; exit 1; fi
    if [ $rc -ne 0 ]; then echo "pass $name failed rc=$rc"; tail -3 $OUT/$name.log; fi
}
run_pass sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES
run_pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_FLAT
run_pass f64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64
python3 - $TAG $C <<'PY' | tee gpurun_out/${TAG}_pmc_prop.txt
import csv,glob,collections,sys,json
print("# rocprofv3 --pmc over `python bench.py --n-par-ct %s --steps 2 --warmup 2` (1M objects); sums over the listed number of dispatches" % sys.argv[2])
for sub in ('sq','sq2','f64'):
    fs=glob.glob('gpurun_out/pmc_prop/%s/**/p_counter_collection.csv'%sub,recursive=True)
    if not fs:
        print(sub, "no output"); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    seen=set()
    for r in csv.DictReader(open(fs[0])):
        k=r['Kernel_Name'].split('(')[0].split('::')[-1][:40]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
        if (r['Dispatch_Id']) not in seen: seen.add(r['Dispatch_Id']); cnt[k]+=1
    for kk in sorted(agg):
        if kk.startswith(('k_propagate','k_chunk_boxes','k_np_roots','k_np_fex')):
            print(sub,kk,"dispatches",cnt[kk],{c:round(v) for c,v in sorted(agg[kk].items())})
PY
