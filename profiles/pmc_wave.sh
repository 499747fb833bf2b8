#!/bin/bash
# PMC passes over k_propagate_wave on the catalogue-shaped J2 + drag workload (19 647 objects, 30 chunks):
#   gpurun -- 'bash profiles/pmc_wave.sh r03a [workload] [n] [C]'; summary in gpurun_out/<tag>_pmc_wave.txt
TAG=${1:-r03a}; WL=${2:-leo_j2drag}; N=${3:-19647}; CC=${4:-30}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_wave; mkdir -p $OUT
ARGS="profiles/bench_configs.py $WL $N $CC 3"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/sq -o p -- python3 $ARGS > $OUT/sq.json 2> $OUT/sq.log || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/sq2 -o p -- python3 $ARGS > $OUT/sq2.json 2> $OUT/sq2.log || exit 1
python3 - $TAG "$WL $N $CC" <<'PY' | tee gpurun_out/${TAG}_pmc_wave.txt
import csv,glob,collections,sys,json
print("# rocprofv3 --pmc over `python profiles/bench_configs.py %s 3`; sums over the listed number of dispatches" % sys.argv[2])
print("#", open('gpurun_out/pmc_wave/sq.json').read().strip().splitlines()[-1][:600])
for sub in ('sq','sq2'):
    f=glob.glob('gpurun_out/pmc_wave/%s/**/p_counter_collection.csv'%sub,recursive=True)[0]
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    seen=set()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].split('::')[-1][:28]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
        if (r['Dispatch_Id']) not in seen: seen.add(r['Dispatch_Id']); cnt[k]+=1
    for kk in sorted(agg):
        if kk.startswith(('k_propagate','k_chunk_boxes')):
            print(sub,kk,"dispatches",cnt[kk],{c:round(v) for c,v in sorted(agg[kk].items())})
PY
