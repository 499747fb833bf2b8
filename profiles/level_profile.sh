cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lvl -o t -- python3 bench.py --n-par-ct 1 --steps 2 --warmup 2 --no-cpu-baseline --no-host-sync-leg > gpurun_out/lvl/bench.json 2> gpurun_out/lvl/err.log
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/lvl/**/t_kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last traversal: find last k_bp_dual_init
idx=[i for i,r in enumerate(rows) if 'k_bp_dual_init' in r['Kernel_Name']][-1]
t0=int(rows[idx]['Start_Timestamp'])
out=[]
for r in rows[idx:]:
    if 'k_bp_dual' not in r['Kernel_Name']: 
        if out and 'k_bp_dual(' not in r['Kernel_Name'] and len(out)>3: break
        continue
    out.append(((int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
print("start_us dur_us per dispatch of the last traversal:")
print(" ".join("%.0f:%.1f"%x for x in out))
print("sum dur", sum(d for _,d in out), "span", out[-1][0]+out[-1][1])
# whole last step timeline: names and durations
names={}
for r in rows[idx-60:idx+80]:
    pass
PY
