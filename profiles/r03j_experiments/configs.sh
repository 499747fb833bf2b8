cd /root/repo
mkdir -p gpurun_out/r03j_configs
for w in "shells 2000000 4 4" "shells 10000000 2 3" "leo_j2drag 19647 30 5" "leo_default 19647 30 5" "haumea 10000 2 5" "leo_full 19647 30 3" "leo_j2drag 1000000 8 3"; do
  set -- $w
  timeout -k 10 400 python -u profiles/bench_configs.py $1 $2 $3 $4 > gpurun_out/r03j_configs/$1_$2.json 2> gpurun_out/r03j_configs/$1_$2.err || { tail -5 gpurun_out/r03j_configs/$1_$2.err; exit 1; }
  python - gpurun_out/r03j_configs/$1_$2.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["workload"], d["n"], d["ms_per_step"], list(d["kernel_ms_per_step"].items())[:3], flush=True)
PY
done
