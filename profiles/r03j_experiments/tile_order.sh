# Experiment: traversal outputs stay in their shard from level S on (CASCADE_BP_SPREAD), tile runs of R batches
# per workgroup (CASCADE_BP_TILE_RUN): per-kernel times of the bench superstep.
cd /root/repo
for cfg in "1000 0" "1000 64" "16 0" "16 64" "14 64" "18 64" "16 16" "16 256" "12 64"; do
  set -- $cfg
  export CASCADE_BP_SPREAD=$1 CASCADE_BP_TILE_RUN=$2
  timeout -k 10 200 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-host-sync-leg --no-ct-sweep > gpurun_out/exp_to.json 2> gpurun_out/exp_to.err || { tail -5 gpurun_out/exp_to.err; exit 1; }
  python - "$1" "$2" <<'PY'
import json, sys
d = json.load(open("gpurun_out/exp_to.json"))
k = d["kernel_ms_per_step"]
print("spread", sys.argv[1], "run", sys.argv[2], "ms/step", d["ms_per_step"], "tiles", k.get("k_bp_tiles"), "dual", k.get("k_bp_dual"), "pairs", d["config"]["bp_pairs_per_step_all_ranks"], flush=True)
PY
done
