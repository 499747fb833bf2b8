set -e
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fixtures.py tests/test_gpu_sim.py tests/test_distributed.py -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/tile_tests.log 2>&1 || { tail -30 gpurun_out/tile_tests.log; exit 1; }
tail -3 gpurun_out/tile_tests.log
for T in 8; do
  for C in 1 8; do
    CASCADE_DT_TILE=$T timeout -k 10 200 python bench.py --n-par-ct $C --steps 10 --no-cpu-baseline --no-host-sync-leg > gpurun_out/tile_T${T}_C${C}.json 2> gpurun_out/tile_T${T}_C${C}.err
    python - <<PY
import json
d=json.loads(open('gpurun_out/tile_T${T}_C${C}.json').read().strip().splitlines()[-1])
k=d['kernel_ms_per_step']
print('T=$T C=$C', d['value'], d['ms_per_step'], {n:k.get(n) for n in ('k_bp_dual','k_bp_tiles','k_np_prefilter','k_bvh_pack')}, d['config']['bp_pairs_per_step_all_ranks'])
PY
  done
done
