#!/bin/bash
# bash profiles/n_sweep.sh "n1 n2 ..." C : per-pair cost of the broad-phase / pre-filter kernels against the population
# (the records of a small population stay in L2). Run on the GPU box.
NS=$1; C=${2:-8}
for N in $NS; do
    timeout -k 10 200 python bench.py --nparts $N --n-par-ct $C --steps 5 --no-cpu-baseline --no-host-sync-leg > gpurun_out/nsweep_${N}.json 2> gpurun_out/nsweep_${N}.err || exit 1
    python - <<PY
import json
d=json.loads(open('gpurun_out/nsweep_${N}.json').read().strip().splitlines()[-1])
k=d['kernel_ms_per_step']; u=d['roofline']['units']
print('n=$N', d['ms_per_step'], 'pairs', u['pairs'], 'tiles', u['bp_tiles'], 'node_pairs', u['bp_node_pairs'],
      'ps/pair prefilter %.2f' % (k['k_np_prefilter']*1e9/max(u['pairs'],1)), 'ps/tile tiles %.1f' % (k['k_bp_tiles']*1e9/max(u['bp_tiles'],1)),
      'ps/node_pair dual %.1f' % (k['k_bp_dual']*1e9/max(u['bp_node_pairs'],1)))
PY
done
