#!/bin/bash
# Compile-side look at the register corruption of round 1: rebuilds
# k_propagate<20, EVENTS> with the terminal-event root search as it was before commit df8f200 --
# an OUT-OF-LINE event_first_root<20> calling the register-resident isolate_and_refine_n<20> from
# inside the integration loop -- and keeps the ISA (--save-temps) under /tmp/regcor. No GPU needed.
#   bash profiles/regcorr_experiment.sh && less /tmp/regcor/exp-hip-amdgcn-amd-amdhsa-gfx950.s
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
rm -rf /tmp/regcor && mkdir -p /tmp/regcor && cd /tmp/regcor
cp -r $ROOT/cascade_amd/csrc src && cp $ROOT/include/cascade_hip.h src/
python3 - <<'PY'
p = "/tmp/regcor/src/k_propagate.hpp"
s = open(p).read()
s = s.replace("template <int P>\nCSC_DEV double event_first_root(", "template <int P>\nCSC_DEV_NOINLINE double event_first_root(")
s = s.replace("    int n = isolate_and_refine(g, P, h, direction, roots, P + 1, warn);", "    int n = isolate_and_refine_n<P>(g, h, direction, roots, P + 1, warn);")
open(p, "w").write(s)
open("/tmp/regcor/exp.hip", "w").write('#include <hip/hip_runtime.h>\n#include "k_propagate.hpp"\ntemplate __global__ void csc::k_propagate<20, true, false, 0>(const csc::prop_params);\n')
PY
hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -Isrc -c exp.hip --save-temps -o exp.o
S=exp-hip-amdgcn-amd-amdhsa-gfx950.s
grep -n "sgpr_spill_count\|vgpr_spill_count\|\.vgpr_count\|private_segment_fixed_size:" $S
echo "call sites:"; grep -n "s_swappc_b64" $S
echo "SGPR-spill VGPRs:"; grep -n "SGPR spill to VGPR lane" $S
