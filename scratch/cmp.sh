set -e
for w in 4 16 64 256; do
  cp cascade_amd/libcascade_hip_x$w.so cascade_amd/libcascade_hip.so
  echo "== run $w"
  timeout -k 10 200 python -u bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/b_x$w.json 2> gpurun_out/b_x$w.err
  python3 -c "
import json;d=json.load(open('gpurun_out/b_x$w.json'));print(d['value'],d['ms_per_step']);print({k:v for k,v in list(d['kernel_ms_per_step'].items())[:4]})"
done
