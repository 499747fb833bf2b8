# The default dynamics through its ahead-of-time tape (0) and through the same tape compiled at run time (2).
cd /root/repo
for j in 0 2; do
  echo "== CASCADE_WAVE_JIT=$j"
  export CASCADE_WAVE_JIT=$j CASCADE_WAVE_DEBUG=1
  t0=$(date +%s%N)
  timeout -k 10 300 python -u profiles/bench_configs.py leo_default 19647 30 5 > gpurun_out/jit_cmp_$j.log 2>&1 || exit 1
  t1=$(date +%s%N)
  grep -vE "ng [0-9]|tape:" gpurun_out/jit_cmp_$j.log
  echo "wall $(( (t1 - t0) / 1000000 )) ms"
done
