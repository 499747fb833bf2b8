cd /root/repo
timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 400 --timeout-method thread > gpurun_out/final_gpu_suite.log 2>&1 || { tail -30 gpurun_out/final_gpu_suite.log; exit 1; }
tail -3 gpurun_out/final_gpu_suite.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final_smoke.log 2>&1 || { tail -20 gpurun_out/final_smoke.log; exit 1; }
tail -2 gpurun_out/final_smoke.log
timeout -k 10 400 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err || { tail -20 gpurun_out/final_bench.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/final_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic_source'], d['roofline']['valu_issue']['source'], d['cpu_baseline']['value'])"
