mkdir -p gpurun_out/r03
timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "fp16_weights_single_pass or h1_fp16" > gpurun_out/r03/unroll_tests.log 2>&1; tail -3 gpurun_out/r03/unroll_tests.log
for i in 1 2; do
for u in 0 1; do
  DP_H2_DW_UNROLL=$u timeout 200 python bench.py --t 20 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('unroll=$u', 'img/s(t20)', round(d['value'],3), 'conv TF', round(r['achieved'],1), 'sclk', r['sclk_mhz']['median'])" 
done; done > gpurun_out/r03/unroll_ab.log 2>&1
cat gpurun_out/r03/unroll_ab.log
