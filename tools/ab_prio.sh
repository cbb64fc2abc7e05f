# A/B of the static wave priority in the 8-wave convolution kernel (DP_H2_DW_PRIO), after the variant bit-identity tests
mkdir -p gpurun_out/r03
timeout 200 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "fp16_weights_single_pass" > gpurun_out/r03/prio_tests.log 2>&1; tail -1 gpurun_out/r03/prio_tests.log
for i in 1 2; do for u in 0 1; do
  DP_H2_DW_PRIO=$u timeout 100 python bench.py --t 20 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('prio=$u', 'img/s(t20)', round(d['value'],3), 'conv TF', round(r['achieved'],1), 'sclk', r['sclk_mhz']['median'])"
done; done > gpurun_out/r03/prio_ab.log 2>&1
cat gpurun_out/r03/prio_ab.log
