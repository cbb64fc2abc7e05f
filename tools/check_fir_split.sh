# after instantiating the FIR stencils apart: operator tests, then the headline (20-step) and adjoint bench lines
mkdir -p gpurun_out/r03
timeout 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_grad.py -q -m gpu -x -k "group_norm or fir or gn_ or resampl or backward_pieces or vjp" > gpurun_out/r03/fir_split_tests.log 2>&1; tail -2 gpurun_out/r03/fir_split_tests.log
timeout 200 python bench.py --t 20 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('headline t20 img/s', round(d['value'],3), 'conv TF', round(r['achieved'],1), 'conv share', round(r['time_share_of_step'],4), 'sclk', r['sclk_mhz']['median'])" > gpurun_out/r03/fir_split_bench.log 2>&1
timeout 200 python bench.py --workload cifar32_ncsnpp_adjoint --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('adjoint img/s', round(d['value'],3))" >> gpurun_out/r03/fir_split_bench.log 2>&1
timeout 200 python bench.py --workload cifar32_ncsnpp --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cifar img/s', round(d['value'],3))" >> gpurun_out/r03/fir_split_bench.log 2>&1
cat gpurun_out/r03/fir_split_bench.log
