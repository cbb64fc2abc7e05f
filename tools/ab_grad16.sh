# A/B of the one-pass fp16 gradient path (DIFFPURE_GRAD16) on the adjoint-ODE bench, after the gradient tests
mkdir -p gpurun_out/r03
timeout 600 python -m pytest tests/test_gpu_grad.py tests/test_gpu_loops.py -q -m gpu -s -k "not guided_loop and not ncsnpp_loop" > gpurun_out/r03/grad16_tests.log 2>&1; tail -2 gpurun_out/r03/grad16_tests.log
for g in 0 1 0 1; do
  DIFFPURE_GRAD16=$g timeout 200 python bench.py --workload cifar32_ncsnpp_adjoint --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('grad16=$g', 'img/s', round(d['value'],3))"
done > gpurun_out/r03/grad16_ab2.log 2>&1
cat gpurun_out/r03/grad16_ab2.log
