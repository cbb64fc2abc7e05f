#!/bin/bash
# Development check on the GPU box (gpurun --timeout 900 -- 'bash tools/dev_check_r04.sh [ops|all]'): operator tests first (fail fast),
# then the whole -m gpu suite, the epilogue probe, and short bench lines with the fp16 residual stream on / off.  Writes gpurun_out/r04/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
S=$(date +%s)
lap() { echo "[$(( $(date +%s) - S )) s] $1" >> $O/dev_timeline.log; }
: > $O/dev_timeline.log
timeout 400 python -m pytest tests/test_gpu_ops.py -m gpu -x -q > $O/dev_ops.log 2>&1; echo "rc=$?" >> $O/dev_ops.log; lap ops
tail -4 $O/dev_ops.log | cut -c1-300
if [ "${1:-all}" = "ops" ]; then exit 0; fi
timeout 700 python -m pytest tests -m gpu -q -s --deselect tests/test_gpu_ops.py > $O/dev_tests.log 2>&1; echo "rc=$?" >> $O/dev_tests.log; lap tests
grep -E "passed|failed|error" $O/dev_tests.log | tail -3 | cut -c1-300
grep -E "^FAILED|^ERROR" $O/dev_tests.log | head -20 | cut -c1-250
grep -hE "max-abs|rel\. error|K-segment" $O/dev_tests.log $O/dev_ops.log | cut -c1-220 | head -60
timeout 200 python tests/probes/conv_epi_probe.py > $O/dev_epi_probe.log 2>&1; lap probe
cat $O/dev_epi_probe.log | cut -c1-250
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}
print('$1', 'img/s', round(d['value'],3), 'conv TF', r.get('achieved') and round(r['achieved'],1), 'share', r.get('time_share_of_step') and round(r['time_share_of_step'],3), 'sclk', (r.get('sclk_mhz') or {}).get('median'))"; }
for l in 1 0 1; do
  DIFFPURE_LEAN16=$l timeout 200 python bench.py --t 20 --steps 1 --warmup 1 --no-cpu-baseline 2>$O/dev_bench.err | line "headline(t20) lean16=$l"
done > $O/dev_bench_ab.log 2>&1; lap bench_ab
for l in 1 0; do
  DIFFPURE_LEAN16=$l timeout 200 python bench.py --workload cifar32_ncsnpp --steps 2 --warmup 1 --no-cpu-baseline 2>>$O/dev_bench.err | line "cifar b256 lean16=$l"
  DIFFPURE_LEAN16=$l timeout 200 python bench.py --workload cifar32_ncsnpp_adjoint --steps 1 --warmup 1 --no-cpu-baseline 2>>$O/dev_bench.err | line "adjoint b128 lean16=$l"
done >> $O/dev_bench_ab.log 2>&1; lap bench_cifar
cat $O/dev_bench_ab.log; tail -3 $O/dev_bench.err | cut -c1-300
cat $O/dev_timeline.log
