#!/bin/bash
# Development check of the packed-arithmetic epilogue and the one-pass GroupNorm backward on the GPU box
# (gpurun --timeout 900 -- 'bash tools/dev_epi_r04.sh'): operator tests, the probes, short bench lines.  Writes gpurun_out/r04/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
S=$(date +%s)
lap() { echo "[$(( $(date +%s) - S )) s] $1" >> $O/epi_timeline.log; }
: > $O/epi_timeline.log
timeout 500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_grad.py -m gpu -x -q > $O/epi_ops.log 2>&1; echo "rc=$?" >> $O/epi_ops.log; lap ops
tail -6 $O/epi_ops.log | cut -c1-400
timeout 200 python tests/probes/conv_epi_probe.py > $O/epi_probe.log 2>&1; lap probe
cat $O/epi_probe.log | cut -c1-250
timeout 200 python tests/probes/dw8_timeline.py > $O/epi_dw8_timeline.log 2>&1; lap timeline
tail -12 $O/epi_dw8_timeline.log | cut -c1-420
timeout 200 python tests/probes/gn_bwd_one_pass_probe.py > $O/gn_bwd_one_pass_probe.log 2>&1; lap gnbwd
cat $O/gn_bwd_one_pass_probe.log | cut -c1-250
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}
print('$1', 'img/s', round(d['value'],3), 'conv TF', r.get('achieved') and round(r['achieved'],1), 'share', r.get('time_share_of_step') and round(r['time_share_of_step'],3), 'sclk', (r.get('sclk_mhz') or {}).get('median'), 'held', r.get('frac_at_held_clock'))"; }
{
timeout 200 python bench.py --t 20 --steps 1 --warmup 1 --no-cpu-baseline 2>$O/epi_bench.err | line "headline(t20)"
timeout 200 python bench.py --workload cifar32_ncsnpp --steps 2 --warmup 1 --no-cpu-baseline 2>>$O/epi_bench.err | line "cifar b256"
timeout 200 python bench.py --workload cifar32_ncsnpp_adjoint --steps 1 --warmup 1 --no-cpu-baseline 2>>$O/epi_bench.err | line "adjoint b128"
} > $O/epi_bench.log 2>&1; lap bench
cat $O/epi_bench.log; tail -3 $O/epi_bench.err | cut -c1-300
if [ "${1:-}" = "loops" ]; then
timeout 300 python -m pytest tests/test_gpu_loops.py tests/test_gpu_models.py -m gpu -x -q > $O/epi_loops.log 2>&1; echo "rc=$?" >> $O/epi_loops.log; lap loops
tail -5 $O/epi_loops.log | cut -c1-300
fi
cat $O/epi_timeline.log
