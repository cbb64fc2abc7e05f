#!/bin/bash
# Whole -m gpu suite + three short bench lines at HEAD (gpurun --timeout 900 -- 'bash tools/dev_full_r04.sh').  Writes gpurun_out/r04/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -s > $O/full_tests.log 2>&1; echo "rc=$?" >> $O/full_tests.log
grep -E "passed|failed" $O/full_tests.log | tail -2; grep -E "^FAILED|^ERROR" $O/full_tests.log | head -10 | cut -c1-250
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}; g=r.get('second_kernel') or {}
print('$1', 'img/s', round(d['value'],3), 'conv TF', r.get('achieved') and round(r['achieved'],1), 'share', r.get('time_share_of_step') and round(r['time_share_of_step'],3), 'sclk', (r.get('sclk_mhz') or {}).get('median'), 'held', r.get('frac_at_held_clock') and round(r['frac_at_held_clock'],3), 'GN GB/s', g.get('achieved') and round(g['achieved']), 'GN share', g.get('time_share_of_step') and round(g['time_share_of_step'],3), 'traffic', r.get('traffic'), 'pmc busy', (r.get('mfma_busy_by_pmc') or {}).get('value'))"; }
{
timeout 200 python bench.py --t 20 --steps 1 --warmup 1 --no-cpu-baseline 2>$O/full_bench.err | line "headline(t20)"
timeout 200 python bench.py --workload cifar32_ncsnpp --steps 2 --warmup 1 --no-cpu-baseline 2>>$O/full_bench.err | line "cifar b256"
timeout 200 python bench.py --workload cifar32_ncsnpp_adjoint --steps 1 --warmup 1 --no-cpu-baseline 2>>$O/full_bench.err | line "adjoint b128"
} > $O/full_bench.log 2>&1
cat $O/full_bench.log; tail -3 $O/full_bench.err | cut -c1-300
