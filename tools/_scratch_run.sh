cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r04; mkdir -p $O; export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_gpu_ops.py -m gpu -q > $O/dev_ops.log 2>&1; echo "rc=$?" >> $O/dev_ops.log
grep -E "passed|failed" $O/dev_ops.log | tail -2; grep -E "^FAILED" $O/dev_ops.log | head -20 | cut -c1-200
timeout 400 python -m pytest tests/test_gpu_loops.py -m gpu -q -s -k "batch_256 or batch_128 or batch_64 or guided_loop_100" > $O/dev_bitid.log 2>&1; grep -E "passed|failed|equal to|100-step" $O/dev_bitid.log | cut -c1-220
timeout 300 python tests/probes/dw8_timeline.py > $O/dw8_timeline_final.log 2>&1; grep -E "mode |res16|\[|wave" $O/dw8_timeline_final.log | cut -c1-330
for i in 1 2; do python bench.py --t 20 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}; g=r.get('second_kernel') or {}
print('headline(t20)', 'img/s', round(d['value'],3), 'conv TF', r.get('achieved') and round(r['achieved'],1), 'frac', round(r.get('frac') or 0,3), 'held', round(r.get('frac_at_held_clock') or 0,3), 'share', r.get('time_share_of_step') and round(r['time_share_of_step'],3), 'sclk', (r.get('sclk_mhz') or {}).get('median'), 'GN GB/s', round(g.get('achieved') or 0), 'GN share', round(g.get('time_share_of_step') or 0,3), 'build_s', d.get('engine_build_s_per_rank'))"; done
