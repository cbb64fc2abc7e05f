cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r04
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}
print('$1', 'img/s', round(d['value'],3), 'conv TF', r.get('achieved') and round(r['achieved'],1), 'share', r.get('time_share_of_step') and round(r['time_share_of_step'],3), 'others', r.get('other_kernels_share_of_step'), 'sclk', (r.get('sclk_mhz') or {}).get('median'))"; }
for v in 2 1 2 1; do
  DP_H2_PP=$v timeout 200 python bench.py --workload cifar32_ncsnpp_adjoint --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | line "adjoint b128 DP_H2_PP=$v"
done > gpurun_out/r04/adjoint_pp_fill_rule_ab.log 2>&1
cat gpurun_out/r04/adjoint_pp_fill_rule_ab.log
