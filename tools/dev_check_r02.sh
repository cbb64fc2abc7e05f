#!/bin/bash
# Development check on the GPU box (gpurun --timeout 600 -- 'bash tools/dev_check_r02.sh'): the whole -m gpu suite, smoke, the
# convolution / GroupNorm / head probes and one default bench step.  Writes gpurun_out/dev_*.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
S=$(date +%s)
lap() { echo "[$(( $(date +%s) - S )) s] $1" >> gpurun_out/dev_timeline.log; }
: > gpurun_out/dev_timeline.log
timeout 500 python -m pytest tests -m gpu -x -q > gpurun_out/dev_tests.log 2>&1; echo "rc=$?" >> gpurun_out/dev_tests.log; lap tests
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/dev_smoke.log 2>&1; echo "rc=$?" >> gpurun_out/dev_smoke.log; lap smoke
timeout 200 python tests/probes/pp_ablate.py --batch 64 --w16 > gpurun_out/dev_ablate.log 2>&1; lap ablate
timeout 90 python tests/probes/gn_bench.py > gpurun_out/dev_gn.log 2>&1; lap gn
timeout 90 python tests/probes/head_conv.py > gpurun_out/dev_head.log 2>&1; lap head
timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/dev_bench_b64.json 2> gpurun_out/dev_bench_b64.err; lap bench
tail -3 gpurun_out/dev_tests.log | cut -c1-300; tail -2 gpurun_out/dev_smoke.log; cat gpurun_out/dev_ablate.log gpurun_out/dev_gn.log gpurun_out/dev_head.log
python - <<'P'
import json
try:
    d = json.loads(open("gpurun_out/dev_bench_b64.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("images/s", round(d["value"], 3), "conv TFLOP/s", r.get("achieved") and round(r["achieved"], 1), "share", r.get("time_share_of_step") and round(r["time_share_of_step"], 3),
          "sclk", (r.get("sclk_mhz") or {}).get("median"))
except Exception as e:
    print("bench line unreadable:", e)
P
cat gpurun_out/dev_timeline.log
