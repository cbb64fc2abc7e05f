#!/bin/bash
# Development check on the GPU box (gpurun).  Writes gpurun_out/dev_*.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
S=$(date +%s)
lap() { echo "[$(( $(date +%s) - S )) s] $1" >> gpurun_out/dev_timeline.log; }
: > gpurun_out/dev_timeline.log
timeout 200 python -m pytest tests/test_gpu_ops.py -q -s -k "finalize_cols or column_sums or group_norm" 2>&1 | grep -a "us per launch\|passed\|failed\|Error" > gpurun_out/dev_tests.log
lap tests
timeout 200 python -m pytest tests/test_gpu_loops.py tests/test_gpu_dist.py -x -q -k "(f16sr and (guided_loop or ncsnpp_loop)) or shard" > gpurun_out/dev_loops.log 2>&1
echo "rc=$?" >> gpurun_out/dev_loops.log; lap loops
timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/dev_bench_b64.json 2> gpurun_out/dev_bench_b64.err; lap bench
cat gpurun_out/dev_tests.log; tail -3 gpurun_out/dev_loops.log | cut -c1-300
python - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/dev_bench_b64.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, "images/s", round(d["value"], 3), "conv TF", r.get("achieved") and round(r["achieved"], 1), "share", r.get("time_share_of_step") and round(r["time_share_of_step"], 3), "sclk", (r.get("sclk_mhz") or {}).get("median"))
    except Exception as e:
        print(f, "unreadable", e)
P
cat gpurun_out/dev_timeline.log
