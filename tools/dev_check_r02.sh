#!/bin/bash
# Short development check on the GPU box (gpurun).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
S=$(date +%s)
lap() { echo "[$(( $(date +%s) - S )) s] $1" >> gpurun_out/dev_timeline.log; }
: > gpurun_out/dev_timeline.log
timeout 200 python -m pytest tests/test_gpu_ops.py -q -k "few_output" > gpurun_out/dev_tests.log 2>&1
echo "rc=$?" >> gpurun_out/dev_tests.log; lap tests
timeout 120 python tests/probes/head_conv.py > gpurun_out/dev_head.log 2>&1; lap head
timeout 200 python -m pytest tests/test_gpu_loops.py tests/test_gpu_models.py -x -q -k "(f16sr and guided_loop) or guided_small" > gpurun_out/dev_loops.log 2>&1
echo "rc=$?" >> gpurun_out/dev_loops.log; lap loops
timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/dev_bench_b64.json 2> gpurun_out/dev_bench_b64.err; lap bench
tail -3 gpurun_out/dev_tests.log | cut -c1-300; cat gpurun_out/dev_head.log; tail -3 gpurun_out/dev_loops.log | cut -c1-300
python - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/dev_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, "images/s", round(d["value"], 3), "conv TF", r.get("achieved") and round(r["achieved"], 1), "share", r.get("time_share_of_step") and round(r["time_share_of_step"], 3), "other", r.get("other_kernels_share_of_step"))
    except Exception as e:
        print(f, "unreadable", e)
P
cat gpurun_out/dev_timeline.log
