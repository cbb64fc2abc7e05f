#!/bin/bash
# One-shot development check on the GPU box (gpurun): new-kernel tests, ablation probes, A/B bench lines.  Writes gpurun_out/dev_*.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
S=$(date +%s)
lap() { echo "[$(( $(date +%s) - S )) s] $1" >> gpurun_out/dev_timeline.log; }
: > gpurun_out/dev_timeline.log
timeout 400 python -m pytest tests/test_gpu_ops.py -q -k "fp16_weights_single_pass or fp16_output or f16in or operand_output" > gpurun_out/dev_tests.log 2>&1
echo "rc=$?" >> gpurun_out/dev_tests.log; lap tests
timeout 200 python tests/probes/pp_ablate.py --batch 64 --w16 > gpurun_out/dev_ablate.log 2>&1; lap ablate
for epi in 0 1; do
  DP_H2_SW_EPI=$epi timeout 150 python bench.py --batch 32 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/dev_bench_b32_epi$epi.json 2> gpurun_out/dev_bench_b32_epi$epi.err; lap bench_b32_epi$epi
done
DP_H2_SW_EPI=1 timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/dev_bench_b64_epi1.json 2> gpurun_out/dev_bench_b64_epi1.err; lap bench_b64_epi1
timeout 300 python -m pytest tests/test_gpu_loops.py -x -q -s -k "f16sr and (guided_loop or ncsnpp_loop)" > gpurun_out/dev_loops.log 2>&1
echo "rc=$?" >> gpurun_out/dev_loops.log; lap loops
tail -30 gpurun_out/dev_tests.log | cut -c1-400; cat gpurun_out/dev_ablate.log; grep -a "max-abs\|passed\|failed" gpurun_out/dev_loops.log
python - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/dev_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, "images/s", round(d["value"], 3), "conv TF", r.get("achieved") and round(r["achieved"], 1), "share", r.get("time_share_of_step") and round(r["time_share_of_step"], 3), "sclk", (r.get("sclk_mhz") or {}).get("median"))
    except Exception as e:
        print(f, "unreadable", e)
P
cat gpurun_out/dev_timeline.log
