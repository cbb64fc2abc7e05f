#!/bin/bash
# One-shot development check on the GPU box (gpurun): new-kernel tests, ablation probes, A/B bench lines.  Writes gpurun_out/dev_*.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
S=$(date +%s)
lap() { echo "[$(( $(date +%s) - S )) s] $1" >> gpurun_out/dev_timeline.log; }
: > gpurun_out/dev_timeline.log
timeout 400 python -m pytest tests/test_gpu_ops.py -x -q -k "fp16_weights_single_pass or fp16_output or f16in or operand_output" > gpurun_out/dev_tests.log 2>&1
echo "rc=$?" >> gpurun_out/dev_tests.log; lap tests
timeout 200 python tests/probes/pp_ablate.py --batch 64 --w16 > gpurun_out/dev_ablate.log 2>&1; lap ablate
timeout 90 python tests/probes/gn_bench.py > gpurun_out/dev_gn.log 2>&1; lap gn
DIFFPURE_LEAN=0 DP_H2_SX=0 timeout 150 python bench.py --batch 32 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/dev_bench_b32_base.json 2> gpurun_out/dev_bench_b32_base.err; lap bench_base
DIFFPURE_LEAN=1 DP_H2_SX=0 timeout 150 python bench.py --batch 32 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/dev_bench_b32_lean.json 2> gpurun_out/dev_bench_b32_lean.err; lap bench_lean
DIFFPURE_LEAN=1 DP_H2_SX=1 timeout 150 python bench.py --batch 32 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/dev_bench_b32_lean_sx.json 2> gpurun_out/dev_bench_b32_lean_sx.err; lap bench_lean_sx
DIFFPURE_LEAN=1 DP_H2_SX=1 timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/dev_bench_b64_lean_sx.json 2> gpurun_out/dev_bench_b64_lean_sx.err; lap bench_b64
DP_H2_SX=1 timeout 300 python -m pytest tests/test_gpu_loops.py -x -q -s -k "f16sr and (guided_loop or ncsnpp_loop or forward_whole or adjoint_ode)" > gpurun_out/dev_loops.log 2>&1
echo "rc=$?" >> gpurun_out/dev_loops.log; lap loops
tail -3 gpurun_out/dev_tests.log; cat gpurun_out/dev_ablate.log gpurun_out/dev_gn.log; tail -5 gpurun_out/dev_loops.log
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
