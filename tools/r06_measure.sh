#!/bin/bash
# Round-6 measurement calls on the GPU box (gpurun).  Everything lands under gpurun_out/r06/<stage>/ (copied into profiles/r06/ afterwards).
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
STAGE=${1:-stem}
O="$R/gpurun_out/r06/$STAGE"
rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp
ulimit -c 0                      # a faulting kernel must not fill the box's disk with host / GPU core dumps (it took the rest of a call with it)
export HSA_ENABLE_COREDUMP=0 AMD_GPU_COREDUMP=0
S=$(date +%s)
lap() { echo "[$(( $(date +%s) - S )) s] $1" >> "$O/timeline.log"; }
: > "$O/timeline.log"
val() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(d['value'],3), 'images/s', d.get('sclk_mhz_median', d['roofline'].get('sclk_mhz')))"; }
rocstats() {   # rocstats <name> <timeout> <bench args...>: rocprofv3 kernel stats of one bench command
  local name=$1 to=$2; shift 2
  ( cd /tmp && timeout "$to" rocprofv3 --kernel-trace --stats -d "$O/prof_$name" -o run --output-format csv -- python "$R/bench.py" "$@" --no-cpu-baseline --no-resident-call > "$O/bench_${name}_under_rocprof.json" 2> "$O/rocprof_$name.err" )
  find "$O/prof_$name" -name "*kernel_trace.csv" -delete; find "$O/prof_$name" -name "*agent_info.csv" -delete
  f=$(find "$O/prof_$name" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/${name}_kernel_stats.csv"
  lap "rocprof_$name"
}
gputests() {
  timeout 1700 python -m pytest tests -m gpu -q -s > "$O/gpu_tests.log" 2>&1; echo "rc=$?" >> "$O/gpu_tests.log"; lap gpu_tests
  timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.log" 2>&1; echo "rc=$?" >> "$O/smoke.log"; lap smoke
  grep -E "passed|failed" "$O/gpu_tests.log" | tail -2; grep -E "^FAILED|^ERROR" "$O/gpu_tests.log" | head; tail -2 "$O/smoke.log"
}
ab() {   # ab <env var> <label> <bench args...>: same-box A/B of one switch, 0 1 0 1
  local var=$1 label=$2; shift 2
  for V in 0 1 0 1; do
    env $var=$V timeout 400 python bench.py "$@" --no-cpu-baseline --no-resident-call > "$O/bench_${label}_$V.json" 2>> "$O/bench_ab.err"
    val "$O/bench_${label}_$V.json" "$label $var=$V" | tee -a "$O/${label}_ab.log"
  done; lap "ab_$label"
}
abv() {   # abv <env var> <value A> <value B> <label> <bench args...>: same-box A/B of one switch, A B A B
  local var=$1 va=$2 vb=$3 label=$4; shift 4
  for V in $va $vb $va $vb; do
    env $var=$V timeout 400 python bench.py "$@" --no-cpu-baseline --no-resident-call > "$O/bench_${label}_$V.json" 2>> "$O/bench_ab.err"
    val "$O/bench_${label}_$V.json" "$label $var=$V" | tee -a "$O/${label}_ab.log"
  done; lap "ab_$label"
}
case "$STAGE" in
xcdpair)   # XCD-aware workgroup maps of the secondary kernels (one-pass GroupNorm backward, gn_finalize_cols, flash attention)
  timeout 600 python -m pytest tests/test_gpu_grad.py tests/test_gpu_ops.py -m gpu -q -x -k "group_norm or attention or gemm" > "$O/tests.log" 2>&1; echo "rc=$?" >> "$O/tests.log"; lap tests
  grep -E "passed|failed|^FAILED|^E  " "$O/tests.log" | head
  timeout 200 python - > "$O/xcd_probe.log" 2>&1 <<'PY'
import torch, sys
sys.path.insert(0, ".")
from diffpure_amd import ops
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B, H, C, G = 64, 256, 256, 32
cs = torch.randn(B * H * H // 64, 2, C, device="cuda")
x = ops.Act(torch.empty(B, H, H, C, device="cuda", dtype=torch.float16), ops.ColStats(cs, 64, C))
del cs
for v in (0, 1, 0, 1):
    with ops.tuning(DP_XCD_MAP=v):
        print("gn_finalize_cols 256^2 x 256 B=64, XCD map", v, ": %.1f us" % t(lambda: ops.group_norm_stats(x, G, 1e-5)))
del x
for (B, hw, c, heads, layout) in ((64, 32, 512, 8, "legacy"), (64, 16, 1024, 16, "legacy"), (256, 16, 256, 1, "split")):
    qkv = torch.randn(B, hw * hw, 3 * c, device="cuda").half()
    for v in (0, 1, 0, 1):
        with ops.tuning(DP_XCD_MAP=v):
            print(f"attention_fused B={B} T={hw*hw} C={c} heads={heads}, XCD map {v}: %.1f us" % t(lambda: ops.attention_fused(qkv, heads, layout, operand_hw=(hw, hw))))
PY
  cat "$O/xcd_probe.log"
  abv DP_XCD_MAP 0 1 guided_t20_xcd --t 20 --steps 1 --warmup 1 --no-conv-profile
  abv DP_XCD_MAP 0 1 cifar_t50_xcd --workload cifar32_ncsnpp --t 50 --steps 1 --warmup 1 --no-conv-profile
  abv DP_XCD_MAP 0 1 guided_adj_b32_t5_xcd --workload imagenet256_guided_sde_adjoint --batch 32 --t 5 --steps 1 --warmup 0 --no-conv-profile
  abv DP_XCD_MAP 0 1 cifar_adj_t20_xcd --workload cifar32_ncsnpp_adjoint --t 20 --steps 1 --warmup 1 --no-conv-profile
  ;;
final2)   # adjoint workloads at HEAD (after the one-pass kernel's addend prefetch)
  for B in 4 64; do
    timeout 500 python bench.py --workload imagenet256_guided_sde_adjoint --batch $B --steps 1 --warmup 0 --no-cpu-baseline --no-resident-call > "$O/bench_guided_sde_adjoint_b$B.json" 2> "$O/bench_guided_sde_adjoint_b$B.err"; lap bench_guided_sde_adjoint_b$B
    val "$O/bench_guided_sde_adjoint_b$B.json" "guided_sde_adjoint B=$B" | cut -c1-110
  done
  rocstats cifar_adjoint_t10 200 --workload cifar32_ncsnpp_adjoint --t 10 --steps 1 --warmup 0
  head -12 "$O/cifar_adjoint_t10_kernel_stats.csv" | cut -c1-150
  ;;
headprof)   # rocprofv3 kernel stats of the default command and of the reference's per-GPU batch at HEAD
  rocstats default 400 --steps 1 --warmup 0
  rocstats guided_b4_t10 200 --batch 4 --t 10 --steps 1 --warmup 0
  for f in "$O"/*_kernel_stats.csv; do echo "== $f"; head -16 "$f" | cut -c1-150; done
  ;;
fusedadd)   # one-pass GroupNorm backward: the skip gradient fetched with x and dy
  timeout 600 python -m pytest tests/test_gpu_grad.py -m gpu -q -x -k "group_norm_bwd or ncsnpp or vjp" > "$O/tests.log" 2>&1; echo "rc=$?" >> "$O/tests.log"; lap tests
  grep -E "passed|failed|^FAILED|^E  " "$O/tests.log" | head
  abv DP_GNB_LEAN 0 1 cifar_adj_t20_fusedadd --workload cifar32_ncsnpp_adjoint --t 20 --steps 1 --warmup 1 --no-conv-profile
  ;;
gnblean)   # the lean apply pass of the three-launch GroupNorm backward
  timeout 600 python -m pytest tests/test_gpu_grad.py -m gpu -q -x -k "group_norm_bwd" > "$O/gnblean_tests.log" 2>&1; echo "rc=$?" >> "$O/gnblean_tests.log"; lap gnblean_tests
  grep -E "passed|failed|^FAILED|^E  " "$O/gnblean_tests.log" | head
  for V in 0 1; do DP_GNB_LEAN=$V PROBE_WHOLE_ONLY=1 timeout 300 python tests/probes/gn_bwd_chunk_probe.py > "$O/gn_bwd_probe_lean$V.log" 2>&1; done; lap probe
  paste -d'\n' "$O/gn_bwd_probe_lean0.log" "$O/gn_bwd_probe_lean1.log" | grep "whole" | cut -c1-110
  abv DP_GNB_LEAN 0 1 guided_adj_b32_t5_gnblean --workload imagenet256_guided_sde_adjoint --batch 32 --t 5 --steps 1 --warmup 0 --no-conv-profile
  abv DP_GNB_LEAN 0 1 guided_adj_b4_t10_gnblean --workload imagenet256_guided_sde_adjoint --batch 4 --t 10 --steps 1 --warmup 0 --no-conv-profile
  abv DP_GNB_LEAN 0 1 cifar_adj_t20_gnblean --workload cifar32_ncsnpp_adjoint --t 20 --steps 1 --warmup 1 --no-conv-profile
  ;;
gnbchunk)   # does the Infinity Cache serve the apply pass of the GroupNorm backward when statistics + apply run sample by sample?
  timeout 600 python tests/probes/gn_bwd_chunk_probe.py > "$O/gn_bwd_chunk_probe.log" 2>&1; lap probe
  cat "$O/gn_bwd_chunk_probe.log"
  ;;
gnbnt)   # non-temporal hints in the three-launch GroupNorm backward (ImageNet adjoint: 17.5 % of the step)
  timeout 600 python -m pytest tests/test_gpu_grad.py -m gpu -q -x -k "group_norm_bwd" > "$O/gnbnt_tests.log" 2>&1; echo "rc=$?" >> "$O/gnbnt_tests.log"; lap gnbnt_tests
  grep -E "passed|failed|^FAILED|^E  " "$O/gnbnt_tests.log" | head
  for V in 0 -1 0 -1 3; do
    DP_GNB_NT=$V timeout 400 python bench.py --workload imagenet256_guided_sde_adjoint --batch 32 --t 5 --steps 1 --warmup 0 --no-conv-profile --no-cpu-baseline --no-resident-call > "$O/bench_gnbnt_$V.json" 2>> "$O/bench_ab.err"
    val "$O/bench_gnbnt_$V.json" "guided_adj_b32_t5 DP_GNB_NT=$V" | cut -c1-100 | tee -a "$O/guided_adj_b32_t5_gnbnt_ab.log"
  done; lap ab
  ;;
closing)   # closing state of round 6: the whole GPU suite, every bench line but the driver's (stage `driver`), rocprofv3 stats of five workloads, batch table
  gputests
  timeout 300 python bench.py --workload cifar32_ncsnpp --steps 3 --warmup 1 > "$O/bench_cifar_b256_f16sr.json" 2> "$O/bench_cifar.err"; lap bench_cifar
  timeout 300 python bench.py --workload cifar32_ncsnpp --steps 3 --warmup 1 --no-conv-profile --no-cpu-baseline > "$O/bench_cifar_b256_f16sr_noprofile.json" 2>> "$O/bench_cifar.err"; lap bench_cifar_np
  timeout 400 python bench.py --workload cifar32_ncsnpp_adjoint --steps 2 --warmup 1 > "$O/bench_cifar_adjoint_b128_f16sr.json" 2> "$O/bench_adjoint.err"; lap bench_adjoint
  timeout 400 python bench.py --workload cifar32_ncsnpp_adjoint --steps 2 --warmup 1 --no-conv-profile --no-cpu-baseline > "$O/bench_cifar_adjoint_b128_f16sr_noprofile.json" 2>> "$O/bench_adjoint.err"; lap bench_adjoint_np
  timeout 300 python bench.py --t 150 --dt 1.5e-3 --steps 1 --warmup 1 --no-cpu-baseline --no-resident-call > "$O/bench_t150_dt1.5e-3_100step.json" 2> "$O/bench_t150a.err"; lap bench_t150_100
  timeout 300 python bench.py --t 150 --steps 1 --warmup 1 --no-cpu-baseline --no-resident-call > "$O/bench_t150_150step.json" 2> "$O/bench_t150b.err"; lap bench_t150_150
  for B in 4 32 64; do
    timeout 500 python bench.py --workload imagenet256_guided_sde_adjoint --batch $B --steps 1 --warmup 0 $([ $B != 32 ] && echo --no-cpu-baseline) --no-resident-call > "$O/bench_guided_sde_adjoint_b$B.json" 2> "$O/bench_guided_sde_adjoint_b$B.err"; lap bench_guided_sde_adjoint_b$B
  done
  rocstats cifar_t10 200 --workload cifar32_ncsnpp --t 10 --steps 1 --warmup 0
  rocstats cifar_adjoint_t10 200 --workload cifar32_ncsnpp_adjoint --t 10 --steps 1 --warmup 0
  rocstats guided_b4_t10 200 --batch 4 --t 10 --steps 1 --warmup 0
  rocstats guided_sde_adjoint_b32_t5 300 --workload imagenet256_guided_sde_adjoint --batch 32 --t 5 --steps 1 --warmup 0
  rocstats default 400 --steps 1 --warmup 0
  timeout 500 python tools/batch_table.py > "$O/batch_table.json" 2> "$O/batch_table.md"; lap batch_table
  cat "$O/batch_table.md"
  python - "$O" <<'PY'
import json, glob, os, sys
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]; g = r.get("second_kernel") or {}
        print(os.path.basename(f), "images/s", round(d["value"], 3), "resident", (d.get("input") or {}).get("value_resident_batch_engine_call"), "conv TF", r.get("achieved") and round(r["achieved"], 1),
              "frac", r.get("frac") and round(r["frac"], 3), "share", r.get("time_share_of_step") and round(r["time_share_of_step"], 3), "sclk", (r.get("sclk_mhz") or {}).get("median"),
              "GN GB/s", g.get("achieved") and round(g["achieved"]), "peak GiB", d.get("peak_device_memory_gib") and round(d["peak_device_memory_gib"], 1), "cpu", (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "unreadable", e)
PY
  ;;
driver)   # the driver's own command, as BENCH_r05.json records it (20 timed + 5 warm-up purifications: ~7 minutes at the power cap)
  timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_driver_command.json" 2> "$O/bench_driver_command.err"; lap driver_command
  python - "$O/bench_driver_command.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print("value", round(d["value"], 4), "ms/step", round(d["ms_per_step"], 1), "frac", round(r["frac"], 4), "share", round(r["time_share_of_step"], 4),
      "sclk", r.get("sclk_mhz"), "traffic/alg", r.get("traffic_over_algorithmic"), "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
  ;;
smallb)   # small per-GPU batches (the reference's own B = 4): GroupNorm-apply rows cut across workgroups; weight rounding prefetched on a side stream
  timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "group_norm or resample or to_h2 or operand" > "$O/gnwg_tests.log" 2>&1; echo "rc=$?" >> "$O/gnwg_tests.log"; lap gnwg_tests
  grep -E "passed|failed|^FAILED|^E  " "$O/gnwg_tests.log" | head -20
  for B in 4 8 16; do
    abv DP_GN_WG 0 2048 guided_b${B}_t20_gnwg --batch $B --t 20 --steps 1 --warmup 1 --no-conv-profile
    ab DIFFPURE_ROUND_PREFETCH guided_b${B}_t20_prefetch --batch $B --t 20 --steps 1 --warmup 1 --no-conv-profile
  done
  abv DP_GN_WG 0 2048 cifar_t50_gnwg --workload cifar32_ncsnpp --t 50 --steps 1 --warmup 1 --no-conv-profile
  abv DP_GN_WG 0 2048 guided_adj_b4_t10_gnwg --workload imagenet256_guided_sde_adjoint --batch 4 --t 10 --steps 1 --warmup 0 --no-conv-profile
  ;;
dw128)   # the 8-wave kernel's 512x128 form on the layers with 128 output channels (NCSN++ 32x32 level)
  timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "512x128 or fp16_weights_single_pass or k_segments or half_height or fp16_output" > "$O/dw128_tests.log" 2>&1; echo "rc=$?" >> "$O/dw128_tests.log"; lap dw128_tests
  grep -E "passed|failed|^FAILED|^E  " "$O/dw128_tests.log" | head -20
  DP_H2_DW=1 timeout 300 python tests/probes/cifar_conv_shapes.py > "$O/cifar_conv_shapes_dw1.log" 2>&1; lap shapes_dw1
  DP_H2_DW=2 timeout 300 python tests/probes/cifar_conv_shapes.py > "$O/cifar_conv_shapes_dw2.log" 2>&1; lap shapes_dw2
  grep -E "^ 32|weighted" "$O/cifar_conv_shapes_dw1.log" | head -12; echo ---; grep -E "^ 32|weighted" "$O/cifar_conv_shapes_dw2.log" | head -12
  abv DP_H2_DW 1 2 cifar_t50_dw128 --workload cifar32_ncsnpp --t 50 --steps 1 --warmup 1 --no-conv-profile
  abv DP_H2_DW 1 2 cifar_adj_t20_dw128 --workload cifar32_ncsnpp_adjoint --t 20 --steps 1 --warmup 1 --no-conv-profile
  abv DP_H2_DW 1 2 guided_t20_dw128 --t 20 --steps 1 --warmup 1 --no-conv-profile
  ;;
stem)
  timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "stem" > "$O/stem_tests.log" 2>&1; echo "rc=$?" >> "$O/stem_tests.log"; lap stem_tests
  tail -15 "$O/stem_tests.log"
  timeout 200 python tests/probes/stem_bench.py > "$O/stem_bench.log" 2>&1; lap stem_bench
  cat "$O/stem_bench.log"
  ab DIFFPURE_STEM16 guided_t20 --t 20 --steps 1 --warmup 1
  ;;
tests)
  gputests
  ;;
profiles)   # rocprofv3 kernel stats of the four workloads at HEAD
  rocstats cifar_t10 200 --workload cifar32_ncsnpp --t 10 --steps 1 --warmup 0
  rocstats cifar_adjoint_t10 200 --workload cifar32_ncsnpp_adjoint --t 10 --steps 1 --warmup 0
  rocstats guided_b4_t10 200 --batch 4 --t 10 --steps 1 --warmup 0
  rocstats guided_sde_adjoint_b32_t5 300 --workload imagenet256_guided_sde_adjoint --batch 32 --t 5 --steps 1 --warmup 0
  rocstats default 400 --steps 1 --warmup 0
  for f in "$O"/*_kernel_stats.csv; do echo "== $f"; head -14 "$f" | cut -c1-150; done
  ;;
pmc)   # HBM traffic per kernel NAME (FETCH_SIZE / WRITE_SIZE in separate passes) + MFMA-busy of the dominant shape; configs[0] golden test
  timeout 300 python -m pytest tests/test_gpu_loops.py -m gpu -q -s -k "config1_cifar" > "$O/c1_tests.log" 2>&1; echo "rc=$?" >> "$O/c1_tests.log"; lap c1_tests
  grep -E "passed|failed|configs\[0\]" "$O/c1_tests.log"
  PMC_TRAFFIC_UPDATE="round 6 (tools/r06_measure.sh pmc): one row per kernel name" timeout 500 bash tools/pmc_traffic.sh --precision f16sr > "$O/pmc.log" 2>&1; lap pmc_traffic
  find "$R/gpurun_out/pmc_traffic" -name "*counter_collection.csv" -delete
  cp "$R"/gpurun_out/pmc_traffic/*.json "$O/" 2>/dev/null
  tail -14 "$O/pmc.log"
  PMC_GROUPS="sq1 grbm" timeout 400 bash tools/pmc_conv.sh dw_r06 --dw 1 --res16 --f16out > "$O/pmc_conv.log" 2>&1; lap pmc_conv
  cp "$R/gpurun_out/pmc_conv/dw_r06.json" "$O/" 2>/dev/null
  tail -30 "$O/pmc_conv.log"
  ;;
taprobe)   # how fast a CU takes MFMA A-fragments straight from L2 (fragment addressing vs quad-contiguous addressing)
  timeout 120 tests/probes/bin/ta_rate_probe > "$O/ta_rate_probe.log" 2>&1; lap ta_rate_probe
  cat "$O/ta_rate_probe.log"
  ;;
h16n64)  # attention backward of the 64-wide heads on the fp16 matrix cores
  timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_grad.py tests/test_gpu_loops.py -m gpu -q -s -k "gemm_strided or attention_backward or attention_bwd or guided_small_vjp or guided_full_vjp or guided_sde_stochastic_adjoint or bucketed" > "$O/h16_tests.log" 2>&1; echo "rc=$?" >> "$O/h16_tests.log"; lap h16_tests
  grep -E "passed|failed|^FAILED|^E  |attention backward|adjoint" "$O/h16_tests.log" | head -30
  ab DIFFPURE_H16_N64 guided_adj_b32_t5_h16n64 --workload imagenet256_guided_sde_adjoint --batch 32 --t 5 --steps 1 --warmup 0 --no-conv-profile
  ;;
gnnt)   # non-temporal hints in GroupNorm-apply
  timeout 300 python tests/probes/gn_bench.py > "$O/gn_bench_nt.log" 2>&1; lap gn_bench
  cat "$O/gn_bench_nt.log"
  for V in 0 3 0 3; do
    DP_GN_NT=$V timeout 400 python bench.py --t 20 --steps 1 --warmup 1 --no-conv-profile --no-cpu-baseline --no-resident-call > "$O/bench_gnnt_$V.json" 2>> "$O/bench_ab.err"
    val "$O/bench_gnnt_$V.json" "guided t20 DP_GN_NT=$V" | cut -c1-70 | tee -a "$O/gnnt_ab.log"
  done; lap ab_gnnt
  ;;
bucket)    # batch-bucketed split-K: the whole GPU suite, then the batch table and the small-batch adjoint with and without it
  gputests
  ab DIFFPURE_BATCH_INVARIANT guided_b4_t20_invariant --batch 4 --t 20 --steps 1 --warmup 1 --no-conv-profile
  ab DIFFPURE_BATCH_INVARIANT guided_b8_t20_invariant --batch 8 --t 20 --steps 1 --warmup 1 --no-conv-profile
  ab DIFFPURE_BATCH_INVARIANT guided_b16_t20_invariant --batch 16 --t 20 --steps 1 --warmup 1 --no-conv-profile
  ab DIFFPURE_BATCH_INVARIANT guided_adj_b4_t10_invariant --workload imagenet256_guided_sde_adjoint --batch 4 --t 10 --steps 1 --warmup 0 --no-conv-profile
  timeout 500 python tools/batch_table.py > "$O/batch_table.json" 2> "$O/batch_table.md"; lap batch_table
  cat "$O/batch_table.md"
  ;;
boundary)    # fused block boundary of the <= 64-pixel levels
  timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -s -k "boundary or prefetch or fp16_input_equals" > "$O/boundary_tests.log" 2>&1; echo "rc=$?" >> "$O/boundary_tests.log"; lap boundary_tests
  grep -E "passed|failed|^FAILED|^E  " "$O/boundary_tests.log" | head -40
  timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_grad.py tests/test_reference_driver.py tests/test_gpu_dist.py -m gpu -q -s -x > "$O/model_tests.log" 2>&1; echo "rc=$?" >> "$O/model_tests.log"; lap model_tests
  grep -E "passed|failed|^FAILED|^E  |gain " "$O/model_tests.log" | head -40
  ab DIFFPURE_BOUNDARY cifar_t50_boundary --workload cifar32_ncsnpp --t 50 --steps 1 --warmup 1 --no-conv-profile
  ab DIFFPURE_BOUNDARY cifar_adj_t20_boundary --workload cifar32_ncsnpp_adjoint --t 20 --steps 1 --warmup 1 --no-conv-profile
  ab DIFFPURE_BOUNDARY guided_t20_boundary --t 20 --steps 1 --warmup 1 --no-conv-profile
  ab DIFFPURE_BOUNDARY guided_b4_t20_boundary --batch 4 --t 20 --steps 1 --warmup 1 --no-conv-profile
  ;;
fuse)    # resampled identity skip as a second output of GroupNorm-apply; weight rounding prefetched on a side stream
  timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_grad.py -m gpu -q -s -k "fp16_input_equals or tiny_cotangent or taped_and or finite_differences or half_height or stem or torch_ops or round_weights or stochastic" > "$O/fuse_tests.log" 2>&1; echo "rc=$?" >> "$O/fuse_tests.log"; lap fuse_tests
  grep -E "passed|failed|^FAILED|^E  |cotangent|probe direction|ode_vjp" "$O/fuse_tests.log" | head -40
  ab DIFFPURE_SKIP_FUSED guided_t20_skip --t 20 --steps 1 --warmup 1 --no-conv-profile
  ab DIFFPURE_ROUND_PREFETCH guided_t20_prefetch --t 20 --steps 1 --warmup 1 --no-conv-profile
  ab DIFFPURE_SKIP_FUSED cifar_t50_skip --workload cifar32_ncsnpp --t 50 --steps 1 --warmup 1 --no-conv-profile
  ab DIFFPURE_ROUND_PREFETCH cifar_t50_prefetch --workload cifar32_ncsnpp --t 50 --steps 1 --warmup 1 --no-conv-profile
  ab DIFFPURE_ROUND_PREFETCH cifar_adj_t20_prefetch --workload cifar32_ncsnpp_adjoint --t 20 --steps 1 --warmup 1 --no-conv-profile
  ;;
*)
  echo "unknown stage $STAGE"; exit 1;;
esac
lap done
cat "$O/timeline.log"
