#!/bin/bash
# Round-5 measurement calls on the GPU box (gpurun).  Everything lands under gpurun_out/r05/<stage>/ (copied into profiles/r05/ afterwards).
#   bash tools/r05_measure.sh explore     per-shape CIFAR table, rocprofv3 kernel stats of the CIFAR forward / adjoint at HEAD, power / clock
#                                         experiment of the dominant kernel, batch-sensitivity table, the ImageNet SDE-adjoint bench lines
#   bash tools/r05_measure.sh tests       the whole -m gpu suite + smoke
#   bash tools/r05_measure.sh bench       default bench line (runner boundary) + rocprofv3 kernel stats of the same command
#   bash tools/r05_measure.sh closing     tests + bench + every other bench line + PMC passes (the round's closing state)
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
STAGE=${1:-explore}
O="$R/gpurun_out/r05/$STAGE"
rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp
S=$(date +%s)
lap() { echo "[$(( $(date +%s) - S )) s] $1" >> "$O/timeline.log"; }
: > "$O/timeline.log"
rocstats() {   # rocstats <name> <timeout> <bench args...>: rocprofv3 kernel stats of one bench command
  local name=$1 to=$2; shift 2
  ( cd /tmp && timeout "$to" rocprofv3 --kernel-trace --stats -d "$O/prof_$name" -o run --output-format csv -- python "$R/bench.py" "$@" --no-cpu-baseline --no-resident-call > "$O/bench_${name}_under_rocprof.json" 2> "$O/rocprof_$name.err" )
  find "$O/prof_$name" -name "*kernel_trace.csv" -delete; find "$O/prof_$name" -name "*agent_info.csv" -delete
  f=$(find "$O/prof_$name" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/${name}_kernel_stats.csv"
  lap "rocprof_$name"
}
gputests() {
  timeout 1500 python -m pytest tests -m gpu -q -s > "$O/gpu_tests.log" 2>&1; echo "rc=$?" >> "$O/gpu_tests.log"; lap gpu_tests
  timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.log" 2>&1; echo "rc=$?" >> "$O/smoke.log"; lap smoke
  grep -E "passed|failed" "$O/gpu_tests.log" | tail -2; grep -E "^FAILED|^ERROR" "$O/gpu_tests.log" | head; tail -2 "$O/smoke.log"
}
benchdefault() {
  timeout 600 python bench.py --steps 2 --warmup 1 > "$O/bench_default_f16sr_b64.json" 2> "$O/bench_default.err"; lap bench_default
  rocstats default 300 --steps 1 --warmup 0
}
case "$STAGE" in
explore)
  timeout 120 python tests/probes/cifar_conv_shapes.py > "$O/cifar_conv_shapes.log" 2>&1; lap cifar_conv_shapes
  rocstats cifar_t10 200 --workload cifar32_ncsnpp --t 10 --steps 1 --warmup 0
  rocstats cifar_adjoint_t10 200 --workload cifar32_ncsnpp_adjoint --t 10 --steps 1 --warmup 0
  timeout 150 python tests/probes/dw8_power.py --seconds 4 > "$O/dw8_power.log" 2>&1; lap dw8_power
  timeout 400 python tools/batch_table.py > "$O/batch_table.json" 2> "$O/batch_table.md"; lap batch_table
  for B in 4 32; do
    timeout 400 python bench.py --workload imagenet256_guided_sde_adjoint --batch $B --steps 1 --warmup 0 --no-cpu-baseline --no-resident-call > "$O/bench_guided_sde_adjoint_b$B.json" 2> "$O/bench_guided_sde_adjoint_b$B.err"; lap bench_guided_sde_adjoint_b$B
  done
  rocstats guided_sde_adjoint_b4_t10 300 --workload imagenet256_guided_sde_adjoint --batch 4 --t 10 --steps 1 --warmup 0
  ;;
dh)     # the half-height tile kernel: its tests, the per-shape A/B, the adjoint / small-batch bench lines with and without it
  timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "half_height or k_segments or fp16_output or offset_strided or torch_ops" > "$O/dh_tests.log" 2>&1; echo "rc=$?" >> "$O/dh_tests.log"; lap dh_tests
  tail -5 "$O/dh_tests.log"
  timeout 200 python tests/probes/cifar_conv_shapes.py > "$O/cifar_conv_shapes.log" 2>&1; lap cifar_conv_shapes
  for DH in 0 1 0 1; do
    DP_H2_DH=$DH timeout 300 python bench.py --workload cifar32_ncsnpp_adjoint --t 20 --steps 1 --warmup 1 --no-cpu-baseline --no-resident-call > "$O/bench_cifar_adjoint_t20_dh$DH.json" 2>> "$O/bench_ab.err"
    python -c "import json,sys; d=json.loads(open('$O/bench_cifar_adjoint_t20_dh$DH.json').read().strip().splitlines()[-1]); print('adjoint t20 DH=$DH', round(d['value'],2), 'images/s', d['roofline']['sclk_mhz'])" | tee -a "$O/dh_ab.log"
  done; lap adjoint_ab
  for DH in 0 1; do
    DP_H2_DH=$DH timeout 300 python bench.py --batch 4 --t 20 --steps 1 --warmup 1 --no-cpu-baseline --no-resident-call > "$O/bench_guided_b4_t20_dh$DH.json" 2>> "$O/bench_ab.err"
    python -c "import json,sys; d=json.loads(open('$O/bench_guided_b4_t20_dh$DH.json').read().strip().splitlines()[-1]); print('guided B=4 t20 DH=$DH', round(d['value'],3), 'images/s', d['roofline']['sclk_mhz'])" | tee -a "$O/dh_ab.log"
  done; lap guided_b4_ab
  timeout 150 python tests/probes/dw8_power.py --seconds 4 > "$O/dw8_power.log" 2>&1; lap dw8_power
  ;;
attn16)   # fp16 attention backward: tests + adjoint A/B
  timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_grad.py -m gpu -q -k "fp16_matrix_cores or torch_ops or attention_bwd or taped_and_untaped or finite_differences" > "$O/attn16_tests.log" 2>&1; echo "rc=$?" >> "$O/attn16_tests.log"; lap attn16_tests
  grep -E "passed|failed|^FAILED|^E  |attention backward|f16sr" "$O/attn16_tests.log" | head -30
  for V in 0 1 0 1; do
    DIFFPURE_ATTN_BWD16=$V timeout 300 python bench.py --workload cifar32_ncsnpp_adjoint --t 20 --steps 1 --warmup 1 --no-cpu-baseline --no-resident-call > "$O/bench_cifar_adjoint_t20_attn16_$V.json" 2>> "$O/bench_ab.err"
    python -c "import json,sys; d=json.loads(open('$O/bench_cifar_adjoint_t20_attn16_$V.json').read().strip().splitlines()[-1]); print('adjoint t20 ATTN_BWD16=$V', round(d['value'],2), 'images/s', d['roofline']['sclk_mhz'])" | tee -a "$O/attn16_ab.log"
  done; lap adjoint_ab
  ;;
tape16)   # the taped forward on the fp16 residual stream: gradient tests + A/B
  timeout 900 python -m pytest tests/test_gpu_grad.py tests/test_gpu_ops.py -m gpu -q -x -s -k "not conv2d" > "$O/grad_tests.log" 2>&1; echo "rc=$?" >> "$O/grad_tests.log"; lap grad_tests
  grep -E "passed|failed|^FAILED|^E  |f16sr|adjoint|VJP" "$O/grad_tests.log" | head -40
  timeout 900 python -m pytest tests/test_gpu_loops.py -m gpu -q -s -k "adjoint and not guided_sde_stochastic_adjoint_100 and not at_batch_8_reproduces or vjp" > "$O/loop_tests.log" 2>&1; echo "rc=$?" >> "$O/loop_tests.log"; lap loop_tests
  grep -E "passed|failed|^FAILED|^E  |adjoint|VJP" "$O/loop_tests.log" | head -40
  for V in 0 1 0 1; do
    DIFFPURE_TAPE16=$V timeout 300 python bench.py --workload cifar32_ncsnpp_adjoint --t 20 --steps 1 --warmup 1 --no-cpu-baseline --no-resident-call > "$O/bench_cifar_adjoint_t20_tape16_$V.json" 2>> "$O/bench_ab.err"
    python -c "import json,sys; d=json.loads(open('$O/bench_cifar_adjoint_t20_tape16_$V.json').read().strip().splitlines()[-1]); print('cifar adjoint t20 TAPE16=$V', round(d['value'],2), 'images/s', d['roofline']['sclk_mhz']['median'], 'peak GiB', round(d['peak_device_memory_gib'],1))" | tee -a "$O/tape16_ab.log"
  done; lap cifar_adjoint_ab
  for V in 0 1; do
    DIFFPURE_TAPE16=$V timeout 300 python bench.py --workload imagenet256_guided_sde_adjoint --batch 32 --t 10 --steps 1 --warmup 0 --no-cpu-baseline --no-resident-call > "$O/bench_guided_adjoint_b32_t10_tape16_$V.json" 2>> "$O/bench_ab.err"
    python -c "import json,sys; d=json.loads(open('$O/bench_guided_adjoint_b32_t10_tape16_$V.json').read().strip().splitlines()[-1]); print('guided sde adjoint B=32 t10 TAPE16=$V', round(d['value'],3), 'images/s', d['roofline']['sclk_mhz'], 'peak GiB', round(d['peak_device_memory_gib'],1))" | tee -a "$O/tape16_ab.log"
  done; lap guided_adjoint_ab
  rocstats guided_sde_adjoint_b32_t5 300 --workload imagenet256_guided_sde_adjoint --batch 32 --t 5 --steps 1 --warmup 0
  ;;
streams)
  timeout 200 python tests/probes/two_stream_cifar.py 128 20 2 > "$O/two_stream_cifar.log" 2>&1; lap two_stream_2x128
  timeout 200 python tests/probes/two_stream_cifar.py 64 20 4 >> "$O/two_stream_cifar.log" 2>&1; lap two_stream_4x64
  cat "$O/two_stream_cifar.log"
  ;;
graph)
  for G in 0 1 0 1; do
    DIFFPURE_GRAPH=$G timeout 300 python bench.py --workload cifar32_ncsnpp --t 20 --steps 2 --warmup 1 --no-cpu-baseline --no-resident-call --no-conv-profile --engine-call > "$O/bench_cifar_t20_graph$G.json" 2>> "$O/bench_ab.err"
    python -c "import json,sys; d=json.loads(open('$O/bench_cifar_t20_graph$G.json').read().strip().splitlines()[-1]); print('cifar t20 engine-call GRAPH=$G', round(d['value'],1), 'images/s', round(d['ms_per_step']/20,2), 'ms/UNet step', d['roofline']['sclk_mhz']['median'])" | tee -a "$O/graph_ab.log"
  done; lap graph_ab
  for G in 0 1; do
    DIFFPURE_GRAPH=$G timeout 300 python bench.py --batch 4 --t 20 --steps 2 --warmup 1 --no-cpu-baseline --no-resident-call --no-conv-profile --engine-call > "$O/bench_guided_b4_t20_graph$G.json" 2>> "$O/bench_ab.err"
    python -c "import json,sys; d=json.loads(open('$O/bench_guided_b4_t20_graph$G.json').read().strip().splitlines()[-1]); print('guided B=4 t20 engine-call GRAPH=$G', round(d['value'],2), 'images/s', d['roofline']['sclk_mhz']['median'])" | tee -a "$O/graph_ab.log"
  done; lap graph_b4
  ;;
dh128)
  timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "half_height" > "$O/dh_tests.log" 2>&1; echo "rc=$?" >> "$O/dh_tests.log"; lap dh_tests
  tail -4 "$O/dh_tests.log"
  PROBE_ALT_DH=3 timeout 200 python tests/probes/cifar_conv_shapes.py > "$O/cifar_conv_shapes_dh3.log" 2>&1; lap cifar_conv_shapes
  grep -E "^ 32|^==|^--" "$O/cifar_conv_shapes_dh3.log"
  for DH in 2 3 2 3; do
    DP_H2_DH=$DH timeout 300 python bench.py --workload cifar32_ncsnpp --t 20 --steps 2 --warmup 1 --no-cpu-baseline --no-resident-call --no-conv-profile > "$O/bench_cifar_t20_dh$DH.json" 2>> "$O/bench_ab.err"
    python -c "import json,sys; d=json.loads(open('$O/bench_cifar_t20_dh$DH.json').read().strip().splitlines()[-1]); print('cifar t20 DH=$DH', round(d['value'],1), 'images/s', d['roofline']['sclk_mhz']['median'])" | tee -a "$O/dh128_ab.log"
  done; lap cifar_ab
  ;;
smallb)
  timeout 300 python tests/probes/smallbatch_conv_shapes.py > "$O/smallbatch_conv_shapes.log" 2>&1; lap smallbatch_shapes
  cat "$O/smallbatch_conv_shapes.log"
  for M in 128 64 32; do
    DP_H2_DH_MIN=$M timeout 300 python bench.py --batch 4 --t 20 --steps 2 --warmup 1 --no-cpu-baseline --no-resident-call --no-conv-profile > "$O/bench_guided_b4_t20_min$M.json" 2>> "$O/bench_ab.err"
    python -c "import json,sys; d=json.loads(open('$O/bench_guided_b4_t20_min$M.json').read().strip().splitlines()[-1]); print('guided B=4 t20 DH_MIN=$M', round(d['value'],3), 'images/s', d['roofline']['sclk_mhz']['median'])" | tee -a "$O/dhmin_ab.log"
  done; lap guided_b4
  for M in 128 32; do
    DP_H2_DH_MIN=$M timeout 300 python bench.py --batch 16 --t 20 --steps 2 --warmup 1 --no-cpu-baseline --no-resident-call --no-conv-profile > "$O/bench_guided_b16_t20_min$M.json" 2>> "$O/bench_ab.err"
    python -c "import json,sys; d=json.loads(open('$O/bench_guided_b16_t20_min$M.json').read().strip().splitlines()[-1]); print('guided B=16 t20 DH_MIN=$M', round(d['value'],3), 'images/s', d['roofline']['sclk_mhz']['median'])" | tee -a "$O/dhmin_ab.log"
  done; lap guided_b16
  ;;
final)    # the state the round closes on: whole suite + smoke + the builder-run bench lines (more timed steps: the first one carries the per-launch profile)
  gputests
  timeout 600 python bench.py --steps 3 --warmup 1 > "$O/bench_default_f16sr_b64.json" 2> "$O/bench_default.err"; lap bench_default
  timeout 400 python bench.py --workload cifar32_ncsnpp --steps 10 --warmup 1 > "$O/bench_cifar_b256_f16sr.json" 2> "$O/bench_cifar.err"; lap bench_cifar
  timeout 500 python bench.py --workload cifar32_ncsnpp_adjoint --steps 5 --warmup 1 > "$O/bench_cifar_adjoint_b128_f16sr.json" 2> "$O/bench_adjoint.err"; lap bench_adjoint
  timeout 300 python bench.py --workload imagenet256_guided_sde_adjoint --batch 4 --steps 2 --warmup 0 --no-cpu-baseline --no-resident-call > "$O/bench_guided_sde_adjoint_b4.json" 2> "$O/bench_guided_sde_adjoint_b4.err"; lap bench_guided_sde_adjoint_b4
  timeout 300 python tools/batch_table.py --batches 4,8,16 > "$O/batch_table.json" 2> "$O/batch_table.md"; lap batch_table
  ;;
abfinal)
  for M in 128 32 128 32; do
    DP_H2_DH_MIN=$M timeout 300 python bench.py --workload cifar32_ncsnpp_adjoint --t 20 --steps 2 --warmup 1 --no-cpu-baseline --no-resident-call --no-conv-profile --engine-call > "$O/bench_cifar_adjoint_t20_min$M.json" 2>> "$O/bench_ab.err"
    python -c "import json,sys; d=json.loads(open('$O/bench_cifar_adjoint_t20_min$M.json').read().strip().splitlines()[-1]); print('cifar adjoint t20 engine-call DH_MIN=$M', round(d['value'],2), 'images/s', d['roofline']['sclk_mhz']['median'])" | tee -a "$O/abfinal.log"
  done; lap adjoint_ab
  for M in 128 32; do
    DP_H2_DH_MIN=$M timeout 300 python bench.py --workload cifar32_ncsnpp --t 20 --steps 2 --warmup 1 --no-cpu-baseline --no-resident-call --no-conv-profile --engine-call > "$O/bench_cifar_t20_min$M.json" 2>> "$O/bench_ab.err"
    python -c "import json,sys; d=json.loads(open('$O/bench_cifar_t20_min$M.json').read().strip().splitlines()[-1]); print('cifar t20 engine-call DH_MIN=$M', round(d['value'],1), 'images/s', d['roofline']['sclk_mhz']['median'])" | tee -a "$O/abfinal.log"
  done; lap cifar_ab
  timeout 200 python tests/probes/cifar_conv_shapes.py > "$O/cifar_conv_shapes.log" 2>&1; lap shapes
  grep -E "^ 16|^  4|^  8|^==" "$O/cifar_conv_shapes.log"
  ;;
last)
  gputests
  timeout 600 python bench.py --steps 3 --warmup 1 > "$O/bench_default_f16sr_b64.json" 2> "$O/bench_default.err"; lap bench_default
  timeout 500 python bench.py --workload cifar32_ncsnpp_adjoint --steps 3 --warmup 1 --no-cpu-baseline > "$O/bench_cifar_adjoint_b128_f16sr.json" 2> "$O/bench_adjoint.err"; lap bench_adjoint
  timeout 300 python bench.py --workload cifar32_ncsnpp --steps 5 --warmup 1 --no-cpu-baseline > "$O/bench_cifar_b256_f16sr.json" 2> "$O/bench_cifar.err"; lap bench_cifar
  ;;
q16)
  timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_grad.py -m gpu -q -k "fp16_matrix_cores or attention_bwd or adjoint or vjp or taped or finite" > "$O/q16_tests.log" 2>&1; echo "rc=$?" >> "$O/q16_tests.log"; lap q16_tests
  tail -4 "$O/q16_tests.log"
  timeout 600 python -m pytest tests/test_gpu_loops.py -m gpu -q -s -k "config5 or sde_stochastic_adjoint_100_plus" > "$O/q16_loops.log" 2>&1; echo "rc=$?" >> "$O/q16_loops.log"; lap q16_loops
  grep -E "passed|failed|adjoint" "$O/q16_loops.log" | head
  timeout 300 python bench.py --workload cifar32_ncsnpp_adjoint --t 20 --steps 2 --warmup 1 --no-cpu-baseline --no-resident-call --no-conv-profile --engine-call > "$O/bench_cifar_adjoint_t20.json" 2>> "$O/bench.err"
  python -c "import json,sys; d=json.loads(open('$O/bench_cifar_adjoint_t20.json').read().strip().splitlines()[-1]); print('cifar adjoint t20 engine-call', round(d['value'],2), 'images/s', d['roofline']['sclk_mhz']['median'])"; lap bench
  ;;
head)
  gputests
  rocstats cifar_adjoint_t10 200 --workload cifar32_ncsnpp_adjoint --t 10 --steps 1 --warmup 0
  ;;
tests) gputests ;;
bench) benchdefault ;;
closing)
  gputests
  benchdefault
  timeout 400 bash tools/pmc_traffic.sh --precision f16sr > "$O/pmc.log" 2>&1; lap pmc_traffic
  find "$R/gpurun_out/pmc_traffic" -name "*counter_collection.csv" -delete
  PMC_GROUPS="sq1 sq2 sq3 tcc1 grbm" timeout 500 bash tools/pmc_conv.sh dw_r05 --dw 1 --res16 --f16out > "$O/pmc_conv.log" 2>&1; lap pmc_conv
  timeout 300 python bench.py --workload cifar32_ncsnpp --steps 3 --warmup 1 > "$O/bench_cifar_b256_f16sr.json" 2> "$O/bench_cifar.err"; lap bench_cifar
  timeout 400 python bench.py --workload cifar32_ncsnpp_adjoint --steps 2 --warmup 1 > "$O/bench_cifar_adjoint_b128_f16sr.json" 2> "$O/bench_adjoint.err"; lap bench_adjoint
  rocstats cifar_t10 200 --workload cifar32_ncsnpp --t 10 --steps 1 --warmup 0
  rocstats cifar_adjoint_t10 200 --workload cifar32_ncsnpp_adjoint --t 10 --steps 1 --warmup 0
  timeout 300 python bench.py --t 150 --dt 1.5e-3 --steps 1 --warmup 1 --no-cpu-baseline --no-resident-call > "$O/bench_t150_dt1.5e-3_100step.json" 2> "$O/bench_t150a.err"; lap bench_t150_100
  timeout 300 python bench.py --t 150 --steps 1 --warmup 1 --no-cpu-baseline --no-resident-call > "$O/bench_t150_150step.json" 2> "$O/bench_t150b.err"; lap bench_t150_150
  for B in 4 32 64; do
    timeout 500 python bench.py --workload imagenet256_guided_sde_adjoint --batch $B --steps 1 --warmup 0 $([ $B != 32 ] && echo --no-cpu-baseline) --no-resident-call > "$O/bench_guided_sde_adjoint_b$B.json" 2> "$O/bench_guided_sde_adjoint_b$B.err"; lap bench_guided_sde_adjoint_b$B
  done
  rocstats guided_sde_adjoint_b32_t5 300 --workload imagenet256_guided_sde_adjoint --batch 32 --t 5 --steps 1 --warmup 0
  timeout 400 python tools/batch_table.py > "$O/batch_table.json" 2> "$O/batch_table.md"; lap batch_table
  ;;
esac
cat "$O/timeline.log"
python - "$O" <<'P'
import json, glob, os, sys
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        g = r.get("second_kernel") or {}
        print(os.path.basename(f), "images/s", round(d["value"], 3), "resident", (d.get("input") or {}).get("value_resident_batch_engine_call"), "conv TF", r.get("achieved") and round(r["achieved"], 1),
              "frac", r.get("frac") and round(r["frac"], 3), "held", r.get("frac_at_held_clock") and round(r["frac_at_held_clock"], 3), "share", r.get("time_share_of_step") and round(r["time_share_of_step"], 3),
              "sclk", (r.get("sclk_mhz") or {}).get("median"), "GN GB/s", g.get("achieved") and round(g["achieved"]), "peak GiB", d.get("peak_device_memory_gib") and round(d["peak_device_memory_gib"], 1),
              "cpu", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("kind"), "cores", (d.get("cpu_baseline") or {}).get("cores"))
    except Exception as e:
        print(f, "unreadable", e)
P
for f in "$O"/*_kernel_stats.csv; do [ -f "$f" ] && { echo "== $f"; head -14 "$f" | cut -c1-160; }; done
[ -f "$O/cifar_conv_shapes.log" ] && cat "$O/cifar_conv_shapes.log"
[ -f "$O/dw8_power.log" ] && cat "$O/dw8_power.log"
[ -f "$O/batch_table.md" ] && cat "$O/batch_table.md"
