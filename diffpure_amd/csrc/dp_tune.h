// Tuning switches of libdiffpure_hip.so.  Each switch is read from its environment variable ONCE, when the library first
// asks for any of them; after that the environment is never consulted again (a variable set mid-process changes nothing).
// Probes and tests flip a switch in-process through the C ABI (dp_set_tuning, include/diffpure_hip.h).  None of these
// switches changes a result: every variant they select is bit-identical to the others (tests/test_gpu_ops.py).  The timing
// ablations that DO break results (DP_ABLATE builds) are not part of this library: tests/probes/build_ablate.py compiles
// them into a separate libdiffpure_hip_ablate.so.
#pragma once

enum DpTune {
    DP_T_H2_PP = 0,        // DP_H2_PP: 8-wave ping-pong kernels - 0 never, 1 whenever the shape allows, 2 when it also fills the chip
    DP_T_H2_HALO,          // DP_H2_HALO: halo-tile variant of the ping-pong kernel - 0 never, 1 W >= 16, 2 W >= 32
    DP_T_H2_SW,            // DP_H2_SW: one-wave-per-SIMD kernel (igemm_h2_sw.hip) - 0 off, 1 its 256x256 tiles only, 2 also 512x128 tiles (N % 256 != 0)
    DP_T_H2_SW_PERSIST,    // DP_H2_SW_PERSIST: that kernel as one persistent workgroup per CU (next tile's operands in flight under the epilogue) - 0 off
    DP_T_H2_NN,            // DP_H2_NN: few-output-channels kernel - 0 off
    DP_T_H2_PP_SCHED,      // DP_H2_PP_SCHED: phases per k-tile of the fp16-operand ping-pong kernels - 0 four, 1 two
    DP_T_H2_PP_STAGGER,    // DP_H2_PP_STAGGER: start-up stagger of the ping-pong kernel, cycles per k-tile and phase (0 off)
    DP_T_GN_APPLY_QUAD,    // DP_GN_APPLY_QUAD: lane-contiguous quad form of GroupNorm-apply - 0 off
    DP_T_H2_DW,            // DP_H2_DW: two-workgroups-per-CU 128x256 kernel (igemm_h2_dw.hip) - 0 off, 1 where the launch fills every CU twice, 2 wherever it applies, 8 its 8-wave 256x256 form
    DP_T_H2_DW_STAGGER,    // DP_H2_DW_STAGGER: half-tile start-up delay of a CU's second workgroup, cycles per k-tile (0 off)
    DP_T_H2_DW_MINROUNDS,  // DP_H2_DW_MINROUNDS: launches with fewer rounds of 512 tiles are not staggered
    DP_T_H2_DW_ADEPTH,     // DP_H2_DW_ADEPTH: stages of its activation ring - 3 (72 KB of LDS per workgroup) or 4 (80 KB)
    DP_T_GN_FOLD,          // DP_GN_FOLD: GroupNorm-apply reduces its own (sample, group) records instead of a finalize launch - 0 off
    DP_T_GN_FINALIZE_SAMPLE, // DP_GN_FINALIZE_SAMPLE: small feature maps - one finalize workgroup per sample instead of per (sample, group) - 0 off
    DP_T_H2_DW_UNROLL,     // DP_H2_DW_UNROLL: 3x3 launches of the 8-wave kernel run the slice-unrolled loop (nine taps per body) - 0 the rolled loop
    DP_T_H2_DW_PRIO,       // DP_H2_DW_PRIO: that loop with s_setprio 1 on waves 4-7 (the younger wave of every SIMD; timing only) - 0 off
    DP_T_COUNT
};

int dp_tune(DpTune k);
