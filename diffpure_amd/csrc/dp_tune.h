// Tuning switches of libdiffpure_hip.so.  Each switch is read from its environment variable ONCE, when the library first
// asks for any of them; after that the environment is never consulted again (a variable set mid-process changes nothing).
// Probes and tests flip a switch in-process through the C ABI (dp_set_tuning, include/diffpure_hip.h).  None of these
// switches but the last (DIFFPURE_BATCH_INVARIANT, see below) changes a result: every variant they select is bit-identical to the others (tests/test_gpu_ops.py).  The timing
// ablations that DO break results (DP_ABLATE builds) are not part of this library: tests/probes/build_ablate.py compiles
// them into a separate libdiffpure_hip_ablate.so.  The switches are PROCESS-WIDE (relaxed atomics: a thread that flips one
// while another thread launches is well-defined, and the other thread's launches pick either variant - same bits either way).
// Round 4 removed the switches of the variants that measured slower and were pruned from the library (two workgroups per CU,
// persistent tile loop, halo tile, four-phase ping-pong schedule, start-up staggers, GroupNorm fold, non-quad GroupNorm-apply) and of
// the slice-unrolled / static-priority forms of the 8-wave kernel that its asymmetric staging superseded: 16 -> 5 switches (round 5 adds DP_H2_DH for the new half-height tile kernel: 6).
#pragma once

enum DpTune {
    DP_T_H2_PP = 0,        // DP_H2_PP: 256-wide tile kernels at all (ping-pong / one-wave-per-SIMD / 8-wave) - 0 never, 1 whenever the shape allows, 2 when it also fills the chip
    DP_T_H2_SW,            // DP_H2_SW: one-wave-per-SIMD kernel (igemm_h2_sw.hip) - 0 off, 1 its 256x256 tiles only, 2 also 512x128 tiles (N % 256 != 0)
    DP_T_H2_NN,            // DP_H2_NN: few-output-channels kernel - 0 off
    DP_T_H2_DW,            // DP_H2_DW: the 8-wave kernel (igemm_h2_dw.hip) on launches of >= 256 tiles - 0 off, 1 its 256x256 tiles only, 2 (default, round 6) also 512x128 tiles (N % 256 != 0)
    DP_T_H2_DH,            // DP_H2_DH: the 4-wave 128x256 kernel (igemm_h2_dh.hip) on launches of fewer than 256 tiles of 256x256 - 0 off, 1 un-split layers only, 2 also the split-K levels
    DP_T_H2_DH_MIN,        // DP_H2_DH_MIN: fewest 128x256 half tiles (x split-K parts) of a launch that kernel takes (default 32; below: the generic tiles)
    DP_T_GN_FINALIZE_SAMPLE, // DP_GN_FINALIZE_SAMPLE: small feature maps - one finalize workgroup per sample instead of per (sample, group) - 0 off
    DP_T_BATCH_INVARIANT,  // DIFFPURE_BATCH_INVARIANT (round 6) - THE ONE SWITCH THAT CHANGES BITS (not accuracy): 0 (default) = few-tile / long-K
                           // convolution launches are split along K by a factor chosen per (layer shape, batch bucket), so a sample's low-order
                           // bits depend on the bucket its per-GPU batch falls in; 1 = the split factor is a function of the layer shape only
                           // (rounds 1-5): bit-identical results for ANY batch size / sharding, at the price of starved launches at small batches
    DP_T_GN_NT,            // DP_GN_NT: non-temporal hints in GroupNorm-apply over the fp16 stream - -1 (default) by tensor size, 0 never, 1 | 2 | 3 forced
                           // (bit 0 loads, bit 1 stores); same bits
    DP_T_GN_WG,            // DP_GN_WG (round 6): GroupNorm-apply over the fp16 stream cuts an output row across several workgroups while the launch has
                           // fewer workgroups than this (default 2048; 0 = always one workgroup per row, rounds 3-5); same bits
    DP_T_GNB_NT,           // DP_GNB_NT (round 6): non-temporal hints in the three-launch GroupNorm backward - -1 (default) by tensor size, 0 never, 1 | 2 | 3 forced; same bits
    DP_T_GNB_LEAN,         // DP_GNB_LEAN (round 6): the lean apply pass of the three-launch GroupNorm backward (un-resampled, fp16 tape) - 0 off; same bits
    DP_T_XCD_MAP,          // DP_XCD_MAP (round 6): XCD-aware workgroup maps of the secondary kernels - workgroups are dealt to the 8 XCDs round-robin by
                           // blockIdx, so neighbours in blockIdx order that share cache lines (gn_finalize_cols: four groups per 128-byte record line;
                           // one-pass GroupNorm backward: two 16-channel blocks per line) or stream the same operand (flash attention: the query blocks of
                           // a head; gemm_strided_h16: the tiles of a batch entry) are re-dealt onto ONE XCD - 0 off; same bits
    DP_T_COUNT
};

int dp_tune(DpTune k);
