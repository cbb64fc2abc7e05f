"""`torch.ops.diffpure_hip.*`: the hot operators registered with PyTorch's dispatcher FROM C++.

The registration lives in csrc/torch_binding.cpp (TORCH_LIBRARY(diffpure_hip, ...) + TORCH_LIBRARY_IMPL for the CUDA =
HIP key; built into csrc/libdiffpure_torch.so by diffpure_amd/build.py); importing this module loads it.  The engines
themselves call `diffpure_amd.ops` (ctypes on the C ABI) - this is the binding for a DiffPure maintainer who wants the
kernels as ordinary torch operators inside the reference's own `nn.Module` code (BASELINE.json's north star:
"hand-written HIP C++ ops ... registered as torch extensions"), e.g. swapping `nn.Conv2d.forward` of a ResBlock for
`torch.ops.diffpure_hip.conv2d_nhwc`.  There is no CPU implementation behind the operators, on purpose.

    import diffpure_amd.torch_ops
    y = torch.ops.diffpure_hip.conv2d_nhwc(x_nhwc, packed_w, bias, n_out, 3)
    y, cols = torch.ops.diffpure_hip.conv2d_h2_stats(xh, wh, bias, n_out, 3)          # column statistics: explicit 2nd return
    stats = torch.ops.diffpure_hip.group_norm_stats_from_cols(cols, B, H * W, 32, 1e-5)
    xh2 = torch.ops.diffpure_hip.group_norm_silu(y, gamma, beta, 32, 1e-5, True, 2, stats)   # -> fp16 operand of the next conv

Operators (NHWC fp32 unless stated; contracts in include/diffpure_hip.h, schemas in csrc/torch_binding.cpp):
    conv2d_nhwc / conv2d_nhwc_stats      fp32-input MFMA implicit GEMM (+ the epilogue's column statistics)
    conv2d_h2 / conv2d_h2_stats          fp16-matrix-core convolution; operand "h2" (hi|lo) or "h1" (plain fp16), passes 0 = full
    group_norm_stats_from_cols           (mean, rstd) per (sample, group) from the column records, no pass over the tensor
    group_norm_silu                      GroupNorm (+SiLU) -> fp32 (out_fmt 0) or the zero-bordered h2 (1) / h1 (2) operand
    attention                            softmax(q k^T / sqrt(d)) v, flash-style when d == 64
    em_step                              fused Euler-Maruyama update, in-kernel Philox noise
    resize_affine                        bilinear resize + affine + layout change
ABI 6 / 7 (rounds 4-5) - the fp16 residual stream and the backward entry points:
    conv2d_h2_ex                         the WHOLE epilogue contract: bias, time-embedding rows, fp32 / fp16 residual, scale, fp16 output,
                                         column records, 1x1 K-segments (a ResBlock's skip convolution folded into its second 3x3)
    gn_apply_h16                         GroupNorm-apply (+FiLM) (+SiLU) (+2x resample) over plain fp16 tensors -> "h1" operand / fp16 tensor
    attention_fused                      flash-style attention on fp32 qkv (three passes) or fp16 qkv (one pass, Q / K in place), optionally
                                         writing the zero-bordered fp16 operand of proj_out
    round_weights                        fp32 masters -> fp16 panels in place (nearest / stochastic, Philox-keyed): precision "f16sr"
    group_norm_stats, group_norm_silu_bwd, attention_bwd, resize_affine_bwd, conv2d_nhwc_dgrad      backward entry points (dL/dx)
ABI 8 (round 6):
    conv2d_stem                          the 3 -> N stem convolution as a write-bound kernel (22-bit operands, three fp16 MFMA passes)
torch.autograd: conv2d_nhwc, group_norm_silu (fp32 output form), attention and resize_affine are registered for the Autograd key with
dL/dx formulas built from those entry points (weights are constants on this path): `y.backward()` / `torch.autograd.grad` work through them.
"""
import os

import torch

from . import _lib

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libdiffpure_torch.so")
OPERATORS = ("conv2d_nhwc", "conv2d_nhwc_stats", "conv2d_h2", "conv2d_h2_stats", "group_norm_stats_from_cols", "group_norm_silu",
             "attention", "em_step", "resize_affine",
             "conv2d_h2_ex", "gn_apply_h16", "attention_fused", "round_weights", "group_norm_stats", "group_norm_silu_bwd", "attention_bwd",
             "resize_affine_bwd", "conv2d_nhwc_dgrad", "conv2d_stem")

if not os.path.exists(LIB_PATH):
    raise _lib.DiffpureHipError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
_lib.load()                        # libdiffpure_hip.so (and torch's HIP runtime) first: the extension links against it
torch.ops.load_library(LIB_PATH)
