"""`torch.ops.diffpure_hip.*`: the hot operators registered with PyTorch's dispatcher.

The engines call `diffpure_amd.ops` (ctypes on the C ABI) directly; this module is for a DiffPure maintainer who
wants the kernels as ordinary torch operators inside the reference's own `nn.Module` code (BASELINE.json's north
star: "hand-written HIP C++ ops ... registered as torch extensions") - e.g. swapping `nn.Conv2d.forward` of a
ResBlock for `torch.ops.diffpure_hip.conv2d_nhwc`.  Registration is done from Python with `torch.library`, so there
is nothing to compile beyond `libdiffpure_hip.so`; the implementations are registered for the CUDA (= HIP) dispatch
key only - there is no CPU kernel behind them, on purpose.

    import diffpure_amd.torch_ops            # registers the namespace
    y = torch.ops.diffpure_hip.conv2d_nhwc(x_nhwc, packed_w, bias, n_out, 3)

Operators (all NHWC fp32 unless stated; see include/diffpure_hip.h for the contracts):
    conv2d_nhwc(x, wp, bias?, n_out, ksize)                       fp32-input MFMA implicit GEMM
    conv2d_h2(xh, wh, bias?, n_out, ksize)                        split-fp16 three-pass MFMA; xh / wh in "h2" form
    group_norm_silu(x, gamma, beta, groups, eps, act, to_h2)      GroupNorm (+SiLU) -> fp32 or zero-bordered h2 operand
    attention(qkv, n_heads, legacy_layout)                        softmax(q k^T / sqrt(d)) v, fused when d == 64
    em_step(x, eps, nhb, gg, sc, div, h, g, sqrt_h, seed, sample0, step)   fused Euler-Maruyama update, Philox noise
    resize_affine(x, ho, wo, shift, scale, in_nhwc, out_nhwc)     bilinear resize + affine + layout change
"""
import torch

from . import ops

_LIB = torch.library.Library("diffpure_hip", "DEF")
_LIB.define("conv2d_nhwc(Tensor x, Tensor wp, Tensor? bias, int n_out, int ksize) -> Tensor")
_LIB.define("conv2d_h2(Tensor xh, Tensor wh, Tensor? bias, int n_out, int ksize) -> Tensor")
_LIB.define("group_norm_silu(Tensor x, Tensor gamma, Tensor beta, int groups, float eps, bool act, bool to_h2) -> Tensor")
_LIB.define("attention(Tensor qkv, int n_heads, bool legacy_layout) -> Tensor")
_LIB.define("em_step(Tensor x, Tensor eps, float nhb, float gg, float sc, bool div, float h, float g, float sqrt_h, int seed, "
            "int sample0, int step) -> Tensor")
_LIB.define("resize_affine(Tensor x, int ho, int wo, float shift, float scale, bool in_nhwc, bool out_nhwc) -> Tensor")


def _conv2d_nhwc(x, wp, bias, n_out, ksize):
    return ops.conv2d(x, wp, n_out, ksize, bias=bias)


def _conv2d_h2(xh, wh, bias, n_out, ksize):
    return ops.conv2d_h2(xh, wh, n_out, ksize, bias=bias)


def _group_norm_silu(x, gamma, beta, groups, eps, act, to_h2):
    return ops.group_norm(x, groups, eps, gamma, beta, act=act, split=to_h2)


def _attention(qkv, n_heads, legacy_layout):
    return ops.attention(qkv, n_heads, "legacy" if legacy_layout else "split")


def _em_step(x, eps, nhb, gg, sc, div, h, g, sqrt_h, seed, sample0, step):
    return ops.em_step(x, eps, nhb, gg, sc, div, h, g, sqrt_h, seed=seed, sample0=sample0, step=step)


def _resize_affine(x, ho, wo, shift, scale, in_nhwc, out_nhwc):
    return ops.resize_affine(x, (ho, wo), shift, scale, in_nhwc, out_nhwc)


for _name, _fn in (("conv2d_nhwc", _conv2d_nhwc), ("conv2d_h2", _conv2d_h2), ("group_norm_silu", _group_norm_silu),
                   ("attention", _attention), ("em_step", _em_step), ("resize_affine", _resize_affine)):
    _LIB.impl(_name, _fn, "CUDA")

OPERATORS = ("conv2d_nhwc", "conv2d_h2", "group_norm_silu", "attention", "em_step", "resize_affine")
