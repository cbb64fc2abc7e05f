"""Tensor-level operators of the purification engine: thin wrappers that validate shapes, allocate
outputs with torch's caching allocator and launch libdiffpure_hip.so kernels on torch's current
HIP stream.  PyTorch is plumbing here (device memory + streams); every FLOP runs in the HIP library.

Layout: activations are NHWC fp32 `[B, H, W, C]`; "rows" tensors are `[M, K]`.
No CPU path exists: tensors must live on a GPU and the library must be built.
"""
import ctypes
import os
import math

import torch

from . import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(t, name, ndim=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.DiffpureHipError(f"{name}: expected a GPU tensor (the HIP engine has no CPU fallback)")
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise _lib.DiffpureHipError(f"{name}: expected contiguous float32, got {t.dtype} contiguous={t.is_contiguous()}")
    if ndim is not None and t.dim() != ndim:
        raise _lib.DiffpureHipError(f"{name}: expected {ndim} dims, got {tuple(t.shape)}")
    return t


def _ptr(t):
    return None if t is None else t.data_ptr()


# ---------------------------------------------------------------------------------------------
# weight packing (host-side, once at load)
# ---------------------------------------------------------------------------------------------
def _pad_cols(w2d):
    k, n = w2d.shape
    ldw = (n + 3) // 4 * 4
    if ldw == n:
        return w2d.contiguous()
    out = w2d.new_zeros(k, ldw)
    out[:, :n] = w2d
    return out


def pack_conv_weight(w):
    """OIHW (or OI for 1x1 / OIk for Conv1d k=1) -> [KH*KW*I, ldw] with k = (ky*KW+kx)*I + ci."""
    if w.dim() == 2:
        w = w[:, :, None, None]
    elif w.dim() == 3:
        w = w[:, :, :, None]
    o, i, kh, kw = w.shape
    return _pad_cols(w.permute(2, 3, 1, 0).reshape(kh * kw * i, o).float())


def pack_linear_weight(w):
    """nn.Linear weight [out, in] -> [in, ldw]."""
    return _pad_cols(w.t().float())


def pack_nin_weight(w):
    """score_sde NIN.W [in, out] -> [in, ldw]."""
    return _pad_cols(w.float())


def pack_h2(t):
    """fp32 GPU tensor [rows, cols] (cols % 8 == 0) -> "h2" split-fp16 tensor [rows, 2*cols] (fp16):
    per 8 columns, 8 x hi then 8 x lo with hi = fp16(v), lo = fp16(v - hi)."""
    _chk(t, "pack_h2.t", 2)
    rows, cols = t.shape
    out = torch.empty((rows, 2 * cols), device=t.device, dtype=torch.float16)
    _lib.call("dp_pack_h2", _ptr(t), rows, cols, cols, _ptr(out), _stream())
    return out


def pack_conv_weight_h2(w, device):
    """OIHW / OI / OIk weight -> h2 panel [N, 2*K] on `device` in the reduction order of
    csrc/igemm_h2.hip: k' = (c32 * KH*KW + tap) * 32 + ci % 32 (32-channel slices outermost, the
    taps innermost).  I % 32 == 0."""
    if w.dim() == 2:
        w = w[:, :, None, None]
    elif w.dim() == 3:
        w = w[:, :, :, None]
    o, i, kh, kw = w.shape
    assert i % 32 == 0, i
    wk = w.float().reshape(o, i // 32, 32, kh, kw).permute(0, 1, 3, 4, 2).reshape(o, kh * kw * i)
    return pack_h2(wk.contiguous().to(device))


def order_conv_weight_h2(w):
    """OIHW / OI / OIk weight -> fp32 panel [N, K] in the reduction order of csrc/igemm_h2.hip (host side)."""
    if w.dim() == 2:
        w = w[:, :, None, None]
    elif w.dim() == 3:
        w = w[:, :, :, None]
    o, i, kh, kw = w.shape
    assert i % 32 == 0, i
    return w.float().reshape(o, i // 32, 32, kh, kw).permute(0, 1, 3, 4, 2).reshape(o, kh * kw * i).contiguous()


def order_conv_weight_w16(w):
    """OIHW / OI / OIk weight -> the PLAIN-fp16 weight panel layout (w_fmt 1) as an fp32 tensor [N32, K] (N32 = N rounded up
    to 32 rows, zero rows appended), to be rounded to fp16 element by element.  Flat element order: 32-row blocks outermost,
    then the 8-element k' groups of the reduction order, then the 32 rows, then the 8 elements -
        index(n, k') = (((n / 32) * (K / 8) + k' / 8) * 32 + n % 32) * 8 + k' % 8
    so that the (lane-ordered) B fragment of one 32x32x16 MFMA - 32 rows x 2 groups - is 1 KB of contiguous memory (a wave loads
    it with one coalesced 16-byte-per-lane instruction, igemm_h2_sw.hip) and a 64-byte LDS row of a 32-channel k-tile is four
    16-byte pieces 512 bytes apart (the LDS-DMA loaders gather them: every lane has its own address anyway)."""
    wk = w.t if isinstance(w, PreOrdered) else order_conv_weight_h2(w)   # [N, K] in reduction order
    n, k = wk.shape
    n32 = (n + 31) // 32 * 32
    if n32 != n:
        wk = torch.cat([wk, wk.new_zeros(n32 - n, k)], dim=0)
    return wk.reshape(n32 // 32, 32, k // 8, 8).permute(0, 2, 1, 3).reshape(n32, k).contiguous()


class PreOrdered:
    """an fp32 weight matrix [N, K] that is ALREADY in the reduction order of the kernels (fuse_skip_weight)"""
    __slots__ = ("t",)

    def __init__(self, t):
        self.t = t


def fuse_skip_weight(w3, ws):
    """[N, KS*KS*C | C_skip] in reduction order: the k' order of the KS x KS convolution `w3`, then the input channels of the
    1x1 skip convolution `ws` (OI / OI11; its channels in their own order - the 1x1 K-segments of conv2d_h2 run over the raw
    input tensors channel by channel).  -> PreOrdered, which order_conv_weight_w16 / WeightPool.add take as already ordered."""
    if ws.dim() == 4:
        ws = ws[:, :, 0, 0]
    elif ws.dim() == 3:
        ws = ws[:, :, 0]
    assert ws.shape[1] % 32 == 0 and ws.shape[0] == w3.shape[0], (ws.shape, w3.shape)
    return PreOrdered(torch.cat([order_conv_weight_h2(w3.detach()), ws.detach().float()], dim=1).contiguous())


def unorder_conv_weight_w16(panel, n_out):
    """inverse of order_conv_weight_w16: [N32, K] panel -> [n_out, K] in plain reduction order (tests / host checks)"""
    n32, k = panel.shape
    return panel.reshape(n32 // 32, k // 8, 32, 8).permute(0, 2, 1, 3).reshape(n32, k)[:n_out]


class PoolSlot:
    """placeholder of a panel registered with a WeightPool until finalize() hands out the views"""
    __slots__ = ("name",)

    def __init__(self, name):
        self.name = name


class WeightPool:
    """Every plain-fp16 convolution panel of one network in ONE flat buffer, next to its fp32 masters (same layout), so
    that (re-)rounding all weights is one kernel launch per network call: `round(key)`.  stochastic=False ("f16"):
    round to nearest, done once; stochastic=True ("f16sr"): unbiased stochastic rounding keyed by (seed, key) - the
    solver passes the step index, so the rounding error of the weights differs from step to step and averages out over
    the loop instead of accumulating as a fixed perturbation of the model (tests/probes/sr_weights_probe.py:
    purified pixels 2.2e-4 instead of 1.0e-3 from the reference over the 100-step 256^2 loop)."""

    def __init__(self, device, stochastic, seed=0x5EEDC0DE):
        self.device, self.stochastic, self.seed = torch.device(device), bool(stochastic), int(seed)
        self._pending, self.master, self.work, self._views, self._last_key = [], None, None, {}, None
        # round 6, verdict item 1b - MEASURED AND OFF BY DEFAULT (DIFFPURE_ROUND_PREFETCH=1 switches it on): the NEXT call's rounding on a
        # side stream into a second fp16 buffer while this call's convolutions read the first (see round()).  Same-box A/B
        # (profiles/r06/*_prefetch_ab.log): headline 21.06 -> 21.02 images/s at t = 20 (-0.2 %), CIFAR B = 256 312.0 -> 305.0 (-2.3 %), CIFAR
        # adjoint 147.0 -> 143.4 (-2.5 %): the HBM-bound rounding kernel does not hide under the convolutions - it takes CUs from the
        # launches on the critical path (at CIFAR sizes every launch is short of workgroups already) and the chip is at its power cap
        # under the big ones, so the 0.56 % of the step it costs in-stream is the cheaper form.
        self._other, self._views_other, self._bound, self._pre, self._side = None, {}, [], None, None
        self._prefetch = (self.stochastic and self.device.type == "cuda" and os.environ.get("DIFFPURE_ROUND_PREFETCH", "0") == "1"
                          and os.environ.get("DIFFPURE_GRAPH", "0") == "0")       # (a captured graph holds the panel addresses)

    def add(self, name, w):
        """register the OIHW / OI / OIk weight `w` under `name`; the panel view is available after finalize()"""
        self._pending.append((name, order_conv_weight_w16(w if isinstance(w, PreOrdered) else w.detach())))

    def finalize(self):
        total = sum(p.numel() for _, p in self._pending)
        pad = (-total) % 8
        self.master = torch.empty(total + pad, dtype=torch.float32, device=self.device)
        self.work = torch.empty(total + pad, dtype=torch.float16, device=self.device)
        if self._prefetch and total:
            self._other = torch.empty_like(self.work)
        if pad:
            self.master[total:].zero_()
        off = 0
        for name, p in self._pending:
            n = p.numel()
            self.master[off:off + n].copy_(p.reshape(-1))       # host -> its slice of the flat buffer directly (no staging tensor + device copy)
            self._views[name] = self.work[off:off + n].view(p.shape)
            if self._other is not None:
                self._views_other[name] = self._other[off:off + n].view(p.shape)
            off += n
        self._pending = []
        self.round(0)
        return self

    def view(self, name):
        return self._views[name]

    def bind(self, table, key, name):
        """table[key] = the panel `name` of the CURRENT buffer, now and after every buffer flip of round() (the engines' parameter dicts)"""
        table[key] = self._views[name]
        self._bound.append((table, key, name))

    def _flip(self):
        self.work, self._other = self._other, self.work
        self._views, self._views_other = self._views_other, self._views
        for table, key, name in self._bound:
            table[key] = self._views[name]

    def _launch(self, dst, key):
        _lib.call("dp_round_weights", _ptr(self.master), _ptr(dst), self.master.numel(), 1 if self.stochastic else 0, self.seed, int(key), _stream())

    def round(self, key):
        """master fp32 -> working fp16 panels.  Round-to-nearest pools are rounded once (key ignored afterwards).
        Stochastic pools (round 6): the loops ask for keys k, k + 1, ... (or descending: the adjoints); once two consecutive keys have
        been seen, the rounding of the NEXT key is launched on a side stream into the second buffer right away - behind everything
        already queued on the main stream, i.e. the previous call's convolutions, which were that buffer's last readers - and runs
        under this call's convolutions (HBM-bound work under MFMA-bound work); the next round() then only waits for its event and
        flips the buffers (the bound parameter-dict entries are re-pointed).  Same bits as the in-stream form: the rounding is a
        pure function of (master, seed, key).  A key that was not predicted is rounded in-stream as before."""
        if not self.stochastic and self._last_key is not None:
            return
        if self.master is None or self.master.numel() == 0:
            return
        if self._pre is not None and self._pre[0] == key and self._other is not None:
            torch.cuda.current_stream(self.device).wait_event(self._pre[1])
            self._flip()
        else:
            if self._pre is not None and self._other is not None:      # an unused prefetch may still be writing the other buffer: harmless, but
                torch.cuda.current_stream(self.device).wait_event(self._pre[1])      # order it before anything that could flip onto it
            self._launch(self.work, key)
        self._pre = None
        stride = None if self._last_key is None else key - self._last_key
        self._last_key = key
        if self._other is not None and stride in (1, -1) and not prof_enabled():
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.device)
            main = torch.cuda.current_stream(self.device)
            self._side.wait_stream(main)            # the other buffer's last readers (the previous call) are queued on the main stream
            with torch.cuda.stream(self._side):
                self._launch(self._other, key + stride)
                ev = torch.cuda.Event()
                ev.record(self._side)
            self._pre = (key + stride, ev)


def _chk_h2(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.float16 or not t.is_contiguous():
        raise _lib.DiffpureHipError(f"{name}: expected a contiguous fp16 (h2 split format) GPU tensor")
    return t


# ---------------------------------------------------------------------------------------------
# convolution / linear
# ---------------------------------------------------------------------------------------------
class ColStats:
    """Per-column (sum, sumsq) partials a convolution epilogue left behind for the GroupNorm that
    consumes its output: `buf` [tiles][2][N] for tiles of `tile_rows` output rows."""
    __slots__ = ("buf", "tile_rows", "n")

    def __init__(self, buf, tile_rows, n):
        self.buf, self.tile_rows, self.n = buf, tile_rows, n


class Act:
    """An activation tensor `t` together with the column-statistics records `cols` its producing convolution left
    behind (conv2d / conv2d_h2 with colstats=True return one).  The pair travels EXPLICITLY through the engines: the
    records are not an attribute of the tensor, so no view / clone / dispatcher hop can silently drop them."""
    __slots__ = ("t", "cols")

    def __init__(self, t, cols=None):
        self.t, self.cols = t, cols

    @property
    def shape(self):
        return self.t.shape


class Deferred(Act):
    """The output of a split-K convolution (a <= 64-pixel level) whose reduction and epilogue have NOT run yet: the partial sums and the
    epilogue's arguments, waiting for their consumer.  A GroupNorm that the fused block boundary serves (group_norm_deferred,
    csrc/boundary.hip) reduces, finishes, normalises and emits the next convolution's operand in ONE launch and leaves the stream tensor
    (+ its column records) behind as `t` / `cols`; any other consumer calls resolve() - the plain reduction + epilogue launch - first.
    Either way the object IS an Act afterwards."""
    __slots__ = ("ws", "parts", "bias", "temb", "ts", "res", "scale", "out_f16", "dims")

    def __init__(self, ws, parts, bias, temb, ts, res, scale, out_f16, dims):
        super().__init__(None, None)
        self.ws, self.parts, self.bias, self.temb, self.ts, self.res, self.scale, self.out_f16, self.dims = ws, parts, bias, temb, ts, res, scale, out_f16, dims

    @property
    def shape(self):
        return torch.Size(self.dims)

    @property
    def dtype(self):
        return torch.float16 if self.out_f16 else torch.float32

    @property
    def device(self):
        return self.ws.device

    @property
    def resolved(self):
        return self.t is not None

    def resolve(self):
        """the plain reduction + epilogue (what conv2d_h2 runs behind its partial sums) -> self (an Act with tensor and column records)"""
        if self.t is None:
            b, h, w, n = self.dims
            out = torch.empty(self.dims, device=self.ws.device, dtype=self.dtype)
            cs, tr = _colstats_alloc(b * h * w, n, self.ws.device)
            _lib.call("dp_splitk_epilogue", _ptr(self.ws), self.parts, b, h, w, n, _ptr(self.bias), _ptr(self.temb), self.ts, _ptr(self.res),
                      0 if self.res is None or self.res.dtype != torch.float16 else 1, float(self.scale), _ptr(out), 1 if self.out_f16 else 0,
                      _ptr(cs), ctypes.addressof(tr), _stream())
            self.t, self.cols = out, ColStats(cs, tr.value, n)
            self.ws = None
        return self


def tensor_of(x):
    """Act | Tensor | None -> Tensor | None  (a Deferred convolution output is finished first)"""
    if isinstance(x, Deferred):
        return x.resolve().t
    return x.t if isinstance(x, Act) else x


def resolved(x):
    """Act | Tensor | None, with a Deferred convolution output finished"""
    return x.resolve() if isinstance(x, Deferred) else x


def dtype_of(x):
    return x.dtype if isinstance(x, Deferred) else tensor_of(x).dtype


def _colstats_alloc(m, n, device):
    """[records][2][n] with one record per 64 output rows.  A tile writes ALL the records of its rows (2 / 4 / 8 per tile of
    128 / 256 / 512 rows), also those that lie wholly beyond a ragged M: the buffer is rounded up to whole 512-row tiles and
    the tail beyond ceil(M / 64) is zeroed (round 1 allocated ceil(M / 64) records: a ragged M wrote past the end)."""
    rec, rec_pad = (m + 63) // 64, (m + 511) // 512 * 8
    buf = torch.empty((rec_pad, 2, n), device=device, dtype=torch.float32)
    if rec_pad > rec:
        buf[rec:].zero_()
    return buf, ctypes.c_int(0)


# precision name -> (MFMA passes per product, activation operand format of the fp16-matrix-core convolutions)
#   operand format 1 = "h2": per 8 channels [8 x fp16 hi | 8 x fp16 lo]  (4 bytes per element, 22 significant bits)
#                  2 = "h1": plain fp16                                   (2 bytes per element)
# Weights are always h2.  See include/diffpure_hip.h (dp_conv2d_nhwc_h2) for the arithmetic of each mode.
# "f16" / "f16sr" additionally keep the WEIGHTS as plain fp16 (w_fmt 1): rounded to nearest once at load ("f16"), or
# re-rounded stochastically from the fp32 masters before every network call ("f16sr": WeightPool.round).
H2_MODES = {"f16x3": (3, 1), "f16x2": (2, 2), "f16": (1, 2), "f16sr": (1, 2), "f16x2w": (12, 1)}
W16_MODES = ("f16", "f16sr")
FMT_F32, FMT_H2, FMT_H1 = 0, 1, 2


def _fmt_of(split):
    """group_norm / to_h2 `split` argument -> output format code: False / None / 0 -> fp32, True / 'h2' / 1 -> h2, 'h1' / 2 -> h1."""
    if split is True or split == "h2":
        return FMT_H2
    if split is None or split is False:
        return FMT_F32
    if split == "h1":
        return FMT_H1
    if split in (FMT_F32, FMT_H2, FMT_H1):
        return int(split)
    raise ValueError(f"unknown operand format {split!r}")


def conv_defers(h, w, ksize, c, n_out):
    """is this LAYER reduced with split-K, so that its reduction + epilogue can be left to the fused block boundary?  (<= 64 pixels per
    sample; a function of the layer shape only.)  DIFFPURE_BOUNDARY=0 switches the fused boundaries off."""
    return os.environ.get("DIFFPURE_BOUNDARY", "1") != "0" and bool(_lib.load().dp_conv2d_nhwc_h2_splits_by_shape(h, w, ksize, c, n_out))


def conv2d_h2(x, wh, n_out, ksize, bias=None, temb=None, res=None, scale=1.0, colstats=False, passes=None, w_fmt=0, out_f16=False,
              segs=None, defer=False):
    """conv2d on the fp16 matrix cores; same epilogue contract as conv2d.  wh: [N, 2*K] fp16 (h2, pack_conv_weight_h2),
    or with w_fmt=1 the plain fp16 panel [N32, K] in the block layout of order_conv_weight_w16 (WeightPool; one pass, h1
    activations).
    x: zero-bordered operand as group_norm(split=...) writes it - h2 [B, H+2, W+2, 2*C] fp16 (three MFMA passes per
    product, `passes` 3, or 12 for the weights-rounded study mode) or h1 [B, H+2, W+2, C] fp16 (`passes` 2 or 1).
    The activation format is read off the shapes; `passes` defaults to the full arithmetic of the formats (3 / 2 / 1).
    out_f16=True: the result is stored as PLAIN fp16 [B, H, W, N] (rounded to nearest; the column statistics are those of
    the unrounded values).  res: fp32 or plain fp16 [B, H, W, N] (the fp16 residual stream).
    segs=(s1,) | (s1, s2): 1x1 K-segments - plain fp16 NHWC tensors [B, H, W, Cs] whose weight columns follow the KS x KS
    part in `wh` (fuse_skip_weight): out += cat(s1, s2) . wh[:, KS*KS*C:] - a ResBlock's 1x1 skip folded into this
    convolution.  Only where takes_segments(...) says so.
    defer=True (fp16 x fp16 only, where conv_defers(...) says so): only the split-K partial sums are formed; -> Deferred."""
    _chk_h2(x, "conv2d_h2.x")
    _chk_h2(wh, "conv2d_h2.w")
    b, h, w = x.shape[0], x.shape[1] - 2, x.shape[2] - 2
    sc = [0, 0]
    sp = [None, None]
    if segs:
        assert w_fmt == 1 and 1 <= len(segs) <= 2
        for i, sg in enumerate(segs):
            if not (isinstance(sg, torch.Tensor) and sg.is_cuda and sg.dtype == torch.float16 and sg.is_contiguous() and sg.dim() == 4
                    and tuple(sg.shape[:3]) == (b, h, w)):
                raise _lib.DiffpureHipError(f"conv2d_h2: K-segment {i} must be a contiguous fp16 GPU tensor [B, H, W, Cs] at the output resolution")
            sc[i], sp[i] = sg.shape[3], sg.data_ptr()
    kk = wh.shape[1] // (1 if w_fmt else 2) - sc[0] - sc[1]
    c = kk // (ksize * ksize)
    if x.shape[3] == 2 * c:
        a_fmt = 0
    elif x.shape[3] == c:
        a_fmt = 1
    else:
        raise _lib.DiffpureHipError(f"conv2d_h2: operand {tuple(x.shape)} does not match the weight panel {tuple(wh.shape)} (ksize {ksize})")
    if passes is None:
        passes = 1 if w_fmt else (2 if a_fmt else 3)
    if w_fmt:
        assert wh.shape == ((n_out + 31) // 32 * 32, ksize * ksize * c + sc[0] + sc[1]), (wh.shape, n_out, ksize, c, sc)
    else:
        assert wh.shape == (n_out, 2 * ksize * ksize * c), (wh.shape, n_out, ksize, c)
    if bias is not None:
        _chk(bias, "conv2d_h2.bias", 1)
    ts = 0
    if temb is not None:
        assert temb.is_cuda and temb.dtype == torch.float32 and temb.dim() == 2 and temb.stride(1) == 1
        assert temb.shape[0] in (1, b) and temb.shape[1] >= n_out
        ts = 0 if temb.shape[0] == 1 else temb.stride(0)
    ldr, rfmt = 0, 0
    if res is not None:
        if res.dtype == torch.float16:
            if not (res.is_cuda and res.is_contiguous() and res.dim() == 4):
                raise _lib.DiffpureHipError("conv2d_h2.res: expected a contiguous fp16 GPU tensor")
            rfmt = 1
        else:
            _chk(res, "conv2d_h2.res", 4)
        assert tuple(res.shape) == (b, h, w, n_out), (res.shape, (b, h, w, n_out))
        ldr = n_out
    if defer:
        if not (w_fmt == 1 and a_fmt == 1 and conv_defers(h, w, ksize, c, n_out)):
            raise _lib.DiffpureHipError("conv2d_h2: defer=True needs an fp16 x fp16 split-K layer (conv_defers)")
        wbytes = int(_lib.load().dp_conv2d_nhwc_h2_workspace(b, h, w, ksize, c, n_out))
        work = torch.empty((wbytes // 4,), device=x.device, dtype=torch.float32)
        parts = ctypes.c_int(0)
        _lib.call("dp_conv2d_nhwc_h2_partials", _ptr(x), c, b, h, w, ksize, _ptr(wh), n_out, _ptr(work), wbytes, int(passes), a_fmt, int(w_fmt),
                  sp[0], sc[0], sp[1], sc[1], ctypes.addressof(parts), _stream())
        return Deferred(work, parts.value, bias, temb, ts, res, scale, bool(out_f16), (b, h, w, n_out))
    out = torch.empty((b, h, w, n_out), device=x.device, dtype=torch.float16 if out_f16 else torch.float32)
    cs, tr = _colstats_alloc(b * h * w, n_out, x.device) if colstats else (None, None)
    # low-resolution levels are reduced with split-K (factor fixed by the layer shape): scratch for the partial sums
    wbytes = int(_lib.load().dp_conv2d_nhwc_h2_workspace(b, h, w, ksize, c, n_out))
    work = torch.empty((wbytes // 4,), device=x.device, dtype=torch.float32) if wbytes else None
    _lib.call("dp_conv2d_nhwc_h2", _ptr(x), c, b, h, w, ksize, _ptr(wh), n_out, _ptr(bias), _ptr(temb), ts, _ptr(res),
              ldr, float(scale), _ptr(out), n_out, _ptr(cs), None if tr is None else ctypes.addressof(tr), _ptr(work), wbytes,
              int(passes), a_fmt, int(w_fmt), 1 if out_f16 else 0, rfmt, sp[0], sc[0], sp[1], sc[1], _stream())
    return Act(out, ColStats(cs, tr.value, n_out)) if colstats else out


def takes_segments(h, w, ksize, c, n_out, c1, c2=0):
    """may an fp16 x fp16 convolution of this LAYER shape carry 1x1 K-segments of c1 (+ c2) channels?  (a function of the layer, never
    of the batch: fused or not, any sharding of a batch takes the same arithmetic)"""
    return bool(_lib.load().dp_conv2d_nhwc_h2_takes_segments(h, w, ksize, c, n_out, c1, c2))


def conv2d(x, wp, n_out, ksize, bias=None, x2=None, temb=None, res=None, scale=1.0, out=None, colstats=False, out_f16=False):
    """out = scale * (res + bias + temb[b] + conv_{ksize x ksize, same}(cat(x, x2)))   (NHWC).
    colstats=True: the epilogue also reduces per-column partial sums and the call returns Act(out, records), which
    `group_norm_stats` turns into GroupNorm statistics without reading the tensor again.
    out_f16=True: the result is stored as plain fp16 (the stem of a network whose residual stream is fp16)."""
    _chk(x, "conv2d.x", 4)
    b, h, w, c1 = x.shape
    c2 = 0
    if x2 is not None:
        _chk(x2, "conv2d.x2", 4)
        assert x2.shape[:3] == x.shape[:3], (x.shape, x2.shape)
        c2 = x2.shape[3]
    _chk(wp, "conv2d.w", 2)
    assert wp.shape[0] == ksize * ksize * (c1 + c2), (wp.shape, ksize, c1, c2)
    assert wp.shape[1] >= n_out
    if bias is not None:
        _chk(bias, "conv2d.bias", 1)
    ts = 0
    if temb is not None:
        # temb: [R, >=n_out] rows, R in {1, B}; may be a column view of a wider table (row stride kept)
        assert temb.is_cuda and temb.dtype == torch.float32 and temb.dim() == 2 and temb.stride(1) == 1
        assert temb.shape[0] in (1, b) and temb.shape[1] >= n_out
        ts = 0 if temb.shape[0] == 1 else temb.stride(0)
    if out is None:
        out = torch.empty((b, h, w, n_out), device=x.device, dtype=torch.float16 if out_f16 else torch.float32)
    assert out.dtype == (torch.float16 if out_f16 else torch.float32)
    ldr = 0
    if res is not None:
        _chk(res, "conv2d.res", 4)
        assert res.shape == out.shape, (res.shape, out.shape)
        ldr = n_out
    cs, tr = _colstats_alloc(b * h * w, n_out, x.device) if colstats else (None, None)
    _lib.call("dp_conv2d_nhwc", _ptr(x), c1, _ptr(x2), c2, b, h, w, ksize, ksize, _ptr(wp), wp.shape[1], n_out,
              _ptr(bias), _ptr(temb), ts, _ptr(res), ldr, float(scale), _ptr(out), n_out, 1 if out_f16 else 0, _ptr(cs),
              None if tr is None else ctypes.addressof(tr), _stream())
    return Act(out, ColStats(cs, tr.value, n_out)) if colstats else out


def pack_stem_weight(w):
    """OIHW [N, 3, 3, 3] stem weight -> the (hi | lo) panel of dp_conv2d_stem: [2, N, 32] fp16, k = (ky*3 + kx)*Cin + ci, columns beyond
    9*Cin zero; hi = fp16(w), lo = fp16(w - hi) (host side, once at load)."""
    n, cin, kh, kw = w.shape
    assert kh == 3 and kw == 3 and 9 * cin <= 32, w.shape
    wk = torch.zeros(n, 32, dtype=torch.float32)
    wk[:, :9 * cin] = w.detach().float().permute(0, 2, 3, 1).reshape(n, 9 * cin)
    hi = wk.half()
    lo = (wk - hi.float()).half()
    return torch.stack([hi, lo], dim=0).contiguous()


def conv2d_stem_ok(cin, b, h, w, n_out):
    """does csrc/stem.hip serve this stem shape?  (a function of the layer; the batch enters only through B*H*W % 64, which every
    power-of-two image size satisfies at any B)"""
    return bool(_lib.load().dp_conv2d_stem_ok(cin, b, h, w, n_out)) and os.environ.get("DIFFPURE_STEM16", "1") != "0"


def conv2d_stem(x, wpanel, n_out, bias=None, colstats=False, out_f16=False):
    """3x3 'same' convolution of the fp32 NHWC state x [B, H, W, 3] with the (hi | lo) panel of pack_stem_weight -> [B, H, W, n_out]
    fp32 / plain fp16 (+ the column records), on the write-bound stem kernel (22-bit operands, three fp16 MFMA passes, fp32 accumulation)."""
    _chk(x, "conv2d_stem.x", 4)
    b, h, w, cin = x.shape
    if not (isinstance(wpanel, torch.Tensor) and wpanel.is_cuda and wpanel.dtype == torch.float16 and wpanel.is_contiguous()
            and tuple(wpanel.shape) == (2, n_out, 32)):
        raise _lib.DiffpureHipError("conv2d_stem.w: expected the [2, N, 32] fp16 panel of pack_stem_weight on the GPU")
    if bias is not None:
        _chk(bias, "conv2d_stem.bias", 1)
    out = torch.empty((b, h, w, n_out), device=x.device, dtype=torch.float16 if out_f16 else torch.float32)
    cs, tr = _colstats_alloc(b * h * w, n_out, x.device) if colstats else (None, None)
    _lib.call("dp_conv2d_stem", _ptr(x), cin, b, h, w, _ptr(wpanel), n_out, _ptr(bias), _ptr(out), 1 if out_f16 else 0, _ptr(cs),
              None if tr is None else ctypes.addressof(tr), _stream())
    return Act(out, ColStats(cs, tr.value, n_out)) if colstats else out


def linear(x, wp, n_out, bias=None):
    """[M, K] @ [K, N] + bias."""
    _chk(x, "linear.x", 2)
    m, k = x.shape
    return conv2d(x.view(m, 1, 1, k), wp, n_out, 1, bias=bias).view(m, n_out)


# ---------------------------------------------------------------------------------------------
# GroupNorm (+FiLM) (+SiLU) (+resample)
# ---------------------------------------------------------------------------------------------
RESAMPLE_NONE, RESAMPLE_UP, RESAMPLE_DOWN = 0, 1, 2
RESAMPLE_FIR_UP, RESAMPLE_FIR_DOWN = 3, 4        # score_sde `fir: True`: upfirdn2d with a separable 4-tap filter


def _out_hw(h, w, mode):
    if mode in (RESAMPLE_UP, RESAMPLE_FIR_UP):
        return h * 2, w * 2
    if mode in (RESAMPLE_DOWN, RESAMPLE_FIR_DOWN):
        return h // 2, w // 2
    return h, w


def fir_taps(kernel):
    """score_sde `fir_kernel` (e.g. [1, 3, 3, 1]) -> the 4 taps normalised to sum 1, as _setup_kernel does per axis
    (up_or_down_sampling.py:189-200)."""
    k = [float(v) for v in kernel]
    if len(k) != 4:
        raise NotImplementedError(f"FIR resampling is built for 4-tap separable filters, got {kernel}")
    s = sum(k)
    return tuple(v / s for v in k)


def _fir_arg(mode, fir):
    if mode not in (RESAMPLE_FIR_UP, RESAMPLE_FIR_DOWN):
        return None, None
    if fir is None or len(fir) != 4:
        raise _lib.DiffpureHipError("FIR resampling needs the 4 filter taps (ops.fir_taps(fir_kernel))")
    arr = (ctypes.c_float * 4)(*[float(v) for v in fir])
    return arr, ctypes.addressof(arr)


def _nsplit(hw):
    # pixel slabs per sample for the statistics pass. A function of the image size ONLY: the slab
    # partition fixes the summation order, and results must not depend on how a batch is sharded.
    return max(1, min(hw // 256, 128))


def group_norm_stats(x, groups, eps, x2=None):
    """-> stats [B, G, 2] = (mean, rstd) of cat(x, x2) per (sample, group).  x / x2: tensors, or `Act` pairs whose
    column records (from the producing convolutions' epilogues) make the pass over the data unnecessary.  fp16 tensors
    (a convolution's fp16 output, the fp16 residual stream) exist ONLY with their records: their statistics are those of the
    unrounded values the epilogue summed."""
    x, x2 = resolved(x), resolved(x2)          # (an unfinished split-K convolution output is finished first: its records come with it)
    k1 = x.cols if isinstance(x, Act) else None
    k2 = x2.cols if isinstance(x2, Act) else None
    x, x2 = tensor_of(x), tensor_of(x2)
    half = x.dtype == torch.float16
    if half:
        ok = x.is_cuda and x.dim() == 4 and k1 is not None and (x.shape[1] * x.shape[2]) % k1.tile_rows == 0
        if x2 is not None:
            ok = ok and x2.dtype == torch.float16 and k2 is not None and (x.shape[1] * x.shape[2]) % k2.tile_rows == 0
        if not ok:
            raise _lib.DiffpureHipError("gn.x: an fp16 activation needs the column records of its producing convolution (both sources)")
    else:
        _chk(x, "gn.x", 4)
    b, h, w, c1 = x.shape
    c2 = 0 if x2 is None else (x2.shape[3] if half else _chk(x2, "gn.x2", 4).shape[3])
    hw = h * w
    stats = torch.empty((b, groups, 2), device=x.device, dtype=torch.float32)
    s = _stream()
    if k1 is not None and hw % k1.tile_rows == 0 and (x2 is None or (k2 is not None and hw % k2.tile_rows == 0)):
        # the producing convolutions already reduced this tensor per column: no pass over the data
        _lib.call("dp_gn_finalize_cols", _ptr(k1.buf), c1, k1.tile_rows, None if k2 is None else _ptr(k2.buf), c2,
                  0 if k2 is None else k2.tile_rows, b, hw, groups, float(eps), _ptr(stats), s)
        return stats
    ns = _nsplit(hw)
    partial = torch.empty((b, ns, groups, 2), device=x.device, dtype=torch.float32)
    _lib.call("dp_gn_stats", _ptr(x), c1, _ptr(x2), c2, b, hw, groups, ns, _ptr(partial), s)
    _lib.call("dp_gn_finalize", _ptr(partial), b, ns, groups, hw * ((c1 + c2) // groups), float(eps), _ptr(stats), s)
    return stats


def splitk_gn_ok(h, w, n, c2, groups):
    """does the fused block boundary serve (convolution output [.., h, w, n], second source of c2 channels, `groups` groups)?"""
    return bool(_lib.load().dp_splitk_gn_ok(h, w, n, c2, groups))


def deferred_fusable(xa, x2a, groups, resample=RESAMPLE_NONE, fir=None):
    """can the GroupNorm over cat(xa, x2a) be the fused block boundary of the unfinished convolution output xa?"""
    if not (isinstance(xa, Deferred) and not xa.resolved) or resample != RESAMPLE_NONE:
        return False
    x2 = tensor_of(x2a)
    b, h, w, n = xa.dims
    if x2 is not None and not (x2.is_cuda and x2.is_contiguous() and x2.dim() == 4 and tuple(x2.shape[:3]) == (b, h, w)
                               and x2.dtype in (torch.float16, torch.float32)):
        return False
    return splitk_gn_ok(h, w, n, 0 if x2 is None else x2.shape[3], groups)


def group_norm_deferred(d, groups, eps, gamma, beta, x2=None, film=None, act=False, raw=False, want_out=True, want_stats=False):
    """The fused block boundary (csrc/boundary.hip): finishes the split-K convolution output `d` (Deferred) and applies the GroupNorm
    (+FiLM) (+SiLU) of cat(d, x2) in ONE launch.  -> (y, stats | None, y_raw | None): y the zero-bordered "h1" operand
    [B, H+2, W+2, N + C2]; with want_out `d` becomes a finished Act (stream tensor, + column records at 64 pixels per sample)."""
    assert isinstance(d, Deferred) and not d.resolved
    b, h, w, n = d.dims
    x2 = tensor_of(x2)
    c2 = 0 if x2 is None else x2.shape[3]
    c = n + c2
    dev = d.ws.device
    fs, fh, fstride = _film_args(film, b, c)
    out = torch.empty(d.dims, device=dev, dtype=d.dtype) if want_out else None
    cs = tr = None
    if want_out and h * w == 64:
        cs, tr = _colstats_alloc(b * 64, n, dev)
    stats = torch.empty((b, groups, 2), device=dev, dtype=torch.float32) if want_stats else None
    y = torch.empty((b, h + 2, w + 2, c), device=dev, dtype=torch.float16)
    yr = torch.empty_like(y) if raw else None
    _lib.call("dp_splitk_gn", _ptr(d.ws), d.parts, b, h, w, n, _ptr(d.bias), _ptr(d.temb), d.ts, _ptr(d.res),
              0 if d.res is None or d.res.dtype != torch.float16 else 1, float(d.scale), _ptr(out), 1 if d.out_f16 else 0, _ptr(cs), _ptr(x2),
              0 if x2 is None or x2.dtype != torch.float16 else 1, c2, groups, float(eps), _ptr(gamma), _ptr(beta), _ptr(fs), _ptr(fh), fstride,
              1 if act else 0, _ptr(stats), _ptr(y), _ptr(yr), _stream())
    if want_out:
        d.t, d.cols = out, (ColStats(cs, 64, n) if cs is not None else None)
    d.ws = None
    return y, stats, yr


def _film_args(film, b, c):
    if film is None:
        return None, None, 0
    fs, fh = film
    assert fs.is_cuda and fs.dtype == torch.float32 and fs.shape[-1] == c and fs.stride(-1) == 1
    assert fh.shape == fs.shape and fh.stride() == fs.stride()
    assert fs.shape[0] in (1, b)
    return fs, fh, (0 if fs.shape[0] == 1 else fs.stride(0))


def _chk_f16(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.float16 or not t.is_contiguous() or t.dim() != 4:
        raise _lib.DiffpureHipError(f"{name}: expected a contiguous fp16 GPU tensor [B, H, W, C]")
    return t


def group_norm(x, groups, eps, gamma, beta, x2=None, film=None, act=False, resample=RESAMPLE_NONE, stats=None,
               split=False, raw=False, fir=None):
    """y = resample(act(FiLM(GroupNorm(cat(x, x2))))).  film = (scale [R,C], shift [R,C]) with R in
    {1, B}; the two may be column views of one [R, 2C] tensor (row stride is taken from them).
    split=True / "h2" writes the split-fp16 operand format of conv2d_h2 with its one-pixel zero border
    ([B, Ho+2, Wo+2, 2C] fp16), split="h1" the plain-fp16 operand ([B, Ho+2, Wo+2, C] fp16).  raw=True (with
    split, no resampling) additionally returns the un-normalised cat(x, x2) in the same operand format (input of
    a 1x1 skip convolution); fp16 inputs only: raw=True WITH 2x resampling additionally returns the resampled un-normalised
    cat(x, x2) as a plain fp16 tensor [B, Ho, Wo, C] (the identity skip of an up / down ResBlock; a K-segment of its second convolution).
    x (and x2) may be PLAIN fp16 tensors [B, H, W, C] (a convolution's fp16 output / the fp16 residual stream): then `stats`
    is required, split must be "h1", and the pass moves 4 instead of 6 bytes per element (dp_gn_apply_h16)."""
    if isinstance(x, torch.Tensor) and x.dtype == torch.float16:
        if _fmt_of(split) != FMT_H1 or stats is None:
            raise _lib.DiffpureHipError("group_norm: an fp16 input goes to the 'h1' operand format and needs its statistics")
        return _group_norm_h16(x, groups, gamma, beta, stats, x2, film, act, resample, 2, raw)
    _chk(x, "gn.x", 4)
    b, h, w, c1 = x.shape
    c2 = 0 if x2 is None else x2.shape[3]
    c = c1 + c2
    if stats is None:
        stats = group_norm_stats(x, groups, eps, x2)
    fs, fh, fstride = _film_args(film, b, c)
    ho, wo = _out_hw(h, w, resample)
    fmt = _fmt_of(split)
    fkeep, fptr = _fir_arg(resample, fir)
    if fmt:
        y = torch.empty((b, ho + 2, wo + 2, (2 * c) if fmt == FMT_H2 else c), device=x.device, dtype=torch.float16)
    else:
        y = torch.empty((b, ho, wo, c), device=x.device, dtype=torch.float32)
    yr = None
    if raw:
        assert fmt and resample == RESAMPLE_NONE
        yr = torch.empty_like(y)
    _lib.call("dp_gn_apply", _ptr(x), c1, _ptr(x2), c2, b, h, w, groups, _ptr(stats), _ptr(gamma), _ptr(beta),
              _ptr(fs), _ptr(fh), fstride, 1 if act else 0, resample, fmt, _ptr(y), _ptr(yr), fptr, _stream())
    return (y, yr) if raw else y


def _group_norm_h16(x, groups, gamma, beta, stats, x2, film, act, resample, out_fmt, raw=False):
    """dp_gn_apply_h16: fp16 [B, H, W, C1 (+ C2)] -> out_fmt 2: the zero-bordered 'h1' operand; 3: a plain fp16 tensor"""
    _chk_f16(x, "gn.x16")
    b, h, w, c1 = x.shape
    c2 = 0
    if x2 is not None:
        _chk_f16(x2, "gn.x2_16")
        assert x2.shape[:3] == x.shape[:3], (x.shape, x2.shape)
        c2 = x2.shape[3]
    c = c1 + c2
    if resample not in (RESAMPLE_NONE, RESAMPLE_UP, RESAMPLE_DOWN):
        raise _lib.DiffpureHipError("group_norm: the FIR resampling modes run on the fp32 residual stream")
    fs, fh, fstride = _film_args(film, b, c)
    ho, wo = _out_hw(h, w, resample)
    y = torch.empty((b, ho + 2, wo + 2, c) if out_fmt == 2 else (b, ho, wo, c), device=x.device, dtype=torch.float16)
    yr = None
    if raw:
        assert out_fmt == 2
        yr = torch.empty_like(y) if resample == RESAMPLE_NONE else torch.empty((b, ho, wo, c), device=x.device, dtype=torch.float16)
    _lib.call("dp_gn_apply_h16", _ptr(x), c1, _ptr(x2), c2, b, h, w, groups, _ptr(stats), _ptr(gamma), _ptr(beta), _ptr(fs), _ptr(fh),
              fstride, 1 if act else 0, resample, out_fmt, _ptr(y), _ptr(yr), _stream())
    return (y, yr) if raw else y


def group_norm_f16in(x16, groups, gamma, beta, stats, film=None, act=False):
    """GroupNorm-apply (+FiLM) (+SiLU) of a tensor its producing convolution stored as plain fp16 (conv2d_h2 out_f16=True):
    x16 [B, H, W, C] fp16 -> the zero-bordered "h1" operand [B, H+2, W+2, C] fp16 of the next convolution.  Same values as
    group_norm(x16.float(), ..., split="h1") at 4 instead of 6 HBM bytes per element."""
    return _group_norm_h16(x16, groups, gamma, beta, stats, None, film, act, RESAMPLE_NONE, 2)


def group_norm_bwd(x, groups, gamma, beta, stats, dy, x2=None, film=None, act=False, resample=RESAMPLE_NONE, split=False,
                   fir=None, addend=None, addend2=None, addend_scale=1.0, one_pass=None):
    """Input gradient of `group_norm` (same arguments - incl. the FIR taps for resample modes 3 / 4 -, `stats` from the forward,
    dy at the forward's output resolution). -> (dx, dx2); with split=True (single source) dx is the zero-bordered h2
    operand for the next dgrad convolution.
    addend / addend2 (fp32 output only): a second gradient arriving at the same tensors (the skip branch of a ResBlock): dx += addend_scale * addend,
    dx2 += addend_scale * addend2 - inside the one-pass kernel where the shape fits one workgroup per channel block (small feature maps), else
    inside the apply pass of the three-launch form.  one_pass=False forces the three-launch form (tests, probes)."""
    # x / x2 (the forward's input as the tape holds it): fp32, or plain fp16 where the taped forward ran on the fp16 residual stream
    x16 = isinstance(x, torch.Tensor) and x.dtype == torch.float16
    if x16:
        _chk_f16(x, "gn_bwd.x16")
        if x2 is not None:
            _chk_f16(x2, "gn_bwd.x2_16")
    else:
        _chk(x, "gn_bwd.x", 4)
        if x2 is not None:
            _chk(x2, "gn_bwd.x2", 4)
    _chk(dy, "gn_bwd.dy", 4)
    b, h, w, c1 = x.shape
    c2 = 0 if x2 is None else x2.shape[3]
    xfmt = 1 if x16 else 0
    c = c1 + c2
    fs = fh = None
    fstride = 0
    if film is not None:
        fs, fh = film
        assert fs.shape[0] in (1, b) and fs.shape[-1] == c and fs.stride(-1) == 1 and fh.stride() == fs.stride()
        fstride = 0 if fs.shape[0] == 1 else fs.stride(0)
    ho, wo = _out_hw(h, w, resample)
    assert dy.shape == (b, ho, wo, c), (dy.shape, (b, ho, wo, c))
    ofmt = 0 if not split else (2 if split == "h1" else 1)     # split: True / "h2" -> h2 operand, "h1" -> plain fp16 operand
    if ofmt:
        assert x2 is None and addend is None and addend2 is None
        dx = torch.empty((b, h + 2, w + 2, c if ofmt == 2 else 2 * c), device=x.device, dtype=torch.float16)
        dx2 = None
    else:
        dx = torch.empty(x.shape, device=x.device, dtype=torch.float32)
        dx2 = None if x2 is None else torch.empty(x2.shape, device=x.device, dtype=torch.float32)
        if addend is not None:
            _chk(addend, "gn_bwd.addend", 4)
            assert addend.shape == x.shape
        if addend2 is not None:
            assert x2 is not None and _chk(addend2, "gn_bwd.addend2", 4).shape == x2.shape
    s = _stream()
    if one_pass is not False and resample <= RESAMPLE_DOWN and gn_bwd_fused_ok(h, w, c1, c2, groups, resample):
        _lib.call("dp_gn_bwd_fused", _ptr(x), c1, _ptr(x2), c2, xfmt, b, h, w, groups, _ptr(stats), _ptr(gamma), _ptr(beta), _ptr(fs), _ptr(fh),
                  fstride, 1 if act else 0, resample, _ptr(dy), ofmt, _ptr(dx), _ptr(dx2), _ptr(addend), _ptr(addend2), float(addend_scale), s)
        return dx, dx2
    fkeep, fptr = _fir_arg(resample, fir)
    ns = _nsplit(h * w)
    partial = torch.empty((b, ns, groups, 2), device=x.device, dtype=torch.float32)
    sums = torch.empty((b, groups, 2), device=x.device, dtype=torch.float32)
    common = (_ptr(x), c1, _ptr(x2), c2, xfmt, b, h, w, groups, _ptr(stats), _ptr(gamma), _ptr(beta), _ptr(fs), _ptr(fh),
              fstride, 1 if act else 0, resample, fptr, _ptr(dy))
    _lib.call("dp_gn_bwd_stats", *common, ns, _ptr(partial), _ptr(sums), s)
    # the skip branch's gradient joins dx inside the apply pass (round 5; rounds 3-4 ran an `add` launch behind it: 6.6 % of the ImageNet
    # adjoint step)
    _lib.call("dp_gn_bwd_apply", *common, _ptr(sums), ofmt, _ptr(dx), _ptr(dx2), _ptr(addend), _ptr(addend2), float(addend_scale), s)
    return dx, dx2


def gn_bwd_fused_ok(h, w, c1, c2, groups, resample=RESAMPLE_NONE):
    """does the one-pass GroupNorm backward serve this tensor shape?  (a function of the shape only)"""
    return bool(_lib.load().dp_gn_bwd_fused_ok(h, w, c1, c2, groups, resample))


def resample_bwd(dy, mode, fir=None):
    """Adjoint of `resample(x, mode, fir)`; dy at the forward's output resolution."""
    _chk(dy, "resample_bwd.dy", 4)
    b, ho, wo, c = dy.shape
    h, w = (ho // 2, wo // 2) if mode in (RESAMPLE_UP, RESAMPLE_FIR_UP) else (ho * 2, wo * 2)
    dx = torch.empty((b, h, w, c), device=dy.device, dtype=torch.float32)
    fkeep, fptr = _fir_arg(mode, fir)
    _lib.call("dp_resample_bwd", _ptr(dy), b, ho, wo, c, mode, fptr, _ptr(dx), _stream())
    return dx


def add(a, b):
    _chk(a, "add.a")
    _chk(b, "add.b")
    assert a.shape == b.shape
    out = torch.empty_like(a)
    _lib.call("dp_add", _ptr(a), _ptr(b), _ptr(out), a.numel(), _stream())
    return out


def _img_dims(shape, nhwc):
    return (shape[0], shape[3], shape[1], shape[2]) if nhwc else tuple(shape)


def resize_affine(x, size, shift, scale, in_nhwc=False, out_nhwc=False):
    """y = (F.interpolate(x, size, mode='bilinear', align_corners=False) + shift) * scale with a free choice
    of layouts - the steps either side of the purifier in SDE_Adv_Model.forward (eval_sde_adv.py:74-89)."""
    _chk(x, "resize_affine.x", 4)
    b, c, hi, wi = _img_dims(x.shape, in_nhwc)
    ho, wo = size
    y = torch.empty((b, ho, wo, c) if out_nhwc else (b, c, ho, wo), device=x.device, dtype=torch.float32)
    _lib.call("dp_resize_affine", _ptr(x), b, c, hi, wi, int(in_nhwc), float(shift), float(scale), _ptr(y), ho, wo, int(out_nhwc),
              _stream())
    return y


def resize_affine_bwd(dy, in_size, scale, in_nhwc=False, out_nhwc=False):
    """Adjoint of `resize_affine`: dy in the forward's output layout -> dx [.., in_size ..] in its input layout."""
    _chk(dy, "resize_affine_bwd.dy", 4)
    b, c, ho, wo = _img_dims(dy.shape, out_nhwc)
    hi, wi = in_size
    dx = torch.empty((b, hi, wi, c) if in_nhwc else (b, c, hi, wi), device=dy.device, dtype=torch.float32)
    _lib.call("dp_resize_affine_bwd", _ptr(dy), b, c, ho, wo, int(out_nhwc), float(scale), _ptr(dx), hi, wi, int(in_nhwc), _stream())
    return dx


def to_h2(x, mode=RESAMPLE_NONE, fmt="h2", fir=None):
    """fp32 NHWC -> zero-bordered convolution operand without normalisation, optionally through the 2x resampler
    (`mode`): fmt "h2" -> [B, H'+2, W'+2, 2C] fp16 (hi|lo octets), "h1" -> [B, H'+2, W'+2, C] plain fp16.  A plain fp16
    input (the fp16 residual stream) goes to "h1"."""
    if isinstance(x, torch.Tensor) and x.dtype == torch.float16:
        if _fmt_of(fmt) != FMT_H1:
            raise _lib.DiffpureHipError("to_h2: an fp16 input goes to the 'h1' operand format")
        return _group_norm_h16(x, 1, None, None, None, None, None, False, mode, 2)
    _chk(x, "to_h2.x", 4)
    b, h, w, c = x.shape
    f = _fmt_of(fmt)
    assert f, fmt
    ho, wo = _out_hw(h, w, mode)
    fkeep, fptr = _fir_arg(mode, fir)
    y = torch.empty((b, ho + 2, wo + 2, (2 * c) if f == FMT_H2 else c), device=x.device, dtype=torch.float16)
    _lib.call("dp_gn_apply", _ptr(x), c, None, 0, b, h, w, 1, None, None, None, None, None, 0, 0, mode, f, _ptr(y), None, fptr, _stream())
    return y


def dgrad_weight(w):
    """Weight of the input-gradient convolution: for y = conv(x, W) (stride 1, 'same'),
    dx = conv(dy, Wd) with Wd[i, o, ky, kx] = W[o, i, KH-1-ky, KW-1-kx]."""
    if w.dim() == 2:
        w = w[:, :, None, None]
    elif w.dim() == 3:
        w = w[:, :, :, None]
    return w.flip(2, 3).transpose(0, 1).contiguous()


def resample(x, mode, fir=None):
    """Nearest x2 up (mode 1), 2x2 mean down (2) or the FIR x2 up / down of `fir: True` networks (3 / 4, taps `fir`) of an
    NHWC tensor, no normalisation.  A plain fp16 tensor (the fp16 residual stream) stays fp16."""
    if isinstance(x, torch.Tensor) and x.dtype == torch.float16:
        return _group_norm_h16(x, 1, None, None, None, None, None, False, mode, 3)
    _chk(x, "resample.x", 4)
    b, h, w, c = x.shape
    ho, wo = _out_hw(h, w, mode)
    fkeep, fptr = _fir_arg(mode, fir)
    y = torch.empty((b, ho, wo, c), device=x.device, dtype=torch.float32)
    _lib.call("dp_gn_apply", _ptr(x), c, None, 0, b, h, w, 1, None, None, None, None, None, 0, 0, mode, 0, _ptr(y), None, fptr, _stream())
    return y


# ---------------------------------------------------------------------------------------------
# attention core
# ---------------------------------------------------------------------------------------------
def _attn_offsets(c, d, layout):
    if layout == "legacy":
        return 0, d, 2 * d, 3 * d
    if layout == "split":
        return 0, c, 2 * c, d
    raise ValueError(layout)


def attention_fused_ok(t, d):
    """shapes the flash-style kernel covers (head dimension 64 with whole 64-token blocks, or 256 with whole 128-token blocks);
    DIFFPURE_ATTN_FUSED=0 disables it"""
    return ((d == 64 and t % 64 == 0) or (d == 256 and t % 128 == 0)) and os.environ.get("DIFFPURE_ATTN_FUSED", "1") != "0"


_BORDERED = {}


def _bordered_f16(shape, device):
    """A zero-bordered fp16 operand buffer [B, H+2, W+2, C] whose border was zeroed ONCE, reused by every launch of the same shape on
    the same (device, stream, host thread): the attention kernel writes only the interior, the 1x1 convolution behind it is the only
    reader and is queued on the same stream before the next writer (round 4 re-zeroed a fresh tensor per attention block: 16 ATen fill
    launches per guided UNet call).  Keyed by the host thread as well, so that two engines driven from two threads on one stream never
    share a buffer.  The tensor is therefore valid only until the next attention_fused(operand_hw=...) call of the same shape on this
    stream and thread: the engines consume it at once (the proj_out / NIN_3 convolution); a caller that wants to keep it clones it
    (torch.ops.diffpure_hip.attention_fused allocates a fresh one per call)."""
    import threading
    if torch.cuda.is_current_stream_capturing():
        # under graph capture (DIFFPURE_GRAPH=1) a cached buffer would either live in this graph's private pool - and be handed, un-zeroed,
        # to later captures - or be dropped by clear() while a captured graph still addresses it: a capture gets its own tensor and its own
        # memset node (round 6, advisor)
        return torch.zeros(shape, device=device, dtype=torch.float16)
    key = (device.index, _stream(), threading.get_ident(), tuple(shape))
    buf = _BORDERED.get(key)
    if buf is None:
        if len(_BORDERED) > 64:      # shapes of a few networks / batch sizes; never grows without bound
            _BORDERED.clear()
        buf = _BORDERED[key] = torch.zeros(shape, device=device, dtype=torch.float16)
    return buf


def attention_fused(qkv, n_heads, layout, operand_hw=None):
    """softmax(q k^T / sqrt(d)) v without materialising the scores (csrc/attention.hip).  operand_hw=(H, W) (H * W tokens):
    the result is written as the zero-bordered fp16 operand [B, H+2, W+2, C] of the following 1x1 convolution
    (conv2d_h2, "h1" format) instead of fp32 [B, T, C].
    qkv fp32: split-fp16 operands, three MFMA passes (fp32-class accuracy).  qkv fp16 (the qkv convolution's out_f16): ONE fp16 pass,
    Q and K read in place - the fp16 x fp16 precision modes."""
    assert qkv.is_cuda and qkv.is_contiguous() and qkv.dim() == 3 and qkv.dtype in (torch.float32, torch.float16), "attention_fused.qkv"
    b, t, c3 = qkv.shape
    c = c3 // 3
    f16 = qkv.dtype == torch.float16
    work = torch.empty((b * t * c,), device=qkv.device, dtype=torch.float16) if f16 else torch.empty((3 * b * t * c,), device=qkv.device, dtype=torch.float32)
    lay = 0 if layout == "legacy" else 1
    if operand_hw is not None:
        hh, ww = operand_hw
        assert hh * ww == t, (operand_hw, t)
        out = _bordered_f16((b, hh + 2, ww + 2, c), qkv.device)       # the border is zero and stays zero: the kernel writes the interior only
        _lib.call("dp_attention_fused", _ptr(qkv), 1 if f16 else 0, b, t, c, n_heads, lay, _ptr(out), 1, ww, _ptr(work), _stream())
        return out
    out = torch.empty((b, t, c), device=qkv.device, dtype=torch.float32)
    _lib.call("dp_attention_fused", _ptr(qkv), 1 if f16 else 0, b, t, c, n_heads, lay, _ptr(out), 0, 0, _ptr(work), _stream())
    return out


def _gemm(h16, a, a16, a_args, bm, b16, b_args, c_args, m, n, k, zb, zh, alpha, s):
    """one strided batched GEMM: on the fp16 matrix cores (dp_gemm_strided_h16: operands rounded to fp16 on their way into LDS - or read in
    place where they ARE fp16 (a16 / b16) -, fp32 accumulation) where `h16` asks for it and the shape is one that kernel serves, else
    the fp32-input MFMA kernel (fp32 operands only).  a_args / b_args = (ld, batch stride, head stride, trans) in elements."""
    if h16 and _h16_ok(m, n, k):
        _lib.call("dp_gemm_strided_h16", a, 1 if a16 else 0, *a_args, bm, 1 if b16 else 0, *b_args, *c_args, m, n, k, zb, zh, float(alpha), s)
        return
    if a16 or b16:
        raise _lib.DiffpureHipError("attention: an fp16 qkv needs every product on dp_gemm_strided_h16 (attention_h16_serves); up-convert it")
    _lib.call("dp_gemm_strided", a, *a_args, bm, *b_args, *c_args, m, n, k, zb, zh, float(alpha), s)


def _h16_ok(m, n, k):
    """dp_gemm_strided_h16 serves the shape (DIFFPURE_H16_N64=0: only its 128 x 128 tile - round 5's coverage, for A/B runs)"""
    return bool(_lib.load().dp_gemm_strided_h16_ok(m, n, k)) and (n % 128 == 0 or os.environ.get("DIFFPURE_H16_N64", "1") != "0")


def attention_h16_serves(t, d):
    """do all five products of the attention backward (q k^T, dV, dP, dQ, dK) have shapes dp_gemm_strided_h16 serves?  Then the taped
    fp16 qkv is read in place; else (head dimension 64: N = 64 in dV / dQ / dK) the caller hands in an fp32 qkv."""
    return bool(_h16_ok(t, t, d) and _h16_ok(t, d, t))


def attention(qkv, n_heads, layout, return_probs=False, probs_only=False, h16=False):
    """softmax(q k^T / sqrt(d)) v for qkv [B, T, 3C] -> [B, T, C]  (, probs [B*heads, T, T]).
    layout 'legacy': channels = heads x [q(d) | k(d) | v(d)]   (QKVAttentionLegacy, unet.py:345-362)
    layout 'split' : channels = [Q(all heads) | K | V]         (QKVAttention unet.py:377-397; NCSN++ q,k,v NINs)
    probs_only=True: only the probabilities are wanted (the backward pass recomputes them from the taped qkv): the P V product is
    skipped and (None, probs) returned.  h16=True (the fp16 x fp16 precision modes' gradient path): the products run on the fp16
    matrix cores where dp_gemm_strided_h16 serves the shape; qkv may then be the plain fp16 tensor the tape holds (attention_h16_serves)."""
    q16 = isinstance(qkv, torch.Tensor) and qkv.dtype == torch.float16
    if q16:
        assert h16 and qkv.is_cuda and qkv.is_contiguous() and qkv.dim() == 3, "attention: an fp16 qkv takes the h16 path"
    else:
        _chk(qkv, "attention.qkv", 3)
    b, t, c3 = qkv.shape
    c = c3 // 3
    d = c // n_heads
    if not return_probs and not probs_only and attention_fused_ok(t, d):
        return attention_fused(qkv, n_heads, layout)
    oq, ok, ov, sh = _attn_offsets(c, d, layout)
    s = _stream()
    scores = torch.empty((b * n_heads, t, t), device=qkv.device, dtype=torch.float32)
    base = qkv.data_ptr()
    el = 2 if q16 else 4
    # scores[z] = (1/sqrt(d)) * Q K^T
    _gemm(h16, base + oq * el, q16, (c3, t * c3, sh, 0), base + ok * el, q16, (c3, t * c3, sh, 1),
          (_ptr(scores), t, n_heads * t * t, t * t), t, t, d, b, n_heads, 1.0 / math.sqrt(d), s)
    _lib.call("dp_softmax_rows", _ptr(scores), b * n_heads * t, t, s)
    if probs_only:
        return None, scores
    out = torch.empty((b, t, c), device=qkv.device, dtype=torch.float32)
    # out[z] = P V
    _gemm(h16, _ptr(scores), False, (t, n_heads * t * t, t * t, 0), base + ov * el, q16, (c3, t * c3, sh, 0),
          (_ptr(out), c, t * c, d), t, d, t, b, n_heads, 1.0, s)
    return (out, scores) if return_probs else out


def attention_bwd(qkv, probs, dout, n_heads, layout, h16=False):
    """Gradient of `attention` w.r.t. qkv: dV = P^T dO, dP = dO V^T, dS = softmax'(P, dP),
    dQ = dS K / sqrt(d), dK = dS^T Q / sqrt(d), written into dqkv [B, T, 3C] (fp32) in the same layout.
    h16=True: the four products on the fp16 matrix cores (see `attention`; qkv may be the taped fp16 tensor); softmax' stays fp32."""
    q16 = isinstance(qkv, torch.Tensor) and qkv.dtype == torch.float16
    if q16:
        assert h16 and qkv.is_cuda and qkv.is_contiguous() and qkv.dim() == 3, "attention_bwd: an fp16 qkv takes the h16 path"
    else:
        _chk(qkv, "attention_bwd.qkv", 3)
    _chk(probs, "attention_bwd.probs", 3)
    _chk(dout, "attention_bwd.dout", 3)
    b, t, c3 = qkv.shape
    c = c3 // 3
    d = c // n_heads
    oq, ok, ov, sh = _attn_offsets(c, d, layout)
    s = _stream()
    el = 2 if q16 else 4
    dqkv = torch.empty(qkv.shape, device=qkv.device, dtype=torch.float32)
    dp = torch.empty_like(probs)
    q0, g0, do0 = qkv.data_ptr(), dqkv.data_ptr(), dout.data_ptr()
    sc = 1.0 / math.sqrt(d)
    zb, zh = n_heads * t * t, t * t
    # dV[s][c] = sum_t P[t][s] dO[t][c]                 (A = P stored [K=t][M=s] -> transA)
    _gemm(h16, _ptr(probs), False, (t, zb, zh, 1), do0, False, (c, t * c, d, 0), (g0 + ov * 4, c3, t * c3, sh), t, d, t, b, n_heads, 1.0, s)
    # dP[t][s] = sum_c dO[t][c] V[s][c]                 (B = V stored [N=s][K=c] -> transB)
    _gemm(h16, do0, False, (c, t * c, d, 0), q0 + ov * el, q16, (c3, t * c3, sh, 1), (_ptr(dp), t, zb, zh), t, t, d, b, n_heads, 1.0, s)
    _lib.call("dp_softmax_bwd_rows", _ptr(probs), _ptr(dp), b * n_heads * t, t, s)
    # dQ[t][c] = sc * sum_s dS[t][s] K[s][c]
    _gemm(h16, _ptr(dp), False, (t, zb, zh, 0), q0 + ok * el, q16, (c3, t * c3, sh, 0), (g0 + oq * 4, c3, t * c3, sh), t, d, t, b, n_heads, sc, s)
    # dK[s][c] = sc * sum_t dS[t][s] Q[t][c]            (A = dS stored [K=t][M=s] -> transA)
    _gemm(h16, _ptr(dp), False, (t, zb, zh, 1), q0 + oq * el, q16, (c3, t * c3, sh, 0), (g0 + ok * 4, c3, t * c3, sh), t, d, t, b, n_heads, sc, s)
    return dqkv


# ---------------------------------------------------------------------------------------------
# small pieces
# ---------------------------------------------------------------------------------------------
def silu(x):
    _chk(x, "silu.x")
    y = torch.empty_like(x)
    _lib.call("dp_silu", _ptr(x), _ptr(y), x.numel(), _stream())
    return y


def axpby(x, a, y, b):
    _chk(x, "axpby.x")
    _chk(y, "axpby.y")
    assert x.shape == y.shape
    out = torch.empty_like(x)
    _lib.call("dp_axpby", _ptr(x), float(a), _ptr(y), float(b), _ptr(out), x.numel(), _stream())
    return out


def timestep_embedding(t, freqs, cos_first):
    _chk(t, "temb.t", 1)
    _chk(freqs, "temb.freqs", 1)
    n, half = t.shape[0], freqs.shape[0]
    emb = torch.empty((n, 2 * half), device=t.device, dtype=torch.float32)
    _lib.call("dp_timestep_embedding", _ptr(t), n, _ptr(freqs), half, 1 if cos_first else 0, _ptr(emb), _stream())
    return emb


def philox_normal(shape, seed, sample0, step, device):
    """Standard normals [B, ...] keyed by (seed, sample0 + b, step, element)."""
    b = shape[0]
    per = 1
    for s_ in shape[1:]:
        per *= s_
    out = torch.empty(shape, device=device, dtype=torch.float32)
    _lib.call("dp_philox_normal", _ptr(out), b, per, int(seed), int(sample0), int(step), _stream())
    return out


def em_step(x, eps, neg_half_beta, gg, score_coef, score_div, h, g, sqrt_h, noise=None, seed=0, sample0=0, step=0,
            out=None):
    """x: [B,H,W,C] state; eps: [B,H,W,Ce] network output (first C channels used)."""
    _chk(x, "em_step.x", 4)
    _chk(eps, "em_step.eps", 4)
    b, hh, ww, c = x.shape
    assert eps.shape[:3] == x.shape[:3] and eps.shape[3] >= c
    if noise is not None:
        _chk(noise, "em_step.noise", 4)
        assert noise.shape == x.shape
    if out is None:
        out = torch.empty_like(x)
    _lib.call("dp_em_step", _ptr(x), _ptr(eps), eps.shape[3], b, hh * ww, c, float(neg_half_beta), float(gg),
              float(score_coef), 1 if score_div else 0, float(h), float(g), float(sqrt_h), _ptr(noise), int(seed),
              int(sample0), int(step), _ptr(out), _stream())
    return out


def ddpm_step(x, out6, sr, srm1, c1, c2, min_log, max_log, nonzero, noise=None, seed=0, sample0=0, step=0, out=None):
    _chk(x, "ddpm_step.x", 4)
    _chk(out6, "ddpm_step.out6", 4)
    b, hh, ww, c = x.shape
    assert out6.shape == (b, hh, ww, 2 * c)
    if noise is not None:
        _chk(noise, "ddpm_step.noise", 4)
    if out is None:
        out = torch.empty_like(x)
    _lib.call("dp_ddpm_step", _ptr(x), _ptr(out6), b, hh * ww, c, float(sr), float(srm1), float(c1), float(c2),
              float(min_log), float(max_log), 1 if nonzero else 0, _ptr(noise), int(seed), int(sample0), int(step),
              _ptr(out), _stream())
    return out


# ---------------------------------------------------------------------------------------------
# profiling hooks (bench.py roofline leg)
# ---------------------------------------------------------------------------------------------
_PROF_ON = False


def prof_enable(on=True):
    """on=True opens a new recording window (old records are discarded); on=False closes it, keeping the records."""
    global _PROF_ON
    _PROF_ON = bool(on)
    _lib.call("dp_prof_enable", 1 if on else 0)


def set_tuning(name, value):
    """Flip a kernel-variant switch of the library in-process (csrc/dp_tune.h; all variants give identical bits).  The library
    reads its DP_* environment variables once, at first use - this is the only way to change one afterwards."""
    _lib.call("dp_set_tuning", name.encode(), int(value))


def get_tuning(name):
    v = ctypes.c_int(0)
    _lib.call("dp_get_tuning", name.encode(), ctypes.addressof(v))
    return v.value


class tuning:
    """with ops.tuning(DP_H2_DW=2, DP_H2_SW=0): ...   (restores the previous values on exit)"""

    def __init__(self, **kw):
        self.kw, self.old = kw, {}

    def __enter__(self):
        for k, v in self.kw.items():
            self.old[k] = get_tuning(k)
            set_tuning(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_tuning(k, v)
        return False


def prof_enabled():
    return _PROF_ON


PROF_KINDS = ("pp3x3", "conv1x1", "other3x3", "pp1x1", "gn_apply", "dh3x3", "dh1x1")       # DP_PROF_* of include/diffpure_hip.h


def prof_collect():
    """-> {kind: dict(ms, n, flop, bytes)} for the launches of the last recording window, plus 'dropped' (launches beyond the
    record buffer: a non-zero value means the window was too long and the totals are a prefix of it)."""
    import ctypes as C
    k = len(PROF_KINDS)
    ms, n, fl, by = (C.c_double * k)(), (C.c_longlong * k)(), (C.c_double * k)(), (C.c_double * k)()
    dropped = C.c_longlong()
    _lib.call("dp_prof_collect", C.addressof(ms), C.addressof(n), C.addressof(fl), C.addressof(by), C.addressof(dropped))
    out = {name: dict(ms=ms[i], n=n[i], flop=fl[i], bytes=by[i]) for i, name in enumerate(PROF_KINDS)}
    out["dropped"] = dropped.value
    return out
