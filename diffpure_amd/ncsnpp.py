"""MI355X engine for the score_sde NCSN++ score network (CIFAR-10 32x32, "ddpmpp deep").

Same function as score_sde.models.ncsnpp.NCSNpp.forward (/root/reference/score_sde/models/
ncsnpp.py:232-381, topology :139-230) for the configuration DiffPure ships (configs/cifar10.yml:
BigGAN ResBlocks, fir False, progressive none, positional embedding); loads the reference's
`all_modules.N.*` state_dict keys (and therefore the EMA parameter list in the same order).

  ResnetBlockBigGANpp (layerspp.py:242-274) = gn_stats -> gn_apply(+SiLU, +2x up/down)
        -> conv3x3(+bias, + Dense_0(SiLU(temb)) row per sample)
        -> gn_stats -> gn_apply(+SiLU) -> conv3x3(+bias, + [x | conv1x1(x)] residual, * 1/sqrt2)
  AttnBlockpp (layerspp.py:75-91) = gn -> ONE 1x1 GEMM for NIN_0|NIN_1|NIN_2 -> QK^T -> softmax
        -> PV -> NIN_3 GEMM (+residual, * 1/sqrt2)
  All Dense_0 layers of the 76 ResBlocks are one packed GEMM per forward.
"""
import math
from collections import OrderedDict

import functools
import os

import torch

from . import ops

INV_SQRT2 = 1.0 / math.sqrt(2.0)


def parse_config(cfg):
    """cfg: dict with 'model' and 'data' sections (configs/cifar10.yml)."""
    m, d = cfg["model"], cfg["data"]
    ok = (
        m["name"] == "ncsnpp" and m["resblock_type"].lower() == "biggan"
        and m["progressive"].lower() == "none" and m["progressive_input"].lower() == "none"
        and m["embedding_type"].lower() == "positional" and m["conditional"]
        and m["nonlinearity"].lower() == "swish" and not m["scale_by_sigma"] and d["centered"] and m["skip_rescale"]
    )
    if not ok:
        raise NotImplementedError("engine covers the DiffPure CIFAR-10 NCSN++ family (configs/cifar10.yml)")
    return dict(
        nf=int(m["nf"]),
        ch_mult=tuple(m["ch_mult"]),
        num_res_blocks=int(m["num_res_blocks"]),
        attn_resolutions=tuple(m["attn_resolutions"]),
        image_size=int(d["image_size"]),
        channels=int(d["num_channels"]),
        sigma_min=float(m["sigma_min"]),
        sigma_max=float(m["sigma_max"]),
        num_scales=int(m["num_scales"]),
        # fir: True -> the BigGAN blocks resample with upfirdn2d (layerspp.py:245-258) instead of nearest / mean
        fir=bool(m.get("fir", False)),
        fir_kernel=tuple(m.get("fir_kernel", (1, 3, 3, 1))),
    )


def _plan(cfg):
    nf, nrb, mults = cfg["nf"], cfg["num_res_blocks"], cfg["ch_mult"]
    nres = len(mults)
    res = [cfg["image_size"] // (2 ** i) for i in range(nres)]
    down, up = [], []
    i = 2
    stem = dict(kind="stem", idx=i, cin=cfg["channels"], cout=nf)
    i += 1
    hs_c = [nf]
    ch = nf
    for lvl in range(nres):
        for _ in range(nrb):
            co = nf * mults[lvl]
            blk = [dict(kind="res", idx=i, cin=ch, cout=co, mode=0)]
            i += 1
            ch = co
            if res[lvl] in cfg["attn_resolutions"]:
                blk.append(dict(kind="attn", idx=i, ch=ch))
                i += 1
            down.append(blk)
            hs_c.append(ch)
        if lvl != nres - 1:
            down.append([dict(kind="res", idx=i, cin=ch, cout=ch, mode=ops.RESAMPLE_DOWN)])
            i += 1
            hs_c.append(ch)
    mid = [dict(kind="res", idx=i, cin=ch, cout=ch, mode=0), dict(kind="attn", idx=i + 1, ch=ch),
           dict(kind="res", idx=i + 2, cin=ch, cout=ch, mode=0)]
    i += 3
    for lvl in reversed(range(nres)):
        for _ in range(nrb + 1):
            co = nf * mults[lvl]
            sk = hs_c.pop()
            up.append(dict(kind="res", idx=i, cin=ch + sk, cout=co, mode=0, pop=True, c1=ch))
            i += 1
            ch = co
        if res[lvl] in cfg["attn_resolutions"]:
            up.append(dict(kind="attn", idx=i, ch=ch))
            i += 1
        if lvl != 0:
            up.append(dict(kind="res", idx=i, cin=ch, cout=ch, mode=ops.RESAMPLE_UP))
            i += 1
    assert not hs_c
    return dict(stem=stem, down=down, mid=mid, up=up, gn_idx=i, conv_idx=i + 1, final_ch=ch)


def param_shapes(cfg):
    """state_dict key -> shape in all_modules order (= the EMA shadow-parameter order)."""
    nf = cfg["nf"]
    plan = _plan(cfg)
    sh = OrderedDict()
    sh["sigmas"] = (cfg["num_scales"],)
    M = "all_modules."
    sh[M + "0.weight"], sh[M + "0.bias"] = (4 * nf, nf), (4 * nf,)
    sh[M + "1.weight"], sh[M + "1.bias"] = (4 * nf, 4 * nf), (4 * nf,)
    st = plan["stem"]
    sh[M + f"{st['idx']}.weight"], sh[M + f"{st['idx']}.bias"] = (st["cout"], st["cin"], 3, 3), (st["cout"],)
    recs = [r for b in plan["down"] for r in b] + plan["mid"] + plan["up"]
    for r in recs:
        p = M + str(r["idx"])
        if r["kind"] == "res":
            ci, co = r["cin"], r["cout"]
            sh[p + ".GroupNorm_0.weight"], sh[p + ".GroupNorm_0.bias"] = (ci,), (ci,)
            sh[p + ".Conv_0.weight"], sh[p + ".Conv_0.bias"] = (co, ci, 3, 3), (co,)
            sh[p + ".Dense_0.weight"], sh[p + ".Dense_0.bias"] = (co, 4 * nf), (co,)
            sh[p + ".GroupNorm_1.weight"], sh[p + ".GroupNorm_1.bias"] = (co,), (co,)
            sh[p + ".Conv_1.weight"], sh[p + ".Conv_1.bias"] = (co, co, 3, 3), (co,)
            if ci != co or r["mode"]:
                sh[p + ".Conv_2.weight"], sh[p + ".Conv_2.bias"] = (co, ci, 1, 1), (co,)
        else:
            c = r["ch"]
            sh[p + ".GroupNorm_0.weight"], sh[p + ".GroupNorm_0.bias"] = (c,), (c,)
            for j in range(4):
                sh[p + f".NIN_{j}.W"], sh[p + f".NIN_{j}.b"] = (c, c), (c,)
    fc = plan["final_ch"]
    sh[M + f"{plan['gn_idx']}.weight"], sh[M + f"{plan['gn_idx']}.bias"] = (fc,), (fc,)
    sh[M + f"{plan['conv_idx']}.weight"] = (cfg["channels"], fc, 3, 3)
    sh[M + f"{plan['conv_idx']}.bias"] = (cfg["channels"],)
    return sh


class NCSNpp:
    """score network: NHWC in, NHWC out ([B, H, W, channels]); `labels` = 999*s float."""

    GN_EPS = 1e-6

    def __init__(self, cfg, device, precision="f32"):
        """precision: "f32" exact fp32-input MFMA; "f16x3" split-fp16 three-pass MFMA for every
        convolution fed by a GroupNorm (see GuidedUNet / csrc/igemm_h2.hip)."""
        if precision != "f32" and precision not in ops.H2_MODES:
            raise ValueError(f"unknown precision {precision!r}")
        self.cfg = cfg
        self.precision = precision
        # fp16-matrix-core convolution path: MFMA passes per product and the operand format GroupNorm-apply emits
        self._fir = ops.fir_taps(cfg["fir_kernel"]) if cfg.get("fir") else None
        self.h2mode = precision in ops.H2_MODES
        passes, ofmt = ops.H2_MODES.get(precision, (3, ops.FMT_H2))
        self._ofmt = "h1" if ofmt == ops.FMT_H1 else "h2"
        self._pool = ops.WeightPool(torch.device(device), stochastic=precision == "f16sr") if precision in ops.W16_MODES else None
        self._ch2 = functools.partial(ops.conv2d_h2, passes=passes, w_fmt=1 if self._pool is not None else 0)
        # fp16 x fp16 modes: the first convolution of a ResBlock stores its output as fp16 (see GuidedUNet)
        self._lean = self._pool is not None and os.environ.get("DIFFPURE_LEAN", "1") != "0"
        # round 6: fused block boundaries of the <= 64-pixel levels (csrc/boundary.hip; ops.Deferred) - fp16 x fp16 modes (the kernel emits
        # plain fp16 operands); DIFFPURE_BOUNDARY=0 restores the four-launch chain
        self._bfuse = self._pool is not None and self._ofmt == "h1"
        self._lean16 = False         # fp16 residual stream: decided in load_state_dict
        self._tape16 = os.environ.get("DIFFPURE_TAPE16", "1") != "0"     # round 5: the taped forward runs on it too (GuidedUNet._o16)
        self.device = torch.device(device)
        self.plan = _plan(cfg)
        self.p = {}
        half = cfg["nf"] // 2
        # layers.py:518-521 on the host, as the reference evaluates it
        e = math.log(10000) / (half - 1)
        self.freqs = torch.exp(torch.arange(half, dtype=torch.float32) * -e).to(self.device)
        self.dense_cols = 0

    @staticmethod
    def _groups(c):
        return min(c // 4, 32)

    def load_state_dict(self, sd):
        want = param_shapes(self.cfg)
        for k, shp in want.items():
            if k == "sigmas":
                continue
            if k not in sd:
                raise KeyError(f"state_dict is missing {k}")
            if tuple(sd[k].shape) != tuple(shp):
                raise ValueError(f"{k}: expected {shp}, got {tuple(sd[k].shape)}")
        dev = self.device
        P = {}
        M = "all_modules."
        self._sd = sd          # host copy for the lazily packed input-gradient panels (enable_grad)
        self._grad_ready = False

        def vec(k):
            return sd[k].detach().float().contiguous().to(dev)

        def conv_w(k, cin):
            if self.h2mode and cin % 32 == 0:
                return self._pack_h2w(sd[k].detach()), True
            return ops.pack_conv_weight(sd[k].detach()).to(dev), False

        P["t0.w"], P["t0.b"] = ops.pack_linear_weight(sd[M + "0.weight"].detach()).to(dev), vec(M + "0.bias")
        P["t1.w"], P["t1.b"] = ops.pack_linear_weight(sd[M + "1.weight"].detach()).to(dev), vec(M + "1.bias")
        si = self.plan["stem"]["idx"]
        P["stem.w"], P["stem.b"] = ops.pack_conv_weight(sd[M + f"{si}.weight"].detach()).to(dev), vec(M + f"{si}.bias")
        if self.h2mode and self.plan["stem"]["cin"] == 3:       # the write-bound stem kernel (csrc/stem.hip) where its shape test passes
            P["stem.w16"] = ops.pack_stem_weight(sd[M + f"{si}.weight"]).to(dev)
        dw, db, off = [], [], 0
        recs = [r for b in self.plan["down"] for r in b] + self.plan["mid"] + self.plan["up"]
        for r in recs:
            p = M + str(r["idx"])
            n = str(r["idx"])
            if r["kind"] == "res":
                P[n + ".g0"], P[n + ".b0"] = vec(p + ".GroupNorm_0.weight"), vec(p + ".GroupNorm_0.bias")
                (P[n + ".w0"], r["h2_0"]), P[n + ".c0"] = conv_w(p + ".Conv_0.weight", r["cin"]), vec(p + ".Conv_0.bias")
                P[n + ".g1"], P[n + ".b1"] = vec(p + ".GroupNorm_1.weight"), vec(p + ".GroupNorm_1.bias")
                (P[n + ".w1"], r["h2_1"]), P[n + ".c1"] = conv_w(p + ".Conv_1.weight", r["cout"]), vec(p + ".Conv_1.bias")
                if r["cin"] != r["cout"] or r["mode"]:
                    # Conv_2 reads the RAW (possibly resampled) block input: on the f16x3 path GroupNorm-apply /
                    # the resampler emit it in operand form, so the 1x1 runs on the fp16 matrix path too
                    r["h2_s"] = r["h2_0"] and r.get("c1", r["cin"]) % 8 == 0
                    if r["h2_s"]:
                        P[n + ".w2"] = self._pack_h2w(sd[p + ".Conv_2.weight"].detach())
                    else:
                        P[n + ".w2"] = ops.pack_conv_weight(sd[p + ".Conv_2.weight"].detach()).to(dev)
                    P[n + ".c2"] = vec(p + ".Conv_2.bias")
                dw.append(sd[p + ".Dense_0.weight"].detach().float())
                db.append(sd[p + ".Dense_0.bias"].detach().float())
                r["dense_off"] = off
                off += r["cout"]
            else:
                P[n + ".g"], P[n + ".b"] = vec(p + ".GroupNorm_0.weight"), vec(p + ".GroupNorm_0.bias")
                wq = torch.cat([sd[p + f".NIN_{j}.W"].detach().float() for j in range(3)], dim=1)
                r["h2"] = self.h2mode and r["ch"] % 32 == 0
                # NIN W is [in, out]; the h2 packer wants [out, in] (conv OI layout)
                P[n + ".wqkv"] = self._pack_h2w(wq.t().contiguous()) if r["h2"] else ops.pack_nin_weight(wq).to(dev)
                P[n + ".cqkv"] = torch.cat([sd[p + f".NIN_{j}.b"].detach().float() for j in range(3)]).contiguous().to(dev)
                P[n + ".w3"], P[n + ".c3"] = ops.pack_nin_weight(sd[p + ".NIN_3.W"].detach()).to(dev), vec(p + ".NIN_3.b")
                # lean fp16 x fp16 modes: the fused attention kernel (head dimension C = 256 at 16x16) writes the zero-bordered
                # fp16 operand of NIN_3 itself, which then runs on the fp16 matrix path (as proj_out does in GuidedUNet)
                r["proj16"] = self._lean and r["h2"] and r["ch"] == 256
                if r["proj16"]:
                    P[n + ".w3h"] = self._pack_h2w(sd[p + ".NIN_3.W"].detach().float().t().contiguous())
        P["dense.w"] = ops.pack_linear_weight(torch.cat(dw, dim=0)).to(dev)
        P["dense.b"] = torch.cat(db, dim=0).contiguous().to(dev)
        self.dense_cols = off
        # fp16 residual stream (round 4; see GuidedUNet.__init__): every convolution on the stream on the fp16 matrix path, no FIR
        # resampling (its stencils live in the fp32 GroupNorm-apply kernels); DIFFPURE_LEAN16=0 switches it off
        res_recs = [r for r in recs if r["kind"] == "res"]
        self._lean16 = (self._lean and os.environ.get("DIFFPURE_LEAN16", "1") != "0" and not self._fir
                        and all(r["h2_0"] and r["h2_1"] and ((r["cin"] == r["cout"] and not r["mode"]) or r.get("h2_s", False)) for r in res_recs)
                        and all(r["h2"] for r in recs if r["kind"] == "attn"))
        if self._lean16:
            for r in res_recs:      # Conv_2 (1x1 shortcut) as K-segments of Conv_1: one fused panel next to the separate ones
                if r["cin"] != r["cout"] or r["mode"]:       # (round 6: also the up / down blocks - their Conv_2 reads the RESAMPLED raw input)
                    p, n = M + str(r["idx"]), str(r["idx"])
                    P[n + ".w1s"] = self._pack_h2w(ops.fuse_skip_weight(sd[p + ".Conv_1.weight"], sd[p + ".Conv_2.weight"]))
                    P[n + ".c1s"] = (sd[p + ".Conv_1.bias"].detach().float() + sd[p + ".Conv_2.bias"].detach().float()).contiguous().to(dev)
        gi, ci = self.plan["gn_idx"], self.plan["conv_idx"]
        P["out.g"], P["out.b"] = vec(M + f"{gi}.weight"), vec(M + f"{gi}.bias")
        (P["out.w"], self._out_h2), P["out.c"] = conv_w(M + f"{ci}.weight", self.plan["final_ch"]), vec(M + f"{ci}.bias")
        self._resolve_pool(P)
        self.p = P
        return self

    # -- forward-path weights of the fp16-matrix-core convolutions --------------------------------------------------
    def _pack_h2w(self, w):
        """h2 (hi|lo) panel, or - precision "f16" / "f16sr" - a slot of the network's fp16 weight pool (ops.WeightPool)"""
        if self._pool is None:
            return ops.pack_conv_weight_h2(w, self.device)
        name = f"w{len(self._pool._pending)}"
        self._pool.add(name, w)
        return ops.PoolSlot(name)

    def _resolve_pool(self, P):
        if self._pool is not None:
            self._pool.finalize()
            for k, v in list(P.items()):
                if isinstance(v, ops.PoolSlot):
                    self._pool.bind(P, k, v.name)

    def _rmode(self, mode):
        """plan mode (0 / up / down) -> kernel resampling mode: nearest / mean, or upfirdn2d when the config says fir: True"""
        if not self._fir or not mode:
            return mode
        return ops.RESAMPLE_FIR_UP if mode == ops.RESAMPLE_UP else ops.RESAMPLE_FIR_DOWN

    def reround(self, key):
        """precision "f16sr": re-round every fp16 weight panel stochastically for this network call (one launch); the
        purification loops pass the step index.  No-op in every other mode."""
        if self._pool is not None:
            self._pool.round(key)
        if getattr(self, "_gpool", None) is not None:
            self._gpool.round(key)

    def _o16(self, hw, tape):
        """is a residual-stream tensor of `hw` pixels per sample stored as plain fp16?  (see GuidedUNet._o16)"""
        return self._lean16 and (tape is None or self._tape16) and hw % 64 == 0

    def _res(self, r, xa, x2a, dense, tape=None):
        """xa, x2a: ops.Act (tensor + the column statistics its producing convolution left), ops.Deferred (a split-K convolution output whose
        reduction / epilogue waits for this block's GroupNorm: the fused block boundary of the <= 64-pixel levels, csrc/boundary.hip) or
        plain tensors; fp32, or plain fp16 on the fp16 residual stream"""
        x2 = ops.tensor_of(x2a)
        xs, xdt = xa.shape, ops.dtype_of(xa)
        P, n, co = self.p, str(r["idx"]), r["cout"]
        mode = self._rmode(r["mode"])
        fir = self._fir
        conv0 = self._ch2 if r["h2_0"] else ops.conv2d
        conv1 = self._ch2 if r["h2_1"] else ops.conv2d
        b = xs[0]
        ho, wo = ops._out_hw(xs[1], xs[2], mode)
        out16 = self._o16(ho * wo, tape)
        o16 = {"out_f16": True} if out16 else {}
        G0, G1 = self._groups(r["cin"]), self._groups(co)
        h2s = r.get("h2_s", False)
        # channel-changing block without resampling: Conv_2 (the 1x1 shortcut, layerspp.py:268-272) as K-segments of Conv_1
        c1 = xs[3]
        fused = (out16 and not mode and r["cin"] != co and xdt == torch.float16 and (x2 is None or x2.dtype == torch.float16) and (n + ".w1s") in P
                 and ops.takes_segments(ho, wo, 3, co, co, c1, r["cin"] - c1))
        # up / down block (round 6): GroupNorm-apply hands out the resampled raw input as a plain fp16 tensor in the same pass, and Conv_2 -
        # the 1x1 shortcut over it (layerspp.py:245-258, 268-272) - becomes K-segments of Conv_1 like the shortcut of a channel-changing block
        fused_rs = (out16 and bool(mode) and xdt == torch.float16 and (x2 is None or x2.dtype == torch.float16) and (n + ".w1s") in P
                    and mode in (ops.RESAMPLE_UP, ops.RESAMPLE_DOWN) and ops.takes_segments(ho, wo, 3, co, co, r["cin"], 0)
                    and os.environ.get("DIFFPURE_SKIP_FUSED", "1") != "0")
        want_raw = h2s and not mode and not fused
        if self._bfuse and r["h2_0"] and ops.deferred_fusable(xa, x2a, G0, mode):
            # GroupNorm_0 IS the block boundary of the previous convolution: its split-K reduction, epilogue, this normalisation and the
            # operand of Conv_0 in one launch; the stream tensor (residual / shortcut input of this block, skip of the up path) comes with it
            h, st0, xraw = ops.group_norm_deferred(xa, G0, self.GN_EPS, P[n + ".g0"], P[n + ".b0"], x2=x2, act=True, raw=want_raw,
                                                   want_out=True, want_stats=tape is not None)
            x = xa.t
        else:
            x = ops.tensor_of(xa)
            st0 = ops.group_norm_stats(xa, G0, self.GN_EPS, x2a)
            h = ops.group_norm(x, G0, self.GN_EPS, P[n + ".g0"], P[n + ".b0"], x2=x2, act=True,
                               resample=mode, split=r["h2_0"] and self._ofmt, stats=st0, raw=want_raw or fused_rs, fir=fir)
            if want_raw:
                h, xraw = h
            elif fused_rs:
                h, xres = h
        off = r["dense_off"]
        # (below 64 pixels per sample the column records straddle samples and
        #  GroupNorm reduces the tensor itself, which it reads as fp32)
        mid16 = (self._lean and (tape is None or self._tape16) and r["h2_0"] and r["h2_1"] and co % (4 * G1) == 0 and (ho * wo) % 64 == 0)
        if self._bfuse and r["h2_0"] and r["h2_1"] and ops.conv_defers(ho, wo, 3, r["cin"], co) and ops.splitk_gn_ok(ho, wo, co, 0, G1):
            # Conv_0 -> GroupNorm_1 inside the block: the tensor between the two convolutions is written only for a tape
            dd = conv0(h, P[n + ".w0"], co, 3, bias=P[n + ".c0"], temb=dense[:, off:off + co], defer=True, **({"out_f16": True} if mid16 else {}))
            h2, st1, _ = ops.group_norm_deferred(dd, G1, self.GN_EPS, P[n + ".g1"], P[n + ".b1"], act=True, want_out=tape is not None,
                                                 want_stats=tape is not None)
            if tape is not None:
                tape.append(dict(r=r, x=x, x2=x2, st0=st0, hmid=dd.t, st1=st1))
            h = h2
        else:
            h = conv0(h, P[n + ".w0"], co, 3, bias=P[n + ".c0"], temb=dense[:, off:off + co], colstats=True, **({"out_f16": True} if mid16 else {}))
            st1 = ops.group_norm_stats(h, G1, self.GN_EPS)
            h = h.t
            if tape is not None:
                tape.append(dict(r=r, x=x, x2=x2, st0=st0, hmid=h, st1=st1))
            h = ops.group_norm(h, G1, self.GN_EPS, P[n + ".g1"], P[n + ".b1"], act=True, split=r["h2_1"] and self._ofmt, stats=st1)
        # the block's output convolution leaves its reduction / epilogue to the NEXT GroupNorm where the level is a split-K one
        last = {"defer": True} if (self._bfuse and r["h2_1"] and ops.conv_defers(ho, wo, 3, co, co)) else {"colstats": True}
        if fused:
            return conv1(h, P[n + ".w1s"], co, 3, bias=P[n + ".c1s"], segs=(x,) if x2 is None else (x, x2), scale=INV_SQRT2, **last, **o16)
        if fused_rs:
            return conv1(h, P[n + ".w1s"], co, 3, bias=P[n + ".c1s"], segs=(xres,), scale=INV_SQRT2, **last, **o16)
        if mode:
            if h2s:
                skip = self._ch2(ops.to_h2(x, mode, fmt=self._ofmt, fir=fir), P[n + ".w2"], co, 1, bias=P[n + ".c2"], **o16)
            else:
                skip = ops.conv2d(ops.resample(x, mode, fir=fir), P[n + ".w2"], co, 1, bias=P[n + ".c2"])
        elif want_raw:
            skip = self._ch2(xraw, P[n + ".w2"], co, 1, bias=P[n + ".c2"], **o16)
        elif r["cin"] != co:
            skip = ops.conv2d(x, P[n + ".w2"], co, 1, bias=P[n + ".c2"], x2=x2)
        else:
            skip = x if x2 is None else torch.cat([x, x2], dim=3)
        return conv1(h, P[n + ".w1"], co, 3, bias=P[n + ".c1"], res=skip, scale=INV_SQRT2, **last, **o16)

    def _attn(self, r, xa, tape=None):
        P, n, c = self.p, str(r["idx"]), r["ch"]
        x = ops.tensor_of(xa)          # (attention sits at 16 x 16: never behind a split-K convolution; a Deferred would be finished here)
        b, hh, ww, _ = x.shape
        st = ops.group_norm_stats(xa, self._groups(c), self.GN_EPS)
        hn = ops.group_norm(x, self._groups(c), self.GN_EPS, P[n + ".g"], P[n + ".b"], split=r["h2"] and self._ofmt, stats=st)
        # fp16 x fp16 modes without a tape: fp16 qkv, one fp16 pass of the flash kernel (see GuidedUNet._attn)
        fused = bool(r.get("proj16")) and ops.attention_fused_ok(hh * ww, c)
        q16 = fused and (tape is None or self._tape16) and os.environ.get("DIFFPURE_ATTN16", "1") != "0"
        qkv = (self._ch2 if r["h2"] else ops.conv2d)(hn, P[n + ".wqkv"], 3 * c, 1, bias=P[n + ".cqkv"], **({"out_f16": True} if q16 else {}))
        # the taped forward keeps only qkv; the probabilities are recomputed per block in the backward pass (see GuidedUNet._attn)
        if tape is not None:
            tape.append(dict(r=r, x=x, st=st, qkv=qkv))
        if fused:                                   # with or without a tape, as GuidedUNet._attn
            ah = ops.attention_fused(qkv.view(b, hh * ww, 3 * c), 1, "split", operand_hw=(hh, ww))
            return self._ch2(ah, P[n + ".w3h"], c, 1, bias=P[n + ".c3"], res=x, scale=INV_SQRT2, colstats=True,
                             **({"out_f16": True} if self._o16(hh * ww, tape) else {}))
        a = ops.attention(qkv.view(b, hh * ww, 3 * c), 1, "split")
        # shapes the fused kernel does not cover: NIN_3 on the fp32 path (fp32 residual), the result in the stream's format of this level
        return ops.conv2d(a.view(b, hh, ww, c), P[n + ".w3"], c, 1, bias=P[n + ".c3"], res=x.float() if x.dtype == torch.float16 else x,
                          scale=INV_SQRT2, colstats=True, out_f16=self._o16(hh * ww, tape))

    def time_table(self, labels):
        """labels: float32 GPU tensor [R] (= 999*s). -> Dense_0 rows of every ResBlock [R, sum(cout)]."""
        P, nf = self.p, self.cfg["nf"]
        e = ops.timestep_embedding(labels, self.freqs, cos_first=False)
        e = ops.linear(e, P["t0.w"], 4 * nf, P["t0.b"])
        e = ops.linear(ops.silu(e), P["t1.w"], 4 * nf, P["t1.b"])
        return ops.linear(ops.silu(e), P["dense.w"], self.dense_cols, P["dense.b"])

    def forward(self, x, labels=None, table_row=None, tape=None):
        if not self.p:
            raise RuntimeError("NCSNpp: weights not loaded")
        P = self.p
        dense = table_row if table_row is not None else self.time_table(labels)
        st = self.plan["stem"]
        o16 = self._o16(x.shape[1] * x.shape[2], tape)
        if "stem.w16" in P and ops.conv2d_stem_ok(x.shape[3], x.shape[0], x.shape[1], x.shape[2], st["cout"]):
            hs = [ops.conv2d_stem(x, P["stem.w16"], st["cout"], bias=P["stem.b"], colstats=True, out_f16=o16)]
        else:
            hs = [ops.conv2d(x, P["stem.w"], st["cout"], 3, bias=P["stem.b"], colstats=True, out_f16=o16)]
        for blk in self.plan["down"]:
            h = hs[-1]
            for r in blk:
                h = self._res(r, h, None, dense, tape) if r["kind"] == "res" else self._attn(r, h, tape)
            hs.append(h)
        h = hs[-1]
        for r in self.plan["mid"]:
            h = self._res(r, h, None, dense, tape) if r["kind"] == "res" else self._attn(r, h, tape)
        for r in self.plan["up"]:
            if r["kind"] == "attn":
                h = self._attn(r, h, tape)
            elif r.get("pop"):
                h = self._res(r, h, hs.pop(), dense, tape)
            else:
                h = self._res(r, h, None, dense, tape)
        assert not hs
        g = self._groups(self.plan["final_ch"])
        h = ops.resolved(h)
        sth = ops.group_norm_stats(h, g, self.GN_EPS)
        h = ops.tensor_of(h)
        if tape is not None:
            tape.append(dict(head=True, x=h, st=sth))
        h = ops.group_norm(h, g, self.GN_EPS, P["out.g"], P["out.b"], act=True, split=self._out_h2 and self._ofmt, stats=sth)
        return (self._ch2 if self._out_h2 else ops.conv2d)(h, P["out.w"], self.cfg["channels"], 3, bias=P["out.c"])

    __call__ = forward

    # -- input gradient (vector-Jacobian product); see GuidedUNet.vjp -------------------------------
    def enable_grad(self):
        if self._grad_ready:
            return self
        sd, dev, P = self._sd, self.device, self.p
        M = "all_modules."

        # fp16 x fp16 modes ("f16", "f16sr"): the input-gradient convolutions run in the arithmetic of the forward they
        # differentiate - ONE fp16 MFMA pass on plain-fp16 gradient operands and fp16 dgrad panels that live in a second
        # weight pool, re-rounded with the forward's panels (stochastically, per network call, in "f16sr") - instead of the
        # three-pass split-fp16 path (which the 22-bit modes keep): a third of the MFMA work, on the fast tile kernels.
        # DIFFPURE_GRAD16=0 keeps the three-pass gradients.
        gpool = None
        if self._pool is not None and os.environ.get("DIFFPURE_GRAD16", "1") != "0":
            gpool = ops.WeightPool(dev, stochastic=self.precision == "f16sr", seed=0x6AAD5EED)
        pooled = []

        def dg(w, n_in_dgrad, lo=None, hi=None, key=None):
            wd = ops.dgrad_weight(w.detach().float())
            if lo is not None:
                wd = wd[lo:hi]
            if gpool is not None and n_in_dgrad % 32 == 0 and key is not None:
                gpool.add(key, wd)
                pooled.append(key)
                return None, "h1"
            if self.h2mode and n_in_dgrad % 32 == 0:
                return ops.pack_conv_weight_h2(wd, dev), True
            return ops.pack_conv_weight(wd).to(dev), False

        P["stem.dw"], self._stem_dh2 = dg(sd[M + f"{self.plan['stem']['idx']}.weight"], self.cfg["nf"] if gpool is not None else -1, key="stem.dw")
        recs = [r for b in self.plan["down"] for r in b] + self.plan["mid"] + self.plan["up"]
        for r in recs:
            p, n = M + str(r["idx"]), str(r["idx"])
            if r["kind"] == "res":
                P[n + ".dw1"], r["dh2_1"] = dg(sd[p + ".Conv_1.weight"], r["cout"], key=n + ".dw1")
                P[n + ".dw0"], r["dh2_0"] = dg(sd[p + ".Conv_0.weight"], r["cout"], key=n + ".dw0")
                if r["cin"] != r["cout"] or r["mode"]:
                    c1 = r.get("c1", r["cin"])
                    g1x1 = r["cout"] if gpool is not None else -1       # the 1x1 dgrads join the fp16 path only in the fp16 x fp16 modes
                    P[n + ".dw2a"], r["dh2_2"] = dg(sd[p + ".Conv_2.weight"], g1x1, 0, c1, key=n + ".dw2a")
                    if c1 != r["cin"]:
                        P[n + ".dw2b"], _ = dg(sd[p + ".Conv_2.weight"], g1x1, c1, r["cin"], key=n + ".dw2b")
            else:
                c = r["ch"]
                wq = torch.cat([sd[p + f".NIN_{j}.W"].detach().float() for j in range(3)], dim=1)   # [C, 3C] (in, out)
                P[n + ".dwqkv"], r["dh2"] = dg(wq.t().contiguous(), 3 * c, key=n + ".dwqkv")          # as OI conv weight
                P[n + ".dw3"], r["dh2_3"] = dg(sd[p + ".NIN_3.W"].detach().float().t().contiguous(), c if gpool is not None else -1, key=n + ".dw3")
        self._gpool = gpool
        if gpool is not None:
            gpool.finalize()
            # built lazily by the first vjp of a loop, i.e. AFTER that call's reround: take the key the forward's panels carry, so
            # that the first backward step is rounded like every later one (and a repeated call reproduces the first bit for bit)
            if self._pool._last_key is not None:
                gpool.round(self._pool._last_key)
            for key in pooled:
                gpool.bind(P, key, key)
        self._grad_ready = True
        return self

    def _dconv(self, dy, key, is_h2, n_out, ksize, scale=1.0):
        if is_h2 == "h1":             # one fp16 pass: plain-fp16 gradient operand x pooled fp16 dgrad panel
            if dy.dtype != torch.float16:
                dy = ops.to_h2(dy, fmt="h1")
            return ops.conv2d_h2(dy, self.p[key], n_out, ksize, scale=scale, w_fmt=1)
        if is_h2:
            if dy.dtype != torch.float16:
                dy = ops.to_h2(dy)
            return ops.conv2d_h2(dy, self.p[key], n_out, ksize, scale=scale)    # gradients: h2 operand, three passes
        return ops.conv2d(dy, self.p[key], n_out, ksize, scale=scale)

    def _res_bwd(self, t, dout):
        r, P = t["r"], self.p
        n, co, ci, mode = str(r["idx"]), r["cout"], r["cin"], self._rmode(r["mode"])
        fir = self._fir
        # out = (skip + conv1(h3)) * s
        if r["dh2_1"] == "h1":        # one plain-fp16 copy of dout serves the 3x3 and the 1x1 dgrad convolutions
            dout16 = ops.to_h2(dout, fmt="h1")
        else:
            dout16 = dout
        g2 = r.get("dh2_2", False) if r["dh2_1"] == "h1" else False
        dh3 = self._dconv(dout16, n + ".dw1", r["dh2_1"], co, 3, scale=INV_SQRT2)
        dh2, _ = ops.group_norm_bwd(t["hmid"], self._groups(co), P[n + ".g1"], P[n + ".b1"], t["st1"], dh3, act=True,
                                    split=r["dh2_0"])
        dh1 = self._dconv(dh2, n + ".dw0", r["dh2_0"], ci, 3)
        # the gradient of the skip branch joins dx inside the GroupNorm backward (one pass on small feature maps; ops.group_norm_bwd)
        ad = ad2 = None
        asc = 1.0
        if mode:
            ds = self._dconv(dout16 if g2 else dout, n + ".dw2a", g2, ci, 1, scale=INV_SQRT2)
            ad = ops.resample_bwd(ds, mode, fir=fir)
        elif ci != co:
            c1 = t["x"].shape[3]
            ad = self._dconv(dout16 if g2 else dout, n + ".dw2a", g2, c1, 1, scale=INV_SQRT2)
            if t["x2"] is not None:
                ad2 = self._dconv(dout16 if g2 else dout, n + ".dw2b", g2, ci - c1, 1, scale=INV_SQRT2)
        else:
            ad, asc = dout, INV_SQRT2
        return ops.group_norm_bwd(t["x"], self._groups(ci), P[n + ".g0"], P[n + ".b0"], t["st0"], dh1, x2=t["x2"], act=True,
                                  resample=mode, fir=fir, addend=ad, addend2=ad2, addend_scale=asc)

    def _attn_bwd(self, t, dout):
        r, P = t["r"], self.p
        n, c = str(r["idx"]), r["ch"]
        b, hh, ww, _ = dout.shape
        da = self._dconv(dout, n + ".dw3", r.get("dh2_3", False) == "h1" and "h1", c, 1, scale=INV_SQRT2)
        h16_ok = getattr(self, "_gpool", None) is not None and os.environ.get("DIFFPURE_ATTN_BWD16", "1") != "0"
        qkv = t["qkv"]
        # taped on the fp16 stream: the backward reads q, k, v IN PLACE where all five of its products run on dp_gemm_strided_h16 (NCSN++:
        # T = d = 256); where some shape stays on the fp32-input kernel (64-wide heads: N = 64) the tensor is up-converted once (same values)
        if qkv.dtype == torch.float16 and not (h16_ok and ops.attention_h16_serves(hh * ww, c // 1)):
            qkv = qkv.float()
        qkv = qkv.view(b, hh * ww, 3 * c)
        # probabilities recomputed (freed after this block); in the fp16 x fp16 modes the five products run on the fp16 matrix cores,
        # as the dgrad convolutions around them do (DIFFPURE_GRAD16=0 keeps fp32-input MFMA)
        h16 = h16_ok
        _, probs = ops.attention(qkv, 1, "split", probs_only=True, h16=h16)
        dqkv = ops.attention_bwd(qkv, probs, da.view(b, hh * ww, c), 1, "split", h16=h16)
        del probs
        dhn = self._dconv(dqkv.view(b, hh, ww, 3 * c), n + ".dwqkv", r["dh2"], c, 1)
        return ops.group_norm_bwd(t["x"], self._groups(c), P[n + ".g"], P[n + ".b"], t["st"], dhn, addend=dout, addend_scale=INV_SQRT2)[0]

    def vjp(self, tape, dout):
        """(d out / d x)^T dout for the forward that filled `tape`; dout [B,H,W,channels] fp32."""
        self.enable_grad()
        P = self.p
        M = "all_modules."
        if "out.dw" not in P:
            P["out.dw"] = ops.pack_conv_weight(ops.dgrad_weight(self._sd[M + f"{self.plan['conv_idx']}.weight"].detach().float())).to(self.device)
        tape = list(tape)
        head = tape.pop()
        fc = self.plan["final_ch"]
        dh = ops.conv2d(dout, P["out.dw"], fc, 3)
        dh, _ = ops.group_norm_bwd(head["x"], self._groups(fc), P["out.g"], P["out.b"], head["st"], dh, act=True)
        skips = []
        for r in reversed(self.plan["up"]):
            t = tape.pop()
            assert t["r"] is r
            if r["kind"] == "attn":
                dh = self._attn_bwd(t, dh)
            else:
                dh, d2 = self._res_bwd(t, dh)
                if d2 is not None:
                    skips.append(d2)
        for r in reversed(self.plan["mid"]):
            t = tape.pop()
            dh = self._res_bwd(t, dh)[0] if r["kind"] == "res" else self._attn_bwd(t, dh)
        for blk in reversed(self.plan["down"]):
            dh = ops.add(dh, skips.pop())
            for r in reversed(blk):
                t = tape.pop()
                dh = self._res_bwd(t, dh)[0] if r["kind"] == "res" else self._attn_bwd(t, dh)
        dh = ops.add(dh, skips.pop())
        assert not skips and not tape
        return self._dconv(dh, "stem.dw", self._stem_dh2 == "h1" and "h1", self.cfg["channels"], 3)
