"""MI355X engine for the guided-diffusion score network (ImageNet 256x256 UNet).

Same function as guided_diffusion.unet.UNetModel.forward (/root/reference/guided_diffusion/
unet.py:642-671, topology :484-626) and it loads the same `state_dict` keys, but it is not a
module tree: the network is flattened once into a list of block records over NHWC fp32
activations, and every record runs on the HIP kernels of libdiffpure_hip.so:

  ResBlock (unet.py:244-264)   = gn_stats -> gn_apply(+SiLU, +2x up/down) -> conv3x3(+bias)
                                 -> gn_stats -> gn_apply(FiLM (1+scale), shift, +SiLU)
                                 -> conv3x3(+bias, +residual [identity | fused 1x1 skip])
  AttentionBlock (:307-313)    = gn -> conv1x1(qkv) -> QK^T -> softmax -> PV -> conv1x1(+residual)
  skip concatenation (:667)    never materialised: GroupNorm and the convolutions read the two
                               source tensors directly (channel-split loaders).
  timestep conditioning        all 40+ per-block `emb_layers` Linear layers (unet.py:209-217) are
                               one packed GEMM per forward; each block reads its FiLM rows from
                               that table (broadcast over the batch when the timestep is uniform,
                               which it always is inside the purification loop).
"""
import math
from collections import OrderedDict

import functools
import os

import torch

from . import ops


def parse_config(model_cfg):
    """yaml/`model_and_diffusion_defaults` dict -> resolved hyper-parameters, as
    guided_diffusion/script_util.py:138-192 (create_model) resolves them."""
    image_size = int(model_cfg["image_size"])
    cm = model_cfg.get("channel_mult", "")
    if cm is None or cm == "":
        table = {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4)}
        if image_size not in table:
            raise ValueError(f"unsupported image size: {image_size}")
        cm = table[image_size]
    elif isinstance(cm, str):
        cm = tuple(int(v) for v in cm.split(","))
    att = model_cfg.get("attention_resolutions", "16,8")
    if isinstance(att, str):
        att = tuple(image_size // int(r) for r in att.split(","))
    nh = int(model_cfg.get("num_heads", 4))
    nhu = int(model_cfg.get("num_heads_upsample", -1))
    return dict(
        image_size=image_size,
        in_channels=3,
        model_channels=int(model_cfg["num_channels"]),
        out_channels=6 if model_cfg.get("learn_sigma", False) else 3,
        channel_mult=tuple(cm),
        num_res_blocks=int(model_cfg["num_res_blocks"]),
        attention_ds=tuple(att),
        num_heads=nh,
        num_heads_upsample=nh if nhu == -1 else nhu,
        num_head_channels=int(model_cfg.get("num_head_channels", -1)),
        resblock_updown=bool(model_cfg.get("resblock_updown", False)),
        use_scale_shift_norm=bool(model_cfg.get("use_scale_shift_norm", True)),
        use_new_attention_order=bool(model_cfg.get("use_new_attention_order", False)),
    )


def _plan(cfg):
    """Flatten the UNet into records. Each record: dict(kind, name, ...)."""
    mc, nrb, mults = cfg["model_channels"], cfg["num_res_blocks"], cfg["channel_mult"]
    if not cfg["resblock_updown"] or not cfg["use_scale_shift_norm"]:
        raise NotImplementedError("engine covers the DiffPure ImageNet family: resblock_updown + scale-shift norm")

    def heads(ch, up):
        if cfg["num_head_channels"] == -1:
            return cfg["num_heads_upsample"] if up else cfg["num_heads"]
        return ch // cfg["num_head_channels"]

    inp, out = [], []
    ch = int(mults[0] * mc)
    chans = [ch]
    inp.append([dict(kind="stem", name="input_blocks.0.0", cin=cfg["in_channels"], cout=ch)])
    idx, ds = 1, 1
    for level, mult in enumerate(mults):
        for _ in range(nrb):
            co = int(mult * mc)
            blk = [dict(kind="res", name=f"input_blocks.{idx}.0", cin=ch, cout=co, mode=0)]
            ch = co
            if ds in cfg["attention_ds"]:
                blk.append(dict(kind="attn", name=f"input_blocks.{idx}.1", ch=ch, heads=heads(ch, False)))
            inp.append(blk)
            chans.append(ch)
            idx += 1
        if level != len(mults) - 1:
            inp.append([dict(kind="res", name=f"input_blocks.{idx}.0", cin=ch, cout=ch, mode=ops.RESAMPLE_DOWN)])
            chans.append(ch)
            idx += 1
            ds *= 2
    mid = [
        dict(kind="res", name="middle_block.0", cin=ch, cout=ch, mode=0),
        dict(kind="attn", name="middle_block.1", ch=ch, heads=heads(ch, False)),
        dict(kind="res", name="middle_block.2", cin=ch, cout=ch, mode=0),
    ]
    oidx = 0
    for level, mult in list(enumerate(mults))[::-1]:
        for i in range(nrb + 1):
            ich = chans.pop()
            co = int(mc * mult)
            blk = [dict(kind="res", name=f"output_blocks.{oidx}.0", cin=ch + ich, cout=co, mode=0, split=ch)]
            ch = co
            sub = 1
            if ds in cfg["attention_ds"]:
                blk.append(dict(kind="attn", name=f"output_blocks.{oidx}.{sub}", ch=ch, heads=heads(ch, True)))
                sub += 1
            if level and i == nrb:
                blk.append(dict(kind="res", name=f"output_blocks.{oidx}.{sub}", cin=ch, cout=ch, mode=ops.RESAMPLE_UP))
                ds //= 2
            out.append(blk)
            oidx += 1
    return dict(inp=inp, mid=mid, out=out, final_ch=ch)


def _res_shapes(r, emb_dim):
    n, ci, co = r["name"], r["cin"], r["cout"]
    sh = OrderedDict()
    sh[f"{n}.in_layers.0.weight"] = (ci,)
    sh[f"{n}.in_layers.0.bias"] = (ci,)
    sh[f"{n}.in_layers.2.weight"] = (co, ci, 3, 3)
    sh[f"{n}.in_layers.2.bias"] = (co,)
    sh[f"{n}.emb_layers.1.weight"] = (2 * co, emb_dim)
    sh[f"{n}.emb_layers.1.bias"] = (2 * co,)
    sh[f"{n}.out_layers.0.weight"] = (co,)
    sh[f"{n}.out_layers.0.bias"] = (co,)
    sh[f"{n}.out_layers.3.weight"] = (co, co, 3, 3)
    sh[f"{n}.out_layers.3.bias"] = (co,)
    if ci != co:
        sh[f"{n}.skip_connection.weight"] = (co, ci, 1, 1)
        sh[f"{n}.skip_connection.bias"] = (co,)
    return sh


def param_shapes(cfg):
    """state_dict key -> shape, in the reference's registration order (unet.py:475-624)."""
    mc = cfg["model_channels"]
    ed = 4 * mc
    plan = _plan(cfg)
    sh = OrderedDict()
    sh["time_embed.0.weight"] = (ed, mc)
    sh["time_embed.0.bias"] = (ed,)
    sh["time_embed.2.weight"] = (ed, ed)
    sh["time_embed.2.bias"] = (ed,)
    for blk in [r for b in plan["inp"] for r in b] + plan["mid"] + [r for b in plan["out"] for r in b]:
        n = blk["name"]
        if blk["kind"] == "stem":
            sh[f"{n}.weight"] = (blk["cout"], blk["cin"], 3, 3)
            sh[f"{n}.bias"] = (blk["cout"],)
        elif blk["kind"] == "res":
            sh.update(_res_shapes(blk, ed))
        else:
            c = blk["ch"]
            sh[f"{n}.norm.weight"] = (c,)
            sh[f"{n}.norm.bias"] = (c,)
            sh[f"{n}.qkv.weight"] = (3 * c, c, 1)
            sh[f"{n}.qkv.bias"] = (3 * c,)
            sh[f"{n}.proj_out.weight"] = (c, c, 1)
            sh[f"{n}.proj_out.bias"] = (c,)
    fc = plan["final_ch"]
    sh["out.0.weight"] = (fc,)
    sh["out.0.bias"] = (fc,)
    sh["out.2.weight"] = (cfg["out_channels"], fc, 3, 3)
    sh["out.2.bias"] = (cfg["out_channels"],)
    return sh


class GuidedUNet:
    """score network eps_theta(x, t): NHWC in, NHWC out ([B, H, W, out_channels])."""

    GN_GROUPS = 32   # nn.py:101-108 normalization(): GroupNorm32(32, C)
    GN_EPS = 1e-5

    def __init__(self, cfg, device, precision="f32"):
        """precision: "f32"  = exact fp32-input MFMA everywhere;
                      "f16x3" = split-fp16 three-pass MFMA for every convolution whose input is a
                      GroupNorm output (all 3x3 convolutions but the stem, and the qkv 1x1):
                      fp32-class accuracy, ~5x the matrix ceiling (csrc/igemm_h2.hip)."""
        if precision != "f32" and precision not in ops.H2_MODES:
            raise ValueError(f"unknown precision {precision!r}")
        self.cfg = cfg
        self.precision = precision
        # fp16-matrix-core convolution path: MFMA passes per product and the operand format GroupNorm-apply emits
        self.h2mode = precision in ops.H2_MODES
        passes, ofmt = ops.H2_MODES.get(precision, (3, ops.FMT_H2))
        self._ofmt = "h1" if ofmt == ops.FMT_H1 else "h2"
        self._pool = ops.WeightPool(torch.device(device), stochastic=precision == "f16sr") if precision in ops.W16_MODES else None
        self._ch2 = functools.partial(ops.conv2d_h2, passes=passes, w_fmt=1 if self._pool is not None else 0)
        # fp16 x fp16 modes: a ResBlock's first convolution stores its output as fp16 (its only reader is the GroupNorm-apply
        # that emits the fp16 operand of the second one) and the attention output reaches proj_out as an fp16 operand
        self._lean = self._pool is not None and os.environ.get("DIFFPURE_LEAN", "1") != "0"
        # round 6: fused block boundaries of the <= 64-pixel levels (csrc/boundary.hip; ops.Deferred) - fp16 x fp16 modes (the kernel emits
        # plain fp16 operands); DIFFPURE_BOUNDARY=0 restores the four-launch chain
        self._bfuse = self._pool is not None and self._ofmt == "h1"
        # ... and (round 4, DIFFPURE_LEAN16=0 switches it off) the RESIDUAL STREAM itself travels as plain fp16 between the blocks -
        # the reference's own arithmetic for this network (`use_fp16: True`, configs/imagenet.yml:18: convert_to_fp16 casts the
        # whole torso, unet.py:626-632, so h IS fp16 there) with fp32 accumulation / epilogues / GroupNorm statistics on top.
        # Decided in load_state_dict (every convolution on the stream must be on the fp16 matrix path).  Round 5: forward passes that
        # keep a tape (the adjoints) run on the same stream (`_tape16`; see _o16).
        self._lean16 = False
        self._tape16 = os.environ.get("DIFFPURE_TAPE16", "1") != "0"
        self.device = torch.device(device)
        self.plan = _plan(cfg)
        self.p = {}
        mc = cfg["model_channels"]
        half = mc // 2
        # nn.py:121-123, evaluated on the host exactly as the reference does, then kept resident
        self.freqs = torch.exp(-math.log(10000) * torch.arange(start=0, end=half, dtype=torch.float32) / half).to(self.device)
        self.emb_cols = 0

    # -- weights ---------------------------------------------------------------------------------
    def load_state_dict(self, sd):
        want = param_shapes(self.cfg)
        missing = [k for k in want if k not in sd]
        if missing:
            raise KeyError(f"state_dict is missing {len(missing)} keys, e.g. {missing[:3]}")
        for k, shp in want.items():
            if tuple(sd[k].shape) != tuple(shp):
                raise ValueError(f"{k}: expected {shp}, got {tuple(sd[k].shape)}")
        dev = self.device
        P = {}
        self._sd = sd          # host copy: the input-gradient (dgrad) panels are packed lazily by enable_grad()
        self._grad_ready = False

        def vec(k):
            return sd[k].detach().float().contiguous().to(dev)

        def conv_w(k, cin):
            """-> (packed weight, is_h2)"""
            if self.h2mode and cin % 32 == 0:
                return self._pack_h2w(sd[k].detach()), True
            return ops.pack_conv_weight(sd[k].detach()).to(dev), False

        P["te0.w"] = ops.pack_linear_weight(sd["time_embed.0.weight"].detach()).to(dev)
        P["te0.b"] = vec("time_embed.0.bias")
        P["te2.w"] = ops.pack_linear_weight(sd["time_embed.2.weight"].detach()).to(dev)
        P["te2.b"] = vec("time_embed.2.bias")
        emb_w, emb_b, off = [], [], 0
        blocks = [r for b in self.plan["inp"] for r in b] + self.plan["mid"] + [r for b in self.plan["out"] for r in b]
        for r in blocks:
            n = r["name"]
            if r["kind"] == "stem":
                P[n + ".w"] = ops.pack_conv_weight(sd[n + ".weight"].detach()).to(dev)
                if self.h2mode and r["cin"] == 3:       # the write-bound stem kernel (csrc/stem.hip) where its shape test passes
                    P[n + ".w16"] = ops.pack_stem_weight(sd[n + ".weight"]).to(dev)
                P[n + ".b"] = vec(n + ".bias")
            elif r["kind"] == "res":
                P[n + ".g1"], P[n + ".b1"] = vec(n + ".in_layers.0.weight"), vec(n + ".in_layers.0.bias")
                P[n + ".w1"], r["h2_1"] = conv_w(n + ".in_layers.2.weight", r["cin"])
                P[n + ".c1"] = vec(n + ".in_layers.2.bias")
                P[n + ".g2"], P[n + ".b2"] = vec(n + ".out_layers.0.weight"), vec(n + ".out_layers.0.bias")
                P[n + ".w2"], r["h2_2"] = conv_w(n + ".out_layers.3.weight", r["cout"])
                P[n + ".c2"] = vec(n + ".out_layers.3.bias")
                if r["cin"] != r["cout"]:
                    # the 1x1 skip reads the RAW block input; with f16x3, GroupNorm-apply emits that input in
                    # operand form in the same pass, so the skip runs on the fp16 matrix path too
                    c1 = r.get("split", r["cin"])
                    r["h2_s"] = r["h2_1"] and c1 % 8 == 0
                    if r["h2_s"]:
                        P[n + ".ws"] = self._pack_h2w(sd[n + ".skip_connection.weight"].detach())
                    else:
                        P[n + ".ws"] = ops.pack_conv_weight(sd[n + ".skip_connection.weight"].detach()).to(dev)
                    P[n + ".cs"] = vec(n + ".skip_connection.bias")
                emb_w.append(sd[n + ".emb_layers.1.weight"].detach().float())
                emb_b.append(sd[n + ".emb_layers.1.bias"].detach().float())
                r["emb_off"] = off
                off += 2 * r["cout"]
            else:
                P[n + ".g"], P[n + ".b"] = vec(n + ".norm.weight"), vec(n + ".norm.bias")
                P[n + ".wqkv"], r["h2"] = conv_w(n + ".qkv.weight", r["ch"])
                P[n + ".cqkv"] = vec(n + ".qkv.bias")
                P[n + ".wproj"] = ops.pack_conv_weight(sd[n + ".proj_out.weight"].detach()).to(dev)
                # lean fp16 x fp16 modes: proj_out runs on the fp16 matrix path too, fed by the attention kernel's fp16 operand
                r["proj16"] = self._lean and r["h2"] and r["ch"] // r["heads"] == 64
                if r["proj16"]:
                    P[n + ".wproj16"] = self._pack_h2w(sd[n + ".proj_out.weight"].detach())
                P[n + ".cproj"] = vec(n + ".proj_out.bias")
        # one packed [emb_dim, sum(2*cout)] panel for every block's emb_layers Linear
        P["emb.w"] = ops.pack_linear_weight(torch.cat(emb_w, dim=0)).to(dev)
        P["emb.b"] = torch.cat(emb_b, dim=0).contiguous().to(dev)
        self.emb_cols = off
        P["out.g"], P["out.b"] = vec("out.0.weight"), vec("out.0.bias")
        P["out.w"], self._out_h2 = conv_w("out.2.weight", self.plan["final_ch"])
        P["out.c"] = vec("out.2.bias")
        res_blocks = [r for r in blocks if r["kind"] == "res"]
        self._lean16 = (self._lean and os.environ.get("DIFFPURE_LEAN16", "1") != "0" and self._out_h2
                        and all(r["h2_1"] and r["h2_2"] and (r["cin"] == r["cout"] or r.get("h2_s", False)) for r in res_blocks)
                        and all(r["h2"] for r in blocks if r["kind"] == "attn"))
        if self._lean16:
            # the 1x1 skip_connection of a channel-changing ResBlock as 1x1 K-segments of its second 3x3 convolution (one panel
            # [w2 | ws] next to the separate ones, which small launches and the taped forward keep using)
            for r in res_blocks:
                if r["cin"] != r["cout"] and not r["mode"]:
                    n = r["name"]
                    P[n + ".w2s"] = self._pack_h2w(ops.fuse_skip_weight(sd[n + ".out_layers.3.weight"], sd[n + ".skip_connection.weight"]))
                    P[n + ".c2s"] = (sd[n + ".out_layers.3.bias"].detach().float() + sd[n + ".skip_connection.bias"].detach().float()).contiguous().to(dev)
        self._resolve_pool(P)
        self.p = P
        return self

    # -- blocks ----------------------------------------------------------------------------------
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

    def reround(self, key):
        """precision "f16sr": re-round every fp16 weight panel stochastically for this network call (one launch); the
        purification loops pass the step index.  No-op in every other mode."""
        if self._pool is not None:
            self._pool.round(key)
        if getattr(self, "_gpool", None) is not None:
            self._gpool.round(key)

    def _o16(self, hw, tape):
        """is a residual-stream tensor of `hw` pixels per sample stored as plain fp16?  (fp16 x fp16 modes and whole column records per
        sample - an fp16 tensor's GroupNorm statistics exist only as its producer's records.)  Round 5: WITH a tape as well
        (`_tape16`, DIFFPURE_TAPE16=0 restores the fp32-stream tape of round 4): the adjoint solves re-run and differentiate exactly
        the network the forward solve evaluated - same stream format, same fused [w2 | skip] panels, same one-pass attention, bit
        for bit - the tape holds fp16 tensors (half the memory) and the GroupNorm backward reads them as stored (x_fmt 1)."""
        return self._lean16 and (tape is None or self._tape16) and hw % 64 == 0

    def _res(self, r, xa, x2a, film_table, tape=None):
        """xa, x2a: ops.Act (tensor + the column statistics its producing convolution left), ops.Deferred (a split-K convolution output whose
        reduction / epilogue waits for this block's GroupNorm - the fused block boundary of the 8 x 8 level, csrc/boundary.hip) or plain
        tensors; fp32, or plain fp16 on the fp16 residual stream"""
        x2 = ops.tensor_of(x2a)
        xs, xdt = xa.shape, ops.dtype_of(xa)
        P, n, co = self.p, r["name"], r["cout"]
        G, eps = self.GN_GROUPS, self.GN_EPS
        mode = r["mode"]
        conv1 = self._ch2 if r["h2_1"] else ops.conv2d
        conv2 = self._ch2 if r["h2_2"] else ops.conv2d
        b = xs[0]
        ho, wo = ops._out_hw(xs[1], xs[2], mode)
        out16 = self._o16(ho * wo, tape)
        # channel-changing block: the 1x1 skip as K-segments of the second convolution (fp16 stream, launches the 8-wave kernel takes)
        c1 = xs[3]
        fused = (out16 and not mode and r["cin"] != co and xdt == torch.float16 and (x2 is None or x2.dtype == torch.float16) and (n + ".w2s") in P
                 and ops.takes_segments(ho, wo, 3, co, co, c1, r["cin"] - c1))
        want_raw = (not mode) and r["cin"] != co and r.get("h2_s", False) and not fused
        # up / down block on the fp16 stream: the resampled identity skip (unet.py:245-250) is a second output of the same pass (round 6)
        want_skip = bool(mode) and r["cin"] == co and xdt == torch.float16 and x2 is None and os.environ.get("DIFFPURE_SKIP_FUSED", "1") != "0"
        if self._bfuse and r["h2_1"] and ops.deferred_fusable(xa, x2a, G, mode):
            # in_layers' GroupNorm IS the block boundary of the previous convolution (split-K reduction + epilogue + normalisation + operand)
            h, st1, xraw = ops.group_norm_deferred(xa, G, eps, P[n + ".g1"], P[n + ".b1"], x2=x2, act=True, raw=want_raw, want_out=True,
                                                   want_stats=tape is not None)
            x = xa.t
        else:
            x = ops.tensor_of(xa)
            st1 = ops.group_norm_stats(xa, G, eps, x2a)
            h = ops.group_norm(x, G, eps, P[n + ".g1"], P[n + ".b1"], x2=x2, act=True, resample=mode, split=r["h2_1"] and self._ofmt, stats=st1,
                               raw=want_raw or want_skip)
            if want_raw:
                h, xraw = h
            elif want_skip:
                h, xskip = h
        # (below 64 pixels per sample the column records straddle samples and
        #  GroupNorm reduces the tensor itself, which it reads as fp32)
        mid16 = self._lean and (tape is None or self._tape16) and r["h2_1"] and r["h2_2"] and (ho * wo) % 64 == 0
        off = r["emb_off"]
        film = (film_table[:, off:off + co], film_table[:, off + co:off + 2 * co])
        if self._bfuse and r["h2_1"] and r["h2_2"] and ops.conv_defers(ho, wo, 3, r["cin"], co) and ops.splitk_gn_ok(ho, wo, co, 0, G):
            # in_layers' convolution -> out_layers' GroupNorm (FiLM) inside the block: the tensor between them is written only for a tape
            dd = conv1(h, P[n + ".w1"], co, 3, bias=P[n + ".c1"], defer=True, **({"out_f16": True} if mid16 else {}))
            h2, st2, _ = ops.group_norm_deferred(dd, G, eps, P[n + ".g2"], P[n + ".b2"], film=film, act=True, want_out=tape is not None,
                                                 want_stats=tape is not None)
            if tape is not None:
                tape.append(dict(r=r, x=x, x2=x2, st1=st1, hmid=dd.t, st2=st2, film=film))
            h = h2
        else:
            h = conv1(h, P[n + ".w1"], co, 3, bias=P[n + ".c1"], colstats=True, **({"out_f16": True} if mid16 else {}))
            st2 = ops.group_norm_stats(h, G, eps)
            h = h.t
            if tape is not None:
                tape.append(dict(r=r, x=x, x2=x2, st1=st1, hmid=h, st2=st2, film=film))
            h = ops.group_norm(h, G, eps, P[n + ".g2"], P[n + ".b2"], film=film, act=True, split=r["h2_2"] and self._ofmt, stats=st2)
        o16 = {"out_f16": True} if out16 else {}
        # the block's output convolution leaves its reduction / epilogue to the NEXT GroupNorm where the level is a split-K one
        last = {"defer": True} if (self._bfuse and r["h2_2"] and ops.conv_defers(ho, wo, 3, co, co)) else {"colstats": True}
        if fused:
            return conv2(h, P[n + ".w2s"], co, 3, bias=P[n + ".c2s"], segs=(x,) if x2 is None else (x, x2), **last, **o16)
        if mode:
            skip = xskip if want_skip else ops.resample(x, mode)
        elif want_raw:
            skip = self._ch2(xraw, P[n + ".ws"], co, 1, bias=P[n + ".cs"], **o16)
        elif r["cin"] != co:
            skip = ops.conv2d(x, P[n + ".ws"], co, 1, bias=P[n + ".cs"], x2=x2)
        else:
            skip = x if x2 is None else torch.cat([x, x2], dim=3)
        return conv2(h, P[n + ".w2"], co, 3, bias=P[n + ".c2"], res=skip, **last, **o16)

    def _attn(self, r, xa, tape=None):
        P, n, c = self.p, r["name"], r["ch"]
        b, hh, ww, _ = xa.shape
        G = self.GN_GROUPS
        if self._bfuse and r["h2"] and ops.deferred_fusable(xa, None, G):      # the attention block's GroupNorm as the block boundary (8 x 8)
            xn, st, _ = ops.group_norm_deferred(xa, G, self.GN_EPS, P[n + ".g"], P[n + ".b"], act=False, want_out=True, want_stats=tape is not None)
            x = xa.t
        else:
            x = ops.tensor_of(xa)
            st = ops.group_norm_stats(xa, G, self.GN_EPS)
            xn = ops.group_norm(x, G, self.GN_EPS, P[n + ".g"], P[n + ".b"], split=r["h2"] and self._ofmt, stats=st)
        # fp16 x fp16 modes (with or without a tape: the backward pass recomputes the probabilities from the taped qkv): qkv is stored as plain fp16 by the
        # convolution and the flash kernel runs ONE fp16 pass on it, Q and K read in place (csrc/attention.hip; the arithmetic of the
        # reference's use_fp16 attention, unet.py:358-361)
        fused = bool(r.get("proj16")) and ops.attention_fused_ok(hh * ww, c // r["heads"])
        q16 = fused and (tape is None or self._tape16) and os.environ.get("DIFFPURE_ATTN16", "1") != "0"
        qkv = (self._ch2 if r["h2"] else ops.conv2d)(xn, P[n + ".wqkv"], 3 * c, 1, bias=P[n + ".cqkv"], **({"out_f16": True} if q16 else {}))
        layout = "split" if self.cfg["use_new_attention_order"] else "legacy"
        # The taped forward keeps only qkv: the [B*heads, T, T] probabilities (2.1 GB per 32x32 layer at B=64) are
        # RECOMPUTED per block in the backward pass, as the reference does by checkpointing exactly these blocks
        # (guided_diffusion/unet.py:305) - so the forward runs the fused flash kernel with or without a tape.
        if tape is not None:
            tape.append(dict(r=r, x=x, st=st, qkv=qkv, layout=layout))
        if fused:
            ah = ops.attention_fused(qkv.view(b, hh * ww, 3 * c), r["heads"], layout, operand_hw=(hh, ww))
            last = {"defer": True} if (self._bfuse and ops.conv_defers(hh, ww, 1, c, c)) else {"colstats": True}
            return self._ch2(ah, P[n + ".wproj16"], c, 1, bias=P[n + ".cproj"], res=x, **last,
                             **({"out_f16": True} if self._o16(hh * ww, tape) else {}))
        a = ops.attention(qkv.view(b, hh * ww, 3 * c), r["heads"], layout)
        # shapes the fused kernel does not cover: proj_out on the fp32 path (fp32 residual), the result in the stream's format of this level
        return ops.conv2d(a.view(b, hh, ww, c), P[n + ".wproj"], c, 1, bias=P[n + ".cproj"], res=x.float() if x.dtype == torch.float16 else x,
                          colstats=True, out_f16=self._o16(hh * ww, tape))

    def _run(self, blk, h, h2, film, tape=None):
        for r in blk:
            if r["kind"] == "res":
                h = self._res(r, h, h2, film, tape)
                h2 = None
            else:
                h = self._attn(r, h, tape)
        return h

    # -- forward ---------------------------------------------------------------------------------
    def time_table(self, timesteps):
        """timesteps: float32 GPU tensor [R] (R = 1 for a batch-uniform step, else B).
        -> FiLM table [R, sum(2*cout)]: (scale | shift) rows of every ResBlock."""
        P = self.p
        e = ops.timestep_embedding(timesteps, self.freqs, cos_first=True)
        ed = 4 * self.cfg["model_channels"]
        e = ops.linear(e, P["te0.w"], ed, P["te0.b"])
        e = ops.linear(ops.silu(e), P["te2.w"], ed, P["te2.b"])
        return ops.linear(ops.silu(e), P["emb.w"], self.emb_cols, P["emb.b"])

    def forward(self, x, timesteps=None, table_row=None, tape=None):
        """x: [B, H, W, 3] NHWC fp32 on the GPU; timesteps: [1] or [B] float32 (integer-valued for
        the SDE path, unet.py:642-671 receives `(s*1000).long()`), or `table_row` = precomputed
        rows of `time_table` ([1, cols] broadcast over the batch, or [B, cols])."""
        if not self.p:
            raise RuntimeError("GuidedUNet: weights not loaded")
        film = table_row if table_row is not None else self.time_table(timesteps)
        P = self.p
        hs = []
        stem = self.plan["inp"][0][0]
        o16 = self._o16(x.shape[1] * x.shape[2], tape)
        if (stem["name"] + ".w16") in P and ops.conv2d_stem_ok(x.shape[3], x.shape[0], x.shape[1], x.shape[2], stem["cout"]):
            h = ops.conv2d_stem(x, P[stem["name"] + ".w16"], stem["cout"], bias=P[stem["name"] + ".b"], colstats=True, out_f16=o16)
        else:
            h = ops.conv2d(x, P[stem["name"] + ".w"], stem["cout"], 3, bias=P[stem["name"] + ".b"], colstats=True, out_f16=o16)
        hs.append(h)
        for blk in self.plan["inp"][1:]:
            h = self._run(blk, h, None, film, tape)
            hs.append(h)
        h = self._run(self.plan["mid"], h, None, film, tape)
        for blk in self.plan["out"]:
            h = self._run(blk, h, hs.pop(), film, tape)
        h = ops.resolved(h)
        st = ops.group_norm_stats(h, self.GN_GROUPS, self.GN_EPS)
        h = ops.tensor_of(h)
        if tape is not None:
            tape.append(dict(head=True, x=h, st=st))
        h = ops.group_norm(h, self.GN_GROUPS, self.GN_EPS, P["out.g"], P["out.b"], act=True, split=self._out_h2 and self._ofmt, stats=st)
        return (self._ch2 if self._out_h2 else ops.conv2d)(h, P["out.w"], self.cfg["out_channels"], 3, bias=P["out.c"])

    __call__ = forward

    # -- input gradient (vector-Jacobian product) ------------------------------------------------
    # Only dL/dx is formed: parameter gradients are never needed by an attack, and the time-
    # conditioning path does not depend on x.  Every convolution's input gradient is the SAME
    # implicit-GEMM kernel run with a flipped/transposed weight panel (packed once, lazily).
    def enable_grad(self):
        if self._grad_ready:
            return self
        sd, dev, P = self._sd, self.device, self.p

        # fp16 x fp16 modes: one-pass fp16 dgrad convolutions on pooled fp16 dgrad panels (see NCSNpp.enable_grad);
        # DIFFPURE_GRAD16=0 keeps the three-pass gradients
        gpool = None
        if self._pool is not None and os.environ.get("DIFFPURE_GRAD16", "1") != "0":
            gpool = ops.WeightPool(dev, stochastic=self.precision == "f16sr", seed=0x6AAD5EED)
        pooled = []

        def dg(key, n_in_dgrad, lo=None, hi=None, out_lo=None, out_hi=None, name=None):
            """dgrad panel of conv `key`; [lo:hi) selects input channels of the forward conv (= output
            columns of the dgrad), [out_lo:out_hi) output channels of the forward (= dgrad input)."""
            w = sd[key].detach().float()
            if out_lo is not None:
                w = w[out_lo:out_hi]
            wd = ops.dgrad_weight(w)                       # [I, O, kh, kw]
            if lo is not None:
                wd = wd[lo:hi]
            if gpool is not None and n_in_dgrad % 32 == 0 and name is not None:
                gpool.add(name, wd)
                pooled.append(name)
                return None, "h1"
            if self.h2mode and n_in_dgrad % 32 == 0:
                return ops.pack_conv_weight_h2(wd, dev), True
            return ops.pack_conv_weight(wd).to(dev), False

        blocks = [r for b in self.plan["inp"] for r in b] + self.plan["mid"] + [r for b in self.plan["out"] for r in b]
        for r in blocks:
            n = r["name"]
            if r["kind"] == "stem":
                P[n + ".dw"], r["dh2"] = dg(n + ".weight", r["cout"] if gpool is not None else -1, name=n + ".dw")
            elif r["kind"] == "res":
                c1 = r.get("split", r["cin"])
                P[n + ".dw2"], r["dh2_2"] = dg(n + ".out_layers.3.weight", r["cout"], name=n + ".dw2")
                # conv1's dgrad writes the gradient of the (concatenated) normalised input: one panel
                P[n + ".dw1"], r["dh2_1"] = dg(n + ".in_layers.2.weight", r["cout"], name=n + ".dw1")
                if r["cin"] != r["cout"]:
                    g1x1 = r["cout"] if gpool is not None else -1       # the 1x1 dgrads join the fp16 path in the fp16 x fp16 modes only
                    P[n + ".dws1"], r["dh2_s"] = dg(n + ".skip_connection.weight", g1x1, 0, c1, name=n + ".dws1")
                    if c1 != r["cin"]:
                        P[n + ".dws2"], _ = dg(n + ".skip_connection.weight", g1x1, c1, r["cin"], name=n + ".dws2")
            else:
                P[n + ".dwqkv"], r["dh2"] = dg(n + ".qkv.weight", 3 * r["ch"], name=n + ".dwqkv")
                P[n + ".dwproj"], r["dh2_p"] = dg(n + ".proj_out.weight", r["ch"] if gpool is not None else -1, name=n + ".dwproj")
        self._gpool = gpool
        if gpool is not None:
            gpool.finalize()
            # built lazily by the first vjp of a loop, i.e. AFTER that call's reround: take the key the forward's panels carry, so
            # that the first backward step is rounded like every later one (and a repeated call reproduces the first bit for bit)
            if self._pool._last_key is not None:
                gpool.round(self._pool._last_key)
            for name in pooled:
                gpool.bind(P, name, name)
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
        n, co, ci, mode = r["name"], r["cout"], r["cin"], r["mode"]
        G = self.GN_GROUPS
        lean = r["dh2_2"] == "h1"
        dout16 = ops.to_h2(dout, fmt="h1") if lean else dout      # one plain-fp16 copy serves the 3x3 and the 1x1 dgrads
        gs = r.get("dh2_s", False) if lean else False
        dh3 = self._dconv(dout16, n + ".dw2", r["dh2_2"], co, 3)
        dh2, _ = ops.group_norm_bwd(t["hmid"], G, P[n + ".g2"], P[n + ".b2"], t["st2"], dh3, film=t["film"], act=True,
                                    split=r["dh2_1"])
        dh1 = self._dconv(dh2, n + ".dw1", r["dh2_1"], ci, 3)
        # the gradient of the skip branch joins dx inside the GroupNorm backward (ops.group_norm_bwd)
        ad = ad2 = None
        if mode:
            ad = ops.resample_bwd(dout, mode)
        elif ci != co:
            c1 = t["x"].shape[3]
            ad = self._dconv(dout16 if gs else dout, n + ".dws1", gs, c1, 1)
            if t["x2"] is not None:
                ad2 = self._dconv(dout16 if gs else dout, n + ".dws2", gs, ci - c1, 1)
        else:
            ad = dout
        return ops.group_norm_bwd(t["x"], G, P[n + ".g1"], P[n + ".b1"], t["st1"], dh1, x2=t["x2"], act=True, resample=mode, addend=ad, addend2=ad2)

    def _attn_bwd(self, t, dout):
        r, P = t["r"], self.p
        n, c = r["name"], r["ch"]
        b, hh, ww, _ = dout.shape
        da = self._dconv(dout, n + ".dwproj", r.get("dh2_p", False) == "h1" and "h1", c, 1)
        h16_ok = getattr(self, "_gpool", None) is not None and os.environ.get("DIFFPURE_ATTN_BWD16", "1") != "0"
        qkv = t["qkv"]
        # taped on the fp16 stream: the backward reads q, k, v IN PLACE where all five of its products run on dp_gemm_strided_h16 (NCSN++:
        # T = d = 256); where some shape stays on the fp32-input kernel (64-wide heads: N = 64) the tensor is up-converted once (same values)
        if qkv.dtype == torch.float16 and not (h16_ok and ops.attention_h16_serves(hh * ww, c // r["heads"])):
            qkv = qkv.float()
        qkv = qkv.view(b, hh * ww, 3 * c)
        # probabilities recomputed (freed after this block), without the P V product nobody reads; in the fp16 x fp16 modes the
        # products whose shapes dp_gemm_strided_h16 serves (q k^T and dP at head dimension 64) run on the fp16 matrix cores
        h16 = h16_ok
        _, probs = ops.attention(qkv, r["heads"], t["layout"], probs_only=True, h16=h16)
        dqkv = ops.attention_bwd(qkv, probs, da.view(b, hh * ww, c), r["heads"], t["layout"], h16=h16)
        del probs
        dxn = self._dconv(dqkv.view(b, hh, ww, 3 * c), n + ".dwqkv", r["dh2"], c, 1)
        return ops.group_norm_bwd(t["x"], self.GN_GROUPS, P[n + ".g"], P[n + ".b"], t["st"], dxn, addend=dout)[0]

    def vjp(self, tape, dout):
        """(d eps / d x)^T dout for the forward that filled `tape`. dout: [B,H,W,Cd] fp32 cotangent on the
        FIRST Cd output channels (Cd = 3: the eps half of a learn_sigma network; the rest is zero)."""
        self.enable_grad()
        P, sd = self.p, self._sd
        cd = dout.shape[3]
        key = f"out.dw{cd}"
        if key not in P:       # head dgrad restricted to the first cd output channels
            P[key] = ops.pack_conv_weight(ops.dgrad_weight(sd["out.2.weight"].detach().float()[:cd])).to(self.device)
        tape = list(tape)
        head = tape.pop()
        dh = ops.conv2d(dout, P[key], self.plan["final_ch"], 3)
        dh, _ = ops.group_norm_bwd(head["x"], self.GN_GROUPS, P["out.g"], P["out.b"], head["st"], dh, act=True)
        skips = []             # gradients of the skip tensors, in the order the up path consumed them
        for blk in reversed(self.plan["out"]):
            for r in reversed(blk):
                t = tape.pop()
                assert t["r"] is r
                if r["kind"] == "res":
                    dh, d2 = self._res_bwd(t, dh)
                    if d2 is not None:
                        skips.append(d2)
                else:
                    dh = self._attn_bwd(t, dh)
        # skips[0] belongs to the LAST out block = hs[0] (the stem output); skips[-1] to hs[-1]
        for r in reversed(self.plan["mid"]):
            t = tape.pop()
            dh = self._res_bwd(t, dh)[0] if r["kind"] == "res" else self._attn_bwd(t, dh)
        for blk in reversed(self.plan["inp"][1:]):
            dh = ops.add(dh, skips.pop())
            for r in reversed(blk):
                t = tape.pop()
                dh = self._res_bwd(t, dh)[0] if r["kind"] == "res" else self._attn_bwd(t, dh)
        dh = ops.add(dh, skips.pop())
        assert not skips and not tape
        stem = self.plan["inp"][0][0]
        return self._dconv(dh, stem["name"] + ".dw", stem.get("dh2", False) == "h1" and "h1", stem["cin"], 3)
