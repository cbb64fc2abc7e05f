"""MI355X engine for the CelebA-HQ DDPM UNet (`--diffusion_type celebahq-ddpm`).

Same function as ddpm.unet_ddpm.Model.forward (/root/reference/ddpm/unet_ddpm.py:304-345, topology :208-302);
loads the reference's state_dict keys (`celeba_hq.ckpt`).  Every block maps onto the kernels of the two
main engines:

  ResnetBlock (:126-149)  = gn_stats -> gn_apply(+SiLU) -> conv3x3(+bias, + temb_proj(SiLU(temb)) row)
                            -> gn_stats -> gn_apply(+SiLU) -> conv3x3(+bias, + [x | nin_shortcut 1x1(x)] residual)
  AttnBlock (:181-205)    = gn -> ONE 1x1 GEMM for q|k|v -> softmax(q k^T / sqrt(C)) v -> proj_out 1x1 (+x)
  Downsample (:78-87)     = pad (0,1,0,1) + 3x3 stride-2 convolution == the odd rows/columns of the same-padded
                            stride-1 convolution (out[i,j] = z[2i+1, 2j+1]); run at full resolution on the matrix
                            cores and sub-sampled (5 of the ~60 convolutions, 4x their own work)
  Upsample (:57-63)       = nearest x2 folded into the operand packer, then conv3x3
  skip concatenations     = never materialised (dual-source GroupNorm / convolution)
  all temb_proj layers    = one packed GEMM; the whole [steps, sum(cout)] table is built once per call
"""
import math
from collections import OrderedDict

import functools

import torch

from . import ops

GN_GROUPS = 32
GN_EPS = 1e-6


def parse_config(cfg):
    """cfg: dict with 'model' and 'data' sections (configs/celeba.yml)."""
    m, d = cfg["model"], cfg["data"]
    if m.get("type", "simple") != "simple" or not m["resamp_with_conv"]:
        raise NotImplementedError("engine covers the DiffPure CelebA-HQ DDPM family (configs/celeba.yml)")
    return dict(ch=int(m["ch"]), out_ch=int(m["out_ch"]), ch_mult=tuple(m["ch_mult"]), num_res_blocks=int(m["num_res_blocks"]),
                attn_resolutions=tuple(m["attn_resolutions"]), in_channels=int(m["in_channels"]),
                resolution=int(d["image_size"]))


def _plan(cfg):
    """Flat program: records in execution order, names = state_dict prefixes."""
    ch, nrb, mults = cfg["ch"], cfg["num_res_blocks"], cfg["ch_mult"]
    nres = len(mults)
    in_mult = (1,) + tuple(mults)
    res = cfg["resolution"]
    down, hs_c = [], [ch]
    cin = None
    for lvl in range(nres):
        cin = ch * in_mult[lvl]
        cout = ch * mults[lvl]
        for ib in range(nrb):
            blk = [dict(kind="res", name=f"down.{lvl}.block.{ib}", cin=cin, cout=cout)]
            cin = cout
            if res in cfg["attn_resolutions"]:
                blk.append(dict(kind="attn", name=f"down.{lvl}.attn.{ib}", ch=cin))
            down.append(blk)
            hs_c.append(cin)
        if lvl != nres - 1:
            down.append([dict(kind="down", name=f"down.{lvl}.downsample.conv", ch=cin)])
            hs_c.append(cin)
            res //= 2
    mid = [dict(kind="res", name="mid.block_1", cin=cin, cout=cin), dict(kind="attn", name="mid.attn_1", ch=cin),
           dict(kind="res", name="mid.block_2", cin=cin, cout=cin)]
    up = []
    for lvl in reversed(range(nres)):
        cout = ch * mults[lvl]
        for ib in range(nrb + 1):
            sk = hs_c.pop()
            up.append(dict(kind="res", name=f"up.{lvl}.block.{ib}", cin=cin + sk, cout=cout, pop=True, c1=cin))
            cin = cout
            if res in cfg["attn_resolutions"]:
                up.append(dict(kind="attn", name=f"up.{lvl}.attn.{ib}", ch=cin))
        if lvl != 0:
            up.append(dict(kind="up", name=f"up.{lvl}.upsample.conv", ch=cin))
            res *= 2
    assert not hs_c
    return dict(down=down, mid=mid, up=up, final_ch=cin)


def param_shapes(cfg):
    """state_dict key -> shape (the keys of ddpm.unet_ddpm.Model.state_dict())."""
    ch = cfg["ch"]
    plan = _plan(cfg)
    sh = OrderedDict()
    sh["temb.dense.0.weight"], sh["temb.dense.0.bias"] = (4 * ch, ch), (4 * ch,)
    sh["temb.dense.1.weight"], sh["temb.dense.1.bias"] = (4 * ch, 4 * ch), (4 * ch,)
    sh["conv_in.weight"], sh["conv_in.bias"] = (ch, cfg["in_channels"], 3, 3), (ch,)
    recs = [r for b in plan["down"] for r in b] + plan["mid"] + plan["up"]

    def module_order(r):
        """position of the record's module in Model.state_dict(): down.L / mid / up.L containers in construction order
        (`up` levels ascending: __init__ prepends them, :296), inside a level all blocks, then all attns, then the resampler"""
        part = r["name"].split(".")
        top = {"down": 0, "mid": 1, "up": 2}[part[0]]
        if top == 1:
            return (1, 0, {"block_1": 0, "attn_1": 1, "block_2": 2}[part[1]], 0)
        return (top, int(part[1]), {"block": 0, "attn": 1, "downsample": 2, "upsample": 2}[part[2]], int(part[3]) if part[3].isdigit() else 0)

    for r in sorted(recs, key=module_order):
        p = r["name"]
        if r["kind"] == "res":
            ci, co = r["cin"], r["cout"]
            sh[p + ".norm1.weight"], sh[p + ".norm1.bias"] = (ci,), (ci,)
            sh[p + ".conv1.weight"], sh[p + ".conv1.bias"] = (co, ci, 3, 3), (co,)
            sh[p + ".temb_proj.weight"], sh[p + ".temb_proj.bias"] = (co, 4 * ch), (co,)
            sh[p + ".norm2.weight"], sh[p + ".norm2.bias"] = (co,), (co,)
            sh[p + ".conv2.weight"], sh[p + ".conv2.bias"] = (co, co, 3, 3), (co,)
            if ci != co:
                sh[p + ".nin_shortcut.weight"], sh[p + ".nin_shortcut.bias"] = (co, ci, 1, 1), (co,)
        elif r["kind"] == "attn":
            c = r["ch"]
            sh[p + ".norm.weight"], sh[p + ".norm.bias"] = (c,), (c,)
            for n in ("q", "k", "v", "proj_out"):
                sh[p + f".{n}.weight"], sh[p + f".{n}.bias"] = (c, c, 1, 1), (c,)
        else:
            c = r["ch"]
            sh[p + ".weight"], sh[p + ".bias"] = (c, c, 3, 3), (c,)
    fc = plan["final_ch"]
    sh["norm_out.weight"], sh["norm_out.bias"] = (fc,), (fc,)
    sh["conv_out.weight"], sh["conv_out.bias"] = (cfg["out_ch"], fc, 3, 3), (cfg["out_ch"],)
    return sh


class DdpmUNet:
    """noise-prediction network: NHWC in, NHWC out ([B, H, W, out_ch]); `timesteps` = integer step index."""

    def __init__(self, cfg, device, precision="f16x3"):
        if precision != "f32" and precision not in ops.H2_MODES:
            raise ValueError(f"unknown precision {precision!r}")
        self.cfg, self.precision, self.device = cfg, precision, torch.device(device)
        # fp16-matrix-core convolution path: MFMA passes per product and the operand format GroupNorm-apply emits
        self.h2mode = precision in ops.H2_MODES
        passes, ofmt = ops.H2_MODES.get(precision, (3, ops.FMT_H2))
        self._ofmt = "h1" if ofmt == ops.FMT_H1 else "h2"
        self._pool = ops.WeightPool(self.device, stochastic=precision == "f16sr") if precision in ops.W16_MODES else None
        self._ch2 = functools.partial(ops.conv2d_h2, passes=passes, w_fmt=1 if self._pool is not None else 0)
        self.plan = _plan(cfg)
        self.p = {}
        half = cfg["ch"] // 2
        e = math.log(10000) / (half - 1)                                   # unet_ddpm.py:24-26, evaluated on the host
        self.freqs = torch.exp(torch.arange(half, dtype=torch.float32) * -e).to(self.device)
        self.dense_cols = 0

    def load_state_dict(self, sd):
        for k, shp in param_shapes(self.cfg).items():
            if k not in sd:
                raise KeyError(f"state_dict is missing {k}")
            if tuple(sd[k].shape) != tuple(shp):
                raise ValueError(f"{k}: expected {shp}, got {tuple(sd[k].shape)}")
        dev, P = self.device, {}

        def vec(k):
            return sd[k].detach().float().contiguous().to(dev)

        def conv_w(k, cin):
            if self.h2mode and cin % 32 == 0:
                return self._pack_h2w(sd[k].detach()), True
            return ops.pack_conv_weight(sd[k].detach()).to(dev), False

        P["t0.w"], P["t0.b"] = ops.pack_linear_weight(sd["temb.dense.0.weight"].detach()).to(dev), vec("temb.dense.0.bias")
        P["t1.w"], P["t1.b"] = ops.pack_linear_weight(sd["temb.dense.1.weight"].detach()).to(dev), vec("temb.dense.1.bias")
        P["stem.w"], P["stem.b"] = ops.pack_conv_weight(sd["conv_in.weight"].detach()).to(dev), vec("conv_in.bias")
        dw, db, off = [], [], 0
        for r in [r for b in self.plan["down"] for r in b] + self.plan["mid"] + self.plan["up"]:
            n = r["name"]
            if r["kind"] == "res":
                P[n + ".g0"], P[n + ".b0"] = vec(n + ".norm1.weight"), vec(n + ".norm1.bias")
                (P[n + ".w0"], r["h2_0"]), P[n + ".c0"] = conv_w(n + ".conv1.weight", r["cin"]), vec(n + ".conv1.bias")
                P[n + ".g1"], P[n + ".b1"] = vec(n + ".norm2.weight"), vec(n + ".norm2.bias")
                (P[n + ".w1"], r["h2_1"]), P[n + ".c1"] = conv_w(n + ".conv2.weight", r["cout"]), vec(n + ".conv2.bias")
                if r["cin"] != r["cout"]:
                    # the 1x1 shortcut reads the RAW block input: GroupNorm-apply emits it in operand form as a second output
                    r["h2_s"] = r["h2_0"] and r.get("c1", r["cin"]) % 8 == 0
                    w2 = sd[n + ".nin_shortcut.weight"].detach()
                    P[n + ".w2"] = self._pack_h2w(w2) if r["h2_s"] else ops.pack_conv_weight(w2).to(dev)
                    P[n + ".c2"] = vec(n + ".nin_shortcut.bias")
                dw.append(sd[n + ".temb_proj.weight"].detach().float())
                db.append(sd[n + ".temb_proj.bias"].detach().float())
                r["dense_off"] = off
                off += r["cout"]
            elif r["kind"] == "attn":
                P[n + ".g"], P[n + ".b"] = vec(n + ".norm.weight"), vec(n + ".norm.bias")
                wq = torch.cat([sd[n + f".{j}.weight"].detach().float() for j in ("q", "k", "v")], dim=0)      # [3C, C, 1, 1]
                r["h2"] = self.h2mode and r["ch"] % 32 == 0
                P[n + ".wqkv"] = self._pack_h2w(wq) if r["h2"] else ops.pack_conv_weight(wq).to(dev)
                P[n + ".cqkv"] = torch.cat([sd[n + f".{j}.bias"].detach().float() for j in ("q", "k", "v")]).contiguous().to(dev)
                P[n + ".w3"], P[n + ".c3"] = ops.pack_conv_weight(sd[n + ".proj_out.weight"].detach()).to(dev), vec(n + ".proj_out.bias")
            else:
                (P[n + ".w"], r["h2"]), P[n + ".c"] = conv_w(n + ".weight", r["ch"]), vec(n + ".bias")
        P["dense.w"] = ops.pack_linear_weight(torch.cat(dw, dim=0)).to(dev)
        P["dense.b"] = torch.cat(db, dim=0).contiguous().to(dev)
        self.dense_cols = off
        P["out.g"], P["out.b"] = vec("norm_out.weight"), vec("norm_out.bias")
        (P["out.w"], self._out_h2), P["out.c"] = conv_w("conv_out.weight", self.plan["final_ch"]), vec("conv_out.bias")
        self._resolve_pool(P)
        self.p = P
        return self

    # -- blocks ---------------------------------------------------------------------------------------
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

    def _res(self, r, xa, x2a, dense):
        """xa, x2a: ops.Act (tensor + the column statistics its producing convolution left) or plain tensors"""
        x, x2 = ops.tensor_of(xa), ops.tensor_of(x2a)
        P, n, co = self.p, r["name"], r["cout"]
        conv0 = self._ch2 if r["h2_0"] else ops.conv2d
        conv1 = self._ch2 if r["h2_1"] else ops.conv2d
        st0 = ops.group_norm_stats(xa, GN_GROUPS, GN_EPS, x2a)
        want_raw = r.get("h2_s", False)
        h = ops.group_norm(x, GN_GROUPS, GN_EPS, P[n + ".g0"], P[n + ".b0"], x2=x2, act=True, split=r["h2_0"] and self._ofmt, stats=st0,
                           raw=want_raw)
        if want_raw:
            h, xraw = h
        off = r["dense_off"]
        h = conv0(h, P[n + ".w0"], co, 3, bias=P[n + ".c0"], temb=dense[:, off:off + co], colstats=True)
        st1 = ops.group_norm_stats(h, GN_GROUPS, GN_EPS)
        h = h.t
        h = ops.group_norm(h, GN_GROUPS, GN_EPS, P[n + ".g1"], P[n + ".b1"], act=True, split=r["h2_1"] and self._ofmt, stats=st1)
        if want_raw:
            skip = self._ch2(xraw, P[n + ".w2"], co, 1, bias=P[n + ".c2"])
        elif r["cin"] != co:
            skip = ops.conv2d(x, P[n + ".w2"], co, 1, bias=P[n + ".c2"], x2=x2)
        else:
            skip = x if x2 is None else torch.cat([x, x2], dim=3)
        return conv1(h, P[n + ".w1"], co, 3, bias=P[n + ".c1"], res=skip, colstats=True)

    def _attn(self, r, xa):
        P, n, c = self.p, r["name"], r["ch"]
        x = ops.tensor_of(xa)
        b, hh, ww, _ = x.shape
        st = ops.group_norm_stats(xa, GN_GROUPS, GN_EPS)
        hn = ops.group_norm(x, GN_GROUPS, GN_EPS, P[n + ".g"], P[n + ".b"], split=r["h2"] and self._ofmt, stats=st)
        qkv = (self._ch2 if r["h2"] else ops.conv2d)(hn, P[n + ".wqkv"], 3 * c, 1, bias=P[n + ".cqkv"])
        a = ops.attention(qkv.view(b, hh * ww, 3 * c), 1, "split")           # one head of dimension C, scale C^-1/2
        return ops.conv2d(a.view(b, hh, ww, c), P[n + ".w3"], c, 1, bias=P[n + ".c3"], res=x, colstats=True)

    def _down(self, r, x):
        """pad (0,1,0,1) + 3x3 stride 2 == positions (2i+1, 2j+1) of the same-padded stride-1 convolution."""
        P, n, c = self.p, r["name"], r["ch"]
        x = ops.tensor_of(x)
        if r["h2"]:
            z = self._ch2(ops.to_h2(x, fmt=self._ofmt), P[n + ".w"], c, 3, bias=P[n + ".c"])
        else:
            z = ops.conv2d(x, P[n + ".w"], c, 3, bias=P[n + ".c"])
        return z[:, 1::2, 1::2, :].contiguous()

    def _up(self, r, x):
        P, n, c = self.p, r["name"], r["ch"]
        x = ops.tensor_of(x)
        if r["h2"]:
            return self._ch2(ops.to_h2(x, ops.RESAMPLE_UP, fmt=self._ofmt), P[n + ".w"], c, 3, bias=P[n + ".c"], colstats=True)
        return ops.conv2d(ops.resample(x, ops.RESAMPLE_UP), P[n + ".w"], c, 3, bias=P[n + ".c"], colstats=True)

    # -- time conditioning ------------------------------------------------------------------------------
    def time_table(self, timesteps):
        """timesteps: float32 GPU tensor [R] (integer values). -> temb_proj rows of every ResnetBlock [R, sum(cout)]."""
        P, ch = self.p, self.cfg["ch"]
        e = ops.timestep_embedding(timesteps, self.freqs, cos_first=False)
        e = ops.linear(e, P["t0.w"], 4 * ch, P["t0.b"])
        e = ops.linear(ops.silu(e), P["t1.w"], 4 * ch, P["t1.b"])
        return ops.linear(ops.silu(e), P["dense.w"], self.dense_cols, P["dense.b"])

    def forward(self, x, timesteps=None, table_row=None, tape=None):
        if not self.p:
            raise RuntimeError("DdpmUNet: weights not loaded")
        if tape is not None:
            raise NotImplementedError("the CelebA-HQ DDPM runner is not differentiated upstream (torch.no_grad, diffpure_ddpm.py:104)")
        P = self.p
        dense = table_row if table_row is not None else self.time_table(timesteps)
        hs = [ops.conv2d(x, P["stem.w"], self.cfg["ch"], 3, bias=P["stem.b"], colstats=True)]
        for blk in self.plan["down"]:
            h = hs[-1]
            for r in blk:
                h = self._res(r, h, None, dense) if r["kind"] == "res" else (self._attn(r, h) if r["kind"] == "attn" else self._down(r, h))
            hs.append(h)
        h = hs[-1]
        for r in self.plan["mid"]:
            h = self._res(r, h, None, dense) if r["kind"] == "res" else self._attn(r, h)
        for r in self.plan["up"]:
            if r["kind"] == "attn":
                h = self._attn(r, h)
            elif r["kind"] == "up":
                h = self._up(r, h)
            else:
                h = self._res(r, h, hs.pop(), dense)
        assert not hs
        sth = ops.group_norm_stats(h, GN_GROUPS, GN_EPS)
        h = ops.tensor_of(h)
        h = ops.group_norm(h, GN_GROUPS, GN_EPS, P["out.g"], P["out.b"], act=True, split=self._out_h2 and self._ofmt, stats=sth)
        conv = self._ch2 if self._out_h2 else ops.conv2d
        return conv(h, P["out.w"], self.cfg["out_ch"], 3, bias=P["out.c"])
