"""Import the reference's own score networks from oracle/_ref (oracle/make_ref.py).  TEST INFRASTRUCTURE ONLY: bench.py's
`cpu_baseline` leg and tests/ - never the product.

The copy is used only if EVERY file matches the tracked digests (oracle/ref_modules.sha256).  Packages the reference imports
but this image lacks are stubbed exactly as tests/golden/make_golden.py stubs them: `score_sde.op` (the CUDA upfirdn2d extension,
unused with fir: False)."""
import hashlib
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "_ref")
SHA = os.path.join(HERE, "ref_modules.sha256")


def available():
    """True iff oracle/_ref holds every file of the manifest, byte for byte."""
    if not (os.path.isdir(REF) and os.path.exists(SHA)):
        return False
    for line in open(SHA):
        digest, rel = line.split()
        p = os.path.join(REF, rel)
        if not os.path.exists(p) or hashlib.sha256(open(p, "rb").read()).hexdigest() != digest:
            return False
    return True


def _enter():
    if REF not in sys.path:
        sys.path.insert(0, REF)
    if "score_sde.op" not in sys.modules:
        op = types.ModuleType("score_sde.op")
        op.upfirdn2d = None
        sys.modules["score_sde.op"] = op


def _load(module, sd):
    missing = module.load_state_dict(sd, strict=False)
    assert set(missing.missing_keys) <= {"sigmas"} and not missing.unexpected_keys, missing
    return module.eval()


def guided_unet(model_cfg, state_dict, use_fp16=False):
    """guided_diffusion.script_util.create_model(**cfg) with `state_dict` loaded, eval mode; fp32 (use_fp16=False), or the reference's own
    fp16 torso - created with use_fp16=True and converted as its runners do (runners/diffpure_sde.py:169-170 -> unet.py:626-632)."""
    assert available(), "oracle/_ref is missing or does not match oracle/ref_modules.sha256"
    _enter()
    from guided_diffusion.script_util import create_model, model_and_diffusion_defaults
    mc = model_and_diffusion_defaults()
    mc.update(model_cfg)
    mc["use_fp16"] = bool(use_fp16)
    keys = ("image_size", "num_channels", "num_res_blocks", "channel_mult", "learn_sigma", "class_cond", "attention_resolutions",
            "num_heads", "num_head_channels", "num_heads_upsample", "use_scale_shift_norm", "resblock_updown", "use_fp16",
            "use_new_attention_order")
    model = _load(create_model(**{k: mc[k] for k in keys}), state_dict)
    if use_fp16:
        model.convert_to_fp16()
    return model


def ncsnpp(config_dict, state_dict):
    """score_sde.models.utils.create_model(config) (NCSNpp) with `state_dict` loaded, eval mode."""
    assert available(), "oracle/_ref is missing or does not match oracle/ref_modules.sha256"
    _enter()
    import argparse

    def d2n(c):
        ns = argparse.Namespace()
        for k, v in c.items():
            setattr(ns, k, d2n(v) if isinstance(v, dict) else v)
        return ns

    from score_sde.models import utils as mutils
    cfg = d2n(config_dict)
    if not hasattr(cfg, "device"):
        import torch
        cfg.device = torch.device("cpu")
    model = mutils.create_model(cfg)
    model = getattr(model, "module", model)
    return _load(model, state_dict)
