"""TEST INFRASTRUCTURE - CPU restatement of the steps around the purifier in SDE_Adv_Model.forward
(/root/reference/eval_sde_adv.py:73-89).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this package.

Pinned by construction: the two operators involved are torch's own `F.interpolate(mode='bilinear',
align_corners=False)` and elementwise affine maps, i.e. the very calls the reference makes.
"""
import torch
import torch.nn.functional as F


def pre(x01, diffusion_size=None):
    """eval_sde_adv.py:74-75 (only when 'imagenet' in args.domain -> pass diffusion_size=(256, 256)) and :78."""
    if diffusion_size is not None:
        x01 = F.interpolate(x01, size=tuple(diffusion_size), mode="bilinear", align_corners=False)
    return (x01 - 0.5) * 2


def post(x_re, classifier_size=None):
    """eval_sde_adv.py:81-82 (imagenet only -> pass classifier_size=(224, 224)) and :89."""
    if classifier_size is not None:
        x_re = F.interpolate(x_re, size=tuple(classifier_size), mode="bilinear", align_corners=False)
    return (x_re + 1) * 0.5


def sde_adv_forward(purify, classifier, x01, diffusion_size=None):
    """classifier(post(purify(pre(x)))) - `purify` is image_editing_sample on [-1,1] NCHW images."""
    size_c = tuple(x01.shape[2:]) if diffusion_size is not None else None
    return classifier(post(purify(pre(x01, diffusion_size)), size_c))
