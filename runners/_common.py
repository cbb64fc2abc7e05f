"""Shared pieces of the drop-in runners (logging side effects of the reference's
image_editing_sample, kept behind the same `bs_id < 2` guard)."""
import os
import random

import torch


def out_dir_for(args, bs_id, tag):
    if tag is None:
        tag = "rnd" + str(random.randint(0, 10000))
    log_dir = getattr(args, "log_dir", None)
    return None if log_dir is None else os.path.join(log_dir, "bs" + str(bs_id) + "_" + tag)


def save_image(x, path):
    """(x+1)/2 PNG via torchvision when it is installed; always the raw tensor next to it."""
    torch.save(x.detach().cpu(), os.path.splitext(path)[0] + ".pth")
    try:
        import torchvision.utils as tvu  # not installed in the ROCm image; optional
    except Exception:
        return
    tvu.save_image((x + 1) * 0.5, path)


def as_nchw(x, nhwc):
    return x.permute(0, 3, 1, 2) if nhwc else x


def as_nchw_shape(shape, nhwc):
    return (shape[0], shape[3], shape[1], shape[2]) if nhwc else tuple(shape)
