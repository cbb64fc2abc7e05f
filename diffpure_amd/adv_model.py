"""MI355X counterpart of `SDE_Adv_Model` (/root/reference/eval_sde_adv.py:31-93): classifier(purify(x)).

SURVEY.md section 8f-2 - "the steps either side of the path": upstream, `forward` runs
    F.interpolate(x, 256, bilinear)  ->  (x - 0.5) * 2  ->  runner.image_editing_sample  ->
    F.interpolate(x_re, 224, bilinear)  ->  (x_re + 1) * 0.5  ->  classifier
(:73-89).  Here each side is ONE HIP kernel (`dp_resize_affine`) that also does the NCHW <-> NHWC repack
the engine needs, the purifier is entered and left in its native NHWC state (`nhwc=True`), and the
whole thing stays differentiable w.r.t. `x` (adjoint resize kernel + the adjoint of the runner), which
is what AutoAttack / BPDA+EOT differentiate through.

`forward(x, mode=...)` also speaks the BPDA+EOT driver's dialect (SURVEY.md section 8f-3,
eval_sde_adv_bpda.py:83-118: modes 'purify' / 'classify' / 'purify_and_classify', `.resnet`), whose
`purify` calls arrive as `eot_defense_reps` (150) replicas per image in one batch.
`forward_eot(x, reps)` serves the EOT repeats of an attack (`eot_iter`, eval_sde_adv.py:148-149) as one
batch of reps*B independent purifications instead of `reps` sequential calls.

Same constructor fields as upstream: args.{diffusion_type, domain, classifier_name, ...}, config.device.
The classifier zoo (utils.get_image_classifier: torchvision / robustbench models) is outside the scope
contract; pass any `nn.Module` taking [0,1] NCHW images, or leave it None inside a DiffPure checkout
where `utils.get_image_classifier` is importable.
"""
import time

import torch

from . import ops


class _ResizeAffine(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, size, shift, scale, in_nhwc, out_nhwc):
        x = x.contiguous().float()
        ctx.cfg = (ops._img_dims(x.shape, in_nhwc)[2:], scale, in_nhwc, out_nhwc)
        return ops.resize_affine(x, size, shift, scale, in_nhwc, out_nhwc)

    @staticmethod
    def backward(ctx, g):
        in_size, scale, in_nhwc, out_nhwc = ctx.cfg
        return ops.resize_affine_bwd(g.contiguous().float(), in_size, scale, in_nhwc, out_nhwc), None, None, None, None, None


def resize_affine(x, size, shift, scale, in_nhwc=False, out_nhwc=False):
    """Differentiable (bilinear(x) + shift) * scale on the HIP engine."""
    return _ResizeAffine.apply(x, tuple(size), float(shift), float(scale), bool(in_nhwc), bool(out_nhwc))


def build_runner(args, config, device):
    """The runner dispatch of eval_sde_adv.py:43-56."""
    if args.diffusion_type == "ddpm":
        from runners.diffpure_guided import GuidedDiffusion
        return GuidedDiffusion(args, config, device=device)
    if args.diffusion_type == "sde":
        from runners.diffpure_sde import RevGuidedDiffusion
        return RevGuidedDiffusion(args, config, device=device)
    if args.diffusion_type == "ode":
        from runners.diffpure_ode import OdeGuidedDiffusion
        return OdeGuidedDiffusion(args, config, device=device)
    if args.diffusion_type == "ldsde":
        from runners.diffpure_ldsde import LDGuidedDiffusion
        return LDGuidedDiffusion(args, config, device=device)
    if args.diffusion_type == "celebahq-ddpm":
        from runners.diffpure_ddpm import Diffusion
        return Diffusion(args, config, device=device)
    raise NotImplementedError("unknown diffusion type")


class SDE_Adv_Model(torch.nn.Module):
    def __init__(self, args, config, classifier=None, runner=None):
        super().__init__()
        self.args = args
        device = getattr(config, "device", None) or torch.device("cuda")
        self.device = torch.device(device)
        if classifier is None:
            from utils import get_image_classifier      # DiffPure's own helper (needs torchvision)
            classifier = get_image_classifier(args.classifier_name)
        self.classifier = classifier.to(self.device)
        print(f"diffusion_type: {args.diffusion_type}")
        self.runner = runner if runner is not None else build_runner(args, config, self.device)
        self.register_buffer("counter", torch.zeros(1, device=self.device))
        self.tag = None

    def reset_counter(self):
        self.counter = torch.zeros(1, dtype=torch.int, device=self.device)

    def set_tag(self, tag=None):
        self.tag = tag

    def purify(self, x, bs_id=0):
        """x in [0,1], NCHW at the classifier's resolution -> purified image in [0,1], same shape
        (times args.sample_step along the batch).  Diffusion resolution as upstream (:73-75): 256x256 when
        'imagenet' is in args.domain, otherwise the input's own (both resize steps are then exact
        identities); `args.diffusion_size = (H, W)` overrides it."""
        x = x.to(self.device)
        size_c = tuple(x.shape[2:])
        size_d = getattr(self.args, "diffusion_size", None)
        if size_d is None:
            size_d = (256, 256) if "imagenet" in getattr(self.args, "domain", "") else size_c
        size_d = tuple(size_d)
        state = resize_affine(x, size_d, -0.5, 2.0, in_nhwc=False, out_nhwc=True)             # (x - 0.5) * 2, NHWC
        state = self.runner.image_editing_sample(state, bs_id=bs_id, tag=self.tag, nhwc=True)
        return resize_affine(state, size_c, 1.0, 0.5, in_nhwc=True, out_nhwc=False)           # (x_re + 1) * 0.5, NCHW

    @property
    def resnet(self):
        """the classifier under the name eval_sde_adv_bpda.py uses (:57)"""
        return self.classifier

    def forward(self, x, mode="purify_and_classify"):
        """eval_sde_adv.py:67-93 (`mode` left at its default) and the three modes of the BPDA+EOT driver's
        model (eval_sde_adv_bpda.py:83-118): 'purify' -> purified images in [0,1] (what bpda_eot_attack.py:98-101
        calls on `purify_reps` replicas of the batch at once), 'classify' -> classifier(x),
        'purify_and_classify' -> classifier(purify(x))."""
        if mode == "classify":
            return self.classifier(x.to(self.device))
        if mode not in ("purify", "purify_and_classify"):
            raise NotImplementedError(f"unknown mode: {mode}")
        counter = int(self.counter.item())
        if counter % 5 == 0:
            print(f"diffusion times: {counter}")
        start_time = time.time()
        x_re = self.purify(x, bs_id=counter)
        if counter % 5 == 0:
            torch.cuda.synchronize(self.device)
            minutes, seconds = divmod(time.time() - start_time, 60)
            print(f"x shape (before diffusion models): {tuple(x.shape)}")
            print(f"x shape (before classifier): {tuple(x_re.shape)}")
            print("Sampling time per batch: {:0>2}:{:05.2f}".format(int(minutes), seconds))
        self.counter += 1
        return x_re if mode == "purify" else self.classifier(x_re)

    def forward_eot(self, x, reps):
        """logits [reps, B, classes] of `reps` independent purifications of every image, as ONE batch
        (global sample index r*B + b keys the noise, so replicas differ exactly as separate calls would)."""
        b = x.shape[0]
        out = self.forward(x.repeat(reps, 1, 1, 1))
        return out.reshape(reps, b * getattr(self.args, "sample_step", 1), *out.shape[1:])
