"""CPU oracle for the DiffPure purification hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain PyTorch-CPU fp32 restatement of the reference's algorithm for the path
named in BASELINE.json: the two score-network UNets, the reverse VP-SDE
Euler-Maruyama loop, the probability-flow ODE Euler loop (+ its adjoint), and the DDPM ancestral
step.  Every function cites the reference file:line it follows.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it, and
only as the checker / reported CPU baseline - never as the product.  The product
(`diffpure_amd/`, `runners/`) must not import anything from here.

Pinning status: the reference ships no tests or golden vectors.  The oracle
is pinned instead against outputs of the reference's own modules (`guided_diffusion.unet.UNetModel`,
`score_sde.models.ncsnpp.NCSNpp`, `runners.diffpure_sde.RevVPSDE.f/.g`,
`runners.diffpure_ode.VPODE.forward`, `GaussianDiffusion.p_sample`) imported in the build container
from a scratch copy of /root/reference; the generating script and the vectors live in
`tests/golden/`.  The fixed-step solvers themselves live in third-party packages that are absent
(torchsde, unpinned; torchdiffeq==0.2.1): their published Euler / Euler-Maruyama updates are
restated in `oracle/solvers.py` - that part is "parity unpinned" against the packages themselves.
"""
