"""diffpure_amd: MI355X (gfx950) engine for DiffPure's diffusion-purification hot path.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed/RCCL); all device
arithmetic runs in hand-written HIP kernels (csrc/, C ABI in include/diffpure_hip.h).
"""
__version__ = "0.1.0"
