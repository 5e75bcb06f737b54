"""jdet_amd -- MI355X (gfx950) implementation of the rotated-box hot path of Jittor/JDet.

Layout mirrors the reference's python/jdet package for the path in scope:
    jdet_amd.ops.*      <-> python/jdet/ops/*       (same module / class / function names)
    jdet_amd.utils.*    <-> python/jdet/utils/registry.py
Host code is Python on PyTorch-ROCm (device memory + streams only); the arithmetic runs in
hand-written HIP kernels behind the C ABI of include/jdet_hip.h (jdet_amd/csrc).
"""
__version__ = "0.1.0"
