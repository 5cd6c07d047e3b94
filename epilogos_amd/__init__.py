"""epilogos_amd -- MI355X (gfx950) scoring engine for epilogos: hand-written HIP kernels behind a C ABI
(include/epilogos_amd.h), a ctypes binding (_abi.py), device plumbing on PyTorch-ROCm (engine.py) and host-side
mirrors of the reference's stage drivers (expected.py, expectedCombination.py, scores.py, run.py)."""
__version__ = "0.1.0"
