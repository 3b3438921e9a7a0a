"""g4s_amd — MI355X-native drop-in for the sparse hot path of CGCL-codes/G4S (SpMV, SpGEMM, gather/apply graph interface).

The product is the C-ABI library g4s_amd/lib/libg4s_hip.so (include/g4s.h). This package is the Python view of it used by
tests/ and bench.py: `capi` (ctypes signatures) and `host` (the reference's names — CSR, HashSpGEMM, spmm_dense, GraphProcess —
over device-resident buffers). PyTorch supplies device memory, streams and torch.distributed only.
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
