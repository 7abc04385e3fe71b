"""diffeqgmrfs.jl_amd -- MI355X-native block-tridiagonal GMRF factor/solve/sample path.

The directory name carries a dot (it mirrors the reference's name, DiffEqGMRFs.jl), so it
cannot be imported with a plain `import`; use `__graft_entry__.load_package()` (or
`importlib` with an alias), which registers it as `diffeqgmrfs_jl_amd`.

Layout:
    csrc/           hand-written HIP kernels + the C ABI (libgmrf_hip.so, include/gmrf_hip.h)
    _cabi.py        ctypes binding of the C ABI
    api.py          host mirror of the reference's operator surface (tridiagonal_cholesky, ldiv, ...)
    posterior.py    posterior mean / samples / variances and their sharding across ranks
    workloads.py    synthetic FEM precision matrices for the BASELINE configs (inputs only)
"""
from . import _cabi, api, workloads  # noqa: F401
from .api import (BurgersP1Tangent, ShallowWaterP1, Comm, StreamSet, ConditionedGMRF, CsrMatrix, DarcyP1Assembler, GmrfError, NotPositiveDefinite, PosteriorAssembler,  # noqa: F401
                  TridiagonalCholeskyFactor, backward_solve, condition_on_observations, extract_blocks, forward_solve,
                  gn_step, ldiv, ldiv_,
                  logdet, make_chunks, tridiagonal_cholesky)

__all__ = ["TridiagonalCholeskyFactor", "tridiagonal_cholesky", "forward_solve", "backward_solve",
           "ldiv", "ldiv_", "make_chunks", "extract_blocks", "logdet", "CsrMatrix", "GmrfError",
           "NotPositiveDefinite", "PosteriorAssembler", "DarcyP1Assembler", "BurgersP1Tangent", "ShallowWaterP1", "Comm", "StreamSet", "gn_step", "ConditionedGMRF", "condition_on_observations", "workloads", "api"]
