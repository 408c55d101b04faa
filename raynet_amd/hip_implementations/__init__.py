"""HIP counterparts of raynet/cuda_implementations/*.py: closure factories with the
reference's names, argument order and assertions, launching gfx950 kernels through
the C ABI (include/raynet_hip.h) instead of PyCUDA."""
from .context import HipContext, to_device  # noqa: F401

_CONTEXTS = {}


def get_context(M=1, D=2, N=2, F=1, H=1, W=1, padding=0, bbox=(0, 0, 0, 1, 1, 1),
                grid_shape=(1, 1, 1)):
    """One HipContext per configuration tuple (what SourceModule caching is to the
    reference)."""
    import numpy as np
    key = (int(M), int(D), int(N), int(F), int(H), int(W), int(padding),
           tuple(float(b) for b in np.asarray(bbox, dtype=np.float32).ravel()),
           tuple(int(g) for g in np.asarray(grid_shape).ravel()))
    ctx = _CONTEXTS.get(key)
    if ctx is None:
        ctx = HipContext(M, D, N, F, H, W, padding, bbox, grid_shape)
        _CONTEXTS[key] = ctx
    return ctx
