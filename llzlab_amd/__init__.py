"""llzlab_amd -- MI355X-native FIR / IIR / resample / FFT hot path of libllzfilter behind the reference's C API.

The product is llzlab_amd/libllzfilter_hip.so (C ABI in include/*.h; HIP kernels in csrc/kernels, host C in
csrc/host).  The Python modules are a ctypes binding (capi), a mirror of the handle interface (filters) and the
channel-sharding helper for one-process-per-GPU runs (shard)."""
from . import capi  # noqa: F401
from .capi import LlzError  # noqa: F401

__all__ = ["capi", "filters", "shard", "LlzError"]
