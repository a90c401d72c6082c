"""katome_amd -- MI355X-native replacement for katome's `build` stage (k-mer extraction +
de Bruijn graph construction), behind katome's own Build/Config surface.

The compute path is the HIP library katome_amd/lib/libkatome_gpu.so (C ABI: include/katome_gpu.h).
There is no CPU fallback: importing works anywhere, building a graph needs an MI355X.
"""
from .build import (CollectionStats, Config, GpuGraph, InputFileType, KatomePanic,  # noqa: F401
                    set_global_k_sizes)
from ._lib import lib, lib_path  # noqa: F401

__all__ = ["Config", "InputFileType", "GpuGraph", "CollectionStats", "KatomePanic", "set_global_k_sizes", "lib",
           "lib_path"]
