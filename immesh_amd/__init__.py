"""immesh_amd -- MI355X-native implementation of ImMesh's per-scan hot path.

The product is the C-ABI shared library ``immesh_amd/csrc/libimmesh_hip.so`` (declared in
``include/immesh_c_api.h``): hand-written HIP kernels for gfx950 + a C++ host layer.  This Python package is
only the ctypes binding used by tests/ and bench.py plus the synthetic-scan generators; it contains no compute
fallback: if the HIP library is missing or no GPU is usable, loading/creating fails loudly.
"""
from .capi import (Config, HotPath, load_hip_library, hip_library_path, STATE_DOUBLES, make_state, avia_config,
                   velodyne_config, PlaneRec)

__all__ = ["Config", "HotPath", "load_hip_library", "hip_library_path", "STATE_DOUBLES", "make_state", "avia_config",
           "velodyne_config", "PlaneRec"]
