"""gradient-sdf_amd: MI355X-native Gradient-SDF hot path (fusion, voxel-hash query, SDF tracking).

Layout
  csrc/     hand-written gfx950 HIP kernels + the C-ABI (include/gsdf.h) -> libgsdf.so
  host/     C++ facade mirroring the reference classes (MapGradPixelSdf, RigidPointOptimizer,
            SdfVoxel) and the Scan3D CLI, on top of the C-ABI only
  binding.py  ctypes view of the C-ABI for tests/ and bench.py
  synth.py    seeded synthetic depth streams (matlab/RenderSpheres.m restated)

The directory name contains a hyphen (task contract); import it through
``__graft_entry__.package()`` which registers it as ``gradient_sdf_amd``.
"""
from . import binding, parallel, synth  # noqa: F401
from .binding import GradSdf, GsdfError  # noqa: F401
