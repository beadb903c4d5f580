"""monte-carlo-path-tracing_amd — MI355X-native path tracer, host-side Python mirror.

The directory name carries hyphens (it mirrors the reference repository's
name), so it is loaded through `_pkg.load_package()` at the repo root, which
registers it in `sys.modules` as `mcpt_amd`.

Sub-modules:
  mcsd     MCSD scene-description reader/writer (include/mcsd_format.h)
  scenes   programmatic test scenes (cornell box, volumetric caustic, ...)
  capi     ctypes binding of the C-ABI in include/mcpt.h (the HIP renderer)
  tiling   multi-GPU image partition + the single gather of finished tiles
  workloads  the BASELINE.json configurations by name (baseline_scenes/ fixtures)
"""
from . import capi, mcsd, scenes, tiling, workloads  # noqa: F401
