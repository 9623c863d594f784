"""3dgs_amd -- MI355X-native differentiable gaussian-splat rasterizer (host-side binding).

The product is the C-ABI shared library built from ``csrc/`` (see ``include/gsplat_hip.h``);
this package is the thin ctypes binding used by the tests, the benchmark and the
view-sharded multi-GPU driver.  The directory name starts with a digit, so import it with
``importlib.import_module("3dgs_amd")``.
"""
