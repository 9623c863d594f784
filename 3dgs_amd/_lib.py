"""ctypes binding of libgsplat_hip.so (declared in include/gsplat_hip.h).

There is no CPU fallback anywhere in this package: if the shared library cannot be built or
loaded, ``load()`` raises, and so does every operator.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libgsplat_hip.so")
_lib = None


class GsplatError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"gsplat error {code}: {text}")
        self.code = code


def build(force=False):
    """Compile every HIP source for gfx950 with hipcc (works without a GPU).  Serialised across processes by a file
    lock: N ranks importing the package at once (bench.py's children, torchrun, pytest-xdist) would otherwise run
    `make` in the same directory together, and one could dlopen a half-linked library."""
    import fcntl
    srcs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".hip", ".h")) or f == "Makefile"]
    srcs.append(os.path.join(_CSRC, "host", "gs_dataset.cpp"))
    srcs.append(os.path.join(_HERE, "..", "include", "gsplat_hip.h"))

    def stale():
        return force or not os.path.exists(LIB_PATH) or any(
            os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)

    if stale():
        with open(os.path.join(_CSRC, ".build.lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                if stale():  # another process may have built it while this one waited
                    subprocess.check_call(["make", "-C", _CSRC, "-j", "5", "-s"] + (["-B"] if force else []))
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


def build_cpp_host(name):
    """Compile tests/cpp/<name>.cpp -- a host program written against the drop-in headers (include/gsplat_cuda/*.cuh) --
    and link it with the library; returns the executable's path.  Rebuilt when the source, a header or the library is
    newer.  Used by the tests, by bench.py's reference_host_path leg and by __graft_entry__.build()."""
    root = os.path.normpath(os.path.join(_HERE, ".."))
    src = os.path.join(root, "tests", "cpp", name + ".cpp")
    exe = os.path.join(root, "tests", "cpp", name)
    if os.environ.get("GSPLAT_NO_BUILD") == "1":  # a process with an initialised GPU (a rank, a profiled run): no compilers
        if not os.path.exists(exe):
            raise OSError(f"{exe} is missing and GSPLAT_NO_BUILD=1 forbids building it")
        return exe
    lib = build()
    inc = os.path.join(root, "include", "gsplat_cuda")
    deps = [src, lib, os.path.join(root, "include", "gsplat_hip.h")] + [os.path.join(inc, h) for h in os.listdir(inc)]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(p) for p in deps):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-x", "hip", "-I",
                               os.path.join(root, "include"), src, "-x", "none", lib, "-Wl,-rpath," + os.path.dirname(lib),
                               "-o", exe])
    return exe


def source_hash():
    """sha256 (first 16 hex digits) over the device sources the library is built from: ties a stored profile
    (profiles/traffic.json) to the code it was measured on."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(os.listdir(_CSRC)):
        if f.endswith((".hip", ".h")) or f == "Makefile":
            h.update(f.encode())
            with open(os.path.join(_CSRC, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


c_float_p = ctypes.POINTER(ctypes.c_float)


class Camera(ctypes.Structure):  # gsplat_camera
    _fields_ = [("width", ctypes.c_int), ("height", ctypes.c_int), ("focal_x", ctypes.c_float),
                ("focal_y", ctypes.c_float), ("campos", ctypes.c_float * 3), ("view", ctypes.c_void_p),
                ("proj", ctypes.c_void_p)]


class Gaussians(ctypes.Structure):  # gsplat_gaussians
    _fields_ = [("num_gaussians", ctypes.c_int), ("xyz", ctypes.c_void_p), ("rgb", ctypes.c_void_p),
                ("sh", ctypes.c_void_p), ("opacity", ctypes.c_void_p), ("scale", ctypes.c_void_p),
                ("quaternion", ctypes.c_void_p)]


class RasterConfig(ctypes.Structure):  # gsplat_raster_config
    _fields_ = [("near_thresh", ctypes.c_float), ("mh_dist", ctypes.c_float), ("cull_mask_padding", ctypes.c_int)]


class ForwardView(ctypes.Structure):  # gsplat_forward_view
    _fields_ = [("num_culled", ctypes.c_size_t), ("num_pairs", ctypes.c_size_t), ("num_splats", ctypes.c_size_t),
                ("mask", ctypes.c_void_p), ("uv", ctypes.c_void_p), ("xyz_c", ctypes.c_void_p),
                ("compact_to_global", ctypes.c_void_p), ("sigma", ctypes.c_void_p), ("conic", ctypes.c_void_p),
                ("J", ctypes.c_void_p), ("precomputed_rgb", ctypes.c_void_p), ("radius", ctypes.c_void_p),
                ("uv_selected", ctypes.c_void_p), ("xyz_c_selected", ctypes.c_void_p),
                ("sorted_gaussians", ctypes.c_void_p), ("splat_start_end_idx_by_tile_idx", ctypes.c_void_p),
                ("image", ctypes.c_void_p), ("weight_per_pixel", ctypes.c_void_p),
                ("splats_per_pixel", ctypes.c_void_p)]


class Gradients(ctypes.Structure):  # gsplat_gradients
    _fields_ = [(n, ctypes.c_void_p) for n in
                ("grad_xyz", "grad_rgb", "grad_sh", "grad_opacity", "grad_scale", "grad_quaternion", "grad_conic",
                 "grad_uv", "grad_J", "grad_sigma", "grad_xyz_c", "grad_precompute_rgb")]


class AdamFused(ctypes.Structure):  # gsplat_adam_fused
    _fields_ = [("exp_avg", ctypes.c_void_p * 6), ("exp_avg_sq", ctypes.c_void_p * 6), ("lr", ctypes.c_float * 6),
                ("b1", ctypes.c_float), ("b2", ctypes.c_float), ("eps", ctypes.c_float), ("bias1", ctypes.c_float),
                ("bias2", ctypes.c_float), ("uv_grad_accum", ctypes.c_void_p), ("grad_accum_dur", ctypes.c_void_p),
                ("mode", ctypes.c_int)]


class AdamGroup(ctypes.Structure):  # gsplat_adam_group
    _fields_ = [("param", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p), ("exp_avg_sq", ctypes.c_void_p),
                ("grad", ctypes.c_void_p), ("stride", ctypes.c_int), ("packed_column", ctypes.c_int),
                ("lr", ctypes.c_float)]


MAX_ADAM_GROUPS = 8
ABI_VERSION = 8  # GSPLAT_ABI_VERSION of include/gsplat_hip.h; bumped with every signature change

# every symbol include/gsplat_hip.h declares, with its argument types
_P, _I, _F, _S = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
SIGNATURES = {
    "gsplat_last_error": (ctypes.c_char_p, []),
    "gsplat_abi_version": (_I, []),
    "gsplat_source_hash": (ctypes.c_char_p, []),
    "gsplat_build_flags": (ctypes.c_char_p, []),
    "gsplat_release_scratch": (_I, []),
    "gsplat_pool_alloc": (_I, [ctypes.POINTER(ctypes.c_void_p), _S]),
    "gsplat_pool_free": (_I, [_P]),
    "gsplat_pool_alloc_on": (_I, [ctypes.POINTER(ctypes.c_void_p), _S, _P]),
    "gsplat_pool_free_on": (_I, [_P, _P]),
    "gsplat_pool_cross_stream_reuses": (ctypes.c_ulonglong, []),
    "gsplat_pool_trim": (_I, [_S]),
    "gsplat_pool_release": (_I, []),
    "gsplat_pool_bytes": (_S, [_I]),
    "gsplat_compute_camera_space_points": (_I, [_P, _P, _I, _P, _P]),
    "gsplat_project_to_screen": (_I, [_P, _P, _I, _I, _I, _P, _P]),
    "gsplat_cull_gaussians": (_I, [_P, _P, _I, _F, _I, _I, _I, _P, _P]),
    "gsplat_compute_sigma": (_I, [_P, _P, _I, _P, _P]),
    "gsplat_compute_conic": (_I, [_P, _P, _P, _F, _F, _F, _F, _F, _I, _P, _P, _P, _P]),
    "gsplat_get_sorted_gaussian_list": (_I, [_P, _P, _P, _I, _I, _I, ctypes.POINTER(_S), _P, _P, _P]),
    "gsplat_precompute_spherical_harmonics": (_I, [_P, _P, _P, _F, _F, _F, _I, _I, _P, _P]),
    "gsplat_render_image": (_I, [_P, _P, _P, _P, _F, _P, _P, _I, _I, _P, _P, _P, _P]),
    "gsplat_project_to_screen_backward": (_I, [_P, _P, _P, _I, _I, _I, _P, _P]),
    "gsplat_compute_camera_space_points_backward": (_I, [_P, _P, _P, _I, _P, _P]),
    "gsplat_compute_projection_jacobian_backward": (_I, [_P, _F, _F, _F, _F, _P, _I, _P, _P]),
    "gsplat_compute_conic_backward": (_I, [_P, _P, _P, _P, _P, _I, _P, _P, _P]),
    "gsplat_compute_sigma_backward": (_I, [_P, _P, _P, _I, _P, _P, _P]),
    "gsplat_precompute_spherical_harmonics_backward": (_I, [_P, _P, _P, _F, _F, _F, _P, _I, _I, _P, _P, _P, _P]),
    "gsplat_render_image_backward": (_I, [_P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P]),
    "gsplat_context_set_binning_route": (_I, [_P, _I]),
    "gsplat_context_set_segment_options": (_I, [_P, _I, _I, ctypes.c_float]),
    "gsplat_backward_render": (_I, [_P, _P, _F, _P, _P]),
    "gsplat_context_detach_forward_outputs": (_I, [_P]),
    "gsplat_context_last_compaction": (_I, [_P, ctypes.POINTER(_P), ctypes.POINTER(_P), ctypes.POINTER(_P), ctypes.POINTER(_I), ctypes.POINTER(_I)]),
    "gsplat_compact_rows_ranked": (_I, [_P, _P, _P, _I, _I, _P, _I, _P]),
    "gsplat_fill_f32": (_I, [_P, _S, _F, _P]),
    "gsplat_mask_selected_rows": (_I, [_P, _I, _P, _I, _P]),
    "gsplat_backward_render_split": (_I, [_P, _P, _F, _P, _P, _P, _P]),
    "gsplat_backward_gaussians_split": (_I, [_P, _P, _P, _I, _P, _P, _I, _I, _P]),
    "gsplat_optimizer_step_sh_views": (_I, [_I, _I, _I, _P, _P, _S, _P, _P, _P, _P, _F, _P, _P, _P, _F, _F, _F, _F, _F, _F, _P]),
    "gsplat_compact_masked_array_bounded": (_I, [_P, _P, _I, _I, _P, _I, ctypes.POINTER(_I), _P]),
    "gsplat_backward_gaussians": (_I, [_P, _P, _P, _I, _P, _P]),
    "gsplat_backward_gaussians_range": (_I, [_P, _P, _P, _I, _P, _I, _I, _P]),
    "gsplat_pack_gradients_split_range": (_I, [_P, _P, _I, _I, _I, _P, _P, _P]),
    "gsplat_pack_gradients_split": (_I, [_P, _P, _I, _P, _P, _P]),
    "gsplat_unpack_gradients_split": (_I, [_P, _P, _P, _S, _I, _I, _I, _P, _P]),
    "gsplat_fused_loss": (_I, [_P, _P, _I, _I, _F, _P, ctypes.POINTER(ctypes.c_float), _P]),
    "gsplat_compute_psnr": (_I, [_P, _P, _I, _I, ctypes.POINTER(ctypes.c_float), _P]),
    "gsplat_adam_step": (_I, [_P, _P, _P, _P, _F, _F, _F, _F, _F, _F, _I, _I, _P]),
    "gsplat_optimizer_step": (_I, [_P, _I, _P, _I, _F, _F, _F, _F, _F, _P, _P, _P, _P]),
    "gsplat_optimizer_step_packed": (_I, [_P, _I, _I, _P, _I, _F, _F, _F, _F, _F, _P, _P, _P, _P]),
    "gsplat_optimizer_step_sh_factored": (_I, [_P, _I, _I, _P, _P, _P, _F, _F, _F, _F, _F, _F, _P, _F, _F, _F, _P, _P]),
    "gsplat_pack_uv_grad_norm": (_I, [_P, ctypes.POINTER(Gradients), _I, _P, _P]),
    "gsplat_initialize_gaussians": (_I, [_P, _P, _I, _P, _P, _P, _P, _P, _P]),
    "gsplat_knn_mean_distance": (_I, [_P, _I, _I, _P, _P]),
    "gsplat_compute_morton_codes": (_I, [_I, _P, _F, _F, _F, _F, _F, _F, _P, _P]),
    "gsplat_clone_gaussians": (_I, [_I, _I] + [_P] * 14 + [_P]),
    "gsplat_split_gaussians": (_I, [_I, _F, _I] + [_P] * 14 + [ctypes.c_ulonglong, _P]),
    "gsplat_density_masks": (_I, [_I, _P, _P, _P, _P, _F, _F, _F, _F, _P, _P, _P, _P, _P, _P]),
    "gsplat_expand_sh": (_I, [_I, _I, _P, _P, _P]),
    "gsplat_gather_rows": (_I, [_I, _I, _P, _P, _P, _P]),
    "gsplat_compact_masked_array": (_I, [_P, _P, _I, _I, _P, ctypes.POINTER(_I), _P]),
    "gsplat_scatter_masked_array": (_I, [_P, _P, _I, _I, _P, _P]),
    "gsplat_context_create": (_I, [ctypes.POINTER(_P), _I, _I, _I]),
    "gsplat_context_destroy": (_I, [_P]),
    "gsplat_context_bytes": (_S, [_P]),
    "gsplat_rasterize_image": (_I, [_P, ctypes.POINTER(Gaussians), ctypes.POINTER(Camera),
                                    ctypes.POINTER(RasterConfig), _F, _I, ctypes.POINTER(ForwardView), _P]),
    "gsplat_backward_pass": (_I, [_P, ctypes.POINTER(Gaussians), ctypes.POINTER(Camera), _P, _F, _I,
                                  ctypes.POINTER(Gradients), _P]),
    "gsplat_backward_gaussians_adam": (_I, [_P, ctypes.POINTER(Gaussians), ctypes.POINTER(Camera), _I,
                                            ctypes.POINTER(AdamFused), ctypes.POINTER(Gradients), _P]),
    "gsplat_context_set_render_only": (_I, [_P, _I]),
    "gsplat_context_set_lean_forward": (_I, [_P, _I]),
    "gsplat_context_set_preprocess_split": (_I, [_P, _I]),
    "gsplat_context_get_counters": (_I, [_P, ctypes.POINTER(ctypes.c_longlong), _I]),
    "gsplat_context_set_timing": (_I, [_P, _I]),
    "gsplat_context_set_timing_stages": (_I, [_P, ctypes.c_uint]),
    "gsplat_context_get_timing": (_I, [_P, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_longlong), _I]),
    "gsplat_pack_gradients_global": (_I, [_P, ctypes.POINTER(Gradients), _I, _I, _P, _P]),
    "gsplat_packed_gradient_width": (_I, [_I]),
    "gsplat_factored_gradient_width": (_I, [_I]),
    "gsplat_pack_gradients_factored": (_I, [_P, ctypes.POINTER(Gradients), _I, _I, _I, _P, _P]),
    "gsplat_unpack_gradients_factored": (_I, [_P, _P, _P, _I, _I, _I, _P, _P]),
}


def load():
    """Load the library (building it first if sources are newer).  Raises if that fails."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("GSPLAT_LIB")  # tooling (tools/ab, diagnostic builds): load exactly this prebuilt variant
    if not path:
        # GSPLAT_NO_BUILD=1: load the library as it is (rank processes whose launcher has already built it; runs under
        # rocprofv3, where this process has an initialised GPU and must not spawn compilers).  Otherwise build(): cheap
        # when the library is fresh (mtime scan).
        if os.environ.get("GSPLAT_NO_BUILD") == "1":
            if not os.path.exists(LIB_PATH):
                raise OSError(f"{LIB_PATH} is missing and GSPLAT_NO_BUILD=1 forbids building it")
        else:
            try:
                build()
            except (OSError, subprocess.CalledProcessError) as e:
                if not os.path.exists(LIB_PATH):
                    raise
                import sys
                print(f"[3dgs_amd] WARNING: rebuilding libgsplat_hip.so failed ({e}); loading the EXISTING library, "
                      "which may be older than the sources (the ABI version is checked below)", file=sys.stderr, flush=True)
        path = LIB_PATH
    # torch owns device memory and ships its own HIP runtime: import it first so that this library binds to the
    # same runtime instance (two runtimes in one process do not see each other's allocations)
    import torch  # noqa: F401
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header and library out of sync
        fn.restype = res
        fn.argtypes = args
    got = lib.gsplat_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"libgsplat_hip.so reports ABI {got}, this binding expects {ABI_VERSION}: stale library")
    # The binary says what it was built from (gsplat_source_hash).  A loader that did not build it -- GSPLAT_NO_BUILD
    # (rank processes, runs under rocprofv3) or a GSPLAT_LIB variant -- may be looking at a library older than, or
    # deliberately different from, the sources next to it: say so once, and let library_matches_sources() keep stored
    # profiles (profiles/traffic.json) from being attached to it.
    built_from, flags = lib.gsplat_source_hash().decode(), lib.gsplat_build_flags().decode()
    if built_from != source_hash() or flags:
        import sys
        print(f"[3dgs_amd] NOTE: {path} was built from sources {built_from}"
              + (f" with extra flags '{flags}'" if flags else "") + f"; the sources here hash to {source_hash()}",
              file=sys.stderr, flush=True)
    _lib = lib
    return lib


def library_source_hash():
    """The source hash the LOADED binary carries (gsplat_source_hash), with its extra build flags appended when it is a
    diagnostic build: equals source_hash() exactly when the library was built from the sources next to it."""
    lib = load()
    flags = lib.gsplat_build_flags().decode()
    return lib.gsplat_source_hash().decode() + (("+" + flags) if flags else "")


def library_matches_sources():
    return library_source_hash() == source_hash()


def check(status):
    if status != 0:
        raise GsplatError(status, load().gsplat_last_error().decode("utf-8", "replace"))
    return status
