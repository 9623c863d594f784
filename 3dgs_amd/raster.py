"""Per-view forward/backward on the device-resident context (binding of gsplat_rasterize_image /
gsplat_backward_pass, include/gsplat_hip.h).

Mirrors the two call sites of the reference trainer: ``rasterize_image(...)``
(/root/reference/cuda/raster.cu:12-136, called at cuda/trainer.cu:1350) and the operator
chain of ``TrainerImpl::backward_pass`` (cuda/trainer.cu:941-1012).  All tensors are torch
device tensors; results are zero-copy views into the context's workspace (valid until the
next forward on the same context).
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import check


class _DevView:
    """Expose a raw device pointer to torch through __cuda_array_interface__ (zero copy)."""

    def __init__(self, ptr, shape, typestr, owner):
        self.__cuda_array_interface__ = {"shape": tuple(int(s) for s in shape), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2}
        self._owner = owner


def _view(ptr, shape, typestr, owner):
    if ptr is None and int(np.prod(shape)) != 0:
        return None  # an output the library did not produce (render-only context)
    if ptr is None or int(np.prod(shape)) == 0:
        dt = {"<f4": torch.float32, "<i4": torch.int32, "|u1": torch.uint8}[typestr]
        return torch.empty(tuple(shape), dtype=dt, device="cuda")
    return torch.as_tensor(_DevView(ptr, shape, typestr, owner), device="cuda")


_EAGER_VIEWS = bool(int(__import__("os").environ.get("GSPLAT_EAGER_VIEWS", "0")))  # debugging: build every view at once


class _LazyViews(dict):
    """The forward's result: counts are plain entries, every array is turned into a tensor view the first time it is
    asked for.  A view costs ~15 us of host time (torch queries the pointer), sixteen of them per call were most of the
    host's work between the forward's record and the launch of whatever consumes the image -- and a training loop
    reads two of them."""

    def __init__(self, plain, specs, owner):
        super().__init__(plain)
        self._specs, self._owner = specs, owner

    def __missing__(self, key):
        ptr, shape, typestr = self._specs.pop(key)  # KeyError for unknown names, as a dict
        value = _view(ptr, shape, typestr, self._owner)
        self[key] = value
        return value

    def _all(self):
        for key in list(self._specs):
            self[key]
        return self

    def get(self, key, default=None):
        return self[key] if key in self else default

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._specs

    def __iter__(self):
        return dict.__iter__(self._all())

    def __len__(self):
        return dict.__len__(self._all())

    def keys(self):
        return dict.keys(self._all())

    def items(self):
        return dict.items(self._all())

    def values(self):
        return dict.values(self._all())


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None and t.numel() else None


def device_params(params, device="cuda"):
    """numpy parameter dict (scene.make_gaussians) -> dict of contiguous float32 device tensors."""
    return {k: torch.as_tensor(np.ascontiguousarray(v, dtype=np.float32)).to(device).contiguous()
            for k, v in params.items()}


def device_camera(cam, device="cuda"):
    out = dict(cam)
    out["view"] = torch.as_tensor(np.asarray(cam["view"], np.float32)).to(device)
    out["proj"] = torch.as_tensor(np.asarray(cam["proj"], np.float32)).to(device)
    out["campos_dev"] = torch.as_tensor(np.asarray(cam["campos"], np.float32)).to(device)  # for the exchange step
    return out


class RasterContext:
    """Owns a gsplat_context (persistent HBM workspace sized for max_gaussians / max image)."""

    def __init__(self, max_gaussians, max_width, max_height):
        self._lib = _lib.load()
        h = ctypes.c_void_p()
        check(self._lib.gsplat_context_create(ctypes.byref(h), int(max_gaussians), int(max_width), int(max_height)))
        self._h = h
        self._last = None
        self.max_gaussians, self.max_width, self.max_height = int(max_gaussians), int(max_width), int(max_height)

    def close(self):
        if self._h:
            self._lib.gsplat_context_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def workspace_bytes(self):
        return int(self._lib.gsplat_context_bytes(self._h))

    STAGES = ("project_cull", "preprocess", "bin_sort", "reserved", "render_forward", "zero_grad_rows",
              "render_backward", "preprocess_backward")

    def set_binning_route(self, route):
        """0 automatic, 1 LDS counting sort + per-tile depth sort, 2 stable radix sorts (identical results)."""
        check(self._lib.gsplat_context_set_binning_route(self._h, int(route)))

    def set_segment_options(self, poll_budget=-1, thin_layer_blocks=-1, gate=-1.0):
        """Testing / tuning hook of the forward's long-list segments (gsplat_context_set_segment_options); results do not
        depend on it."""
        check(self._lib.gsplat_context_set_segment_options(self._h, int(poll_budget), int(thin_layer_blocks), float(gate)))

    def set_render_only(self, enabled):
        """Serving mode: forwards skip the outputs only a backward reads (see gsplat_context_set_render_only)."""
        check(self._lib.gsplat_context_set_render_only(self._h, int(bool(enabled))))

    def set_lean_forward(self, enabled):
        """Training through backward_pass: the forward stops storing Sigma, J, conic and the SH colour (the fused backward
        recomputes what it needs); those four views are then absent from the forward's result."""
        check(self._lib.gsplat_context_set_lean_forward(self._h, int(bool(enabled))))

    def set_preprocess_split(self, mode):
        """How the per-gaussian forward is launched: 0 (default) the single fused kernel, 1 SH colour then geometry on the
        caller's stream, 2 the two side by side on two streams (gsplat_context_set_preprocess_split); every output is
        bit-identical in all modes, 1 and 2 measure slower (profiles/r06_ab_preprocess_split.txt)."""
        check(self._lib.gsplat_context_set_preprocess_split(self._h, int(mode)))

    def counters(self):
        """What the context's forwards did so far (gsplat_context_get_counters)."""
        v = (ctypes.c_longlong * 10)()
        self._lib.gsplat_context_get_counters(self._h, v, 10)
        return dict(forwards=int(v[0]), tail_redone=int(v[1]), compact_walks=int(v[2]), instance_growths=int(v[3]),
                    ordered_backwards=int(v[4]), segmented_backwards=int(v[5]), segmented_forwards=int(v[6]),
                    longest_chain=int(v[7]), chain_sum=int(v[8]), segment_fallbacks=int(v[9]))

    def set_timing(self, enabled, stages=None):
        """Per-stage HIP-event timing on / off; `stages`: names from STAGES to time only those (each timed stage costs
        two event records per call)."""
        if enabled and stages is not None:
            mask = 0
            for name in stages:
                mask |= 1 << self.STAGES.index(name)
            check(self._lib.gsplat_context_set_timing_stages(self._h, mask))
        else:
            check(self._lib.gsplat_context_set_timing(self._h, int(bool(enabled))))

    def get_timing(self):
        """{stage: (mean ms per launch, launches)} since timing was enabled (HIP events on the launch stream)."""
        ms = (ctypes.c_double * 8)()
        cnt = (ctypes.c_longlong * 8)()
        self._lib.gsplat_context_get_timing(self._h, ms, cnt, 8)
        return {s: ((ms[i] / cnt[i]) if cnt[i] else 0.0, int(cnt[i])) for i, s in enumerate(self.STAGES)}

    @staticmethod
    def _structs(params, cam, l_max):
        g = _lib.Gaussians()
        g.num_gaussians = int(params["xyz"].shape[0])
        g.xyz, g.rgb, g.opacity = _ptr(params["xyz"]), _ptr(params["rgb"]), _ptr(params["opacity"])
        g.scale, g.quaternion = _ptr(params["scale"]), _ptr(params["quaternion"])
        g.sh = _ptr(params["sh"]) if l_max > 0 else None
        c = _lib.Camera()
        c.width, c.height = int(cam["width"]), int(cam["height"])
        c.focal_x, c.focal_y = float(cam["fx"]), float(cam["fy"])
        for k in range(3):
            c.campos[k] = float(cam["campos"][k])
        c.view, c.proj = _ptr(cam["view"]), _ptr(cam["proj"])
        return g, c

    def rasterize_image(self, params, cam, config, bg_color, l_max):
        """params: dict of device tensors (xyz rgb sh opacity scale quaternion); sh must be stored with the
        current band's stride, [N,(l_max+1)^2-1,3] (cuda_data.cuh layout).  Returns a dict of views."""
        g, c = self._structs(params, cam, l_max)
        if l_max > 0:
            want = ((l_max + 1) ** 2 - 1) * 3
            if params["sh"].numel() != g.num_gaussians * want:
                raise ValueError("sh must be packed at the current band's stride")
        cfg = _lib.RasterConfig(float(config["near_thresh"]), float(config["mh_dist"]), int(config["cull_mask_padding"]))
        fv = _lib.ForwardView()
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(self._lib.gsplat_rasterize_image(self._h, ctypes.byref(g), ctypes.byref(c), ctypes.byref(cfg),
                                               float(bg_color), int(l_max), ctypes.byref(fv), st))
        N, M, S = g.num_gaussians, int(fv.num_culled), int(fv.num_splats)
        W, H = c.width, c.height
        T = ((W + 15) // 16) * ((H + 15) // 16)
        out = _LazyViews(
            dict(num_culled=M, num_pairs=int(fv.num_pairs), num_splats=S),
            dict(mask=(fv.mask, (N,), "|u1"), uv_all=(fv.uv, (N, 2), "<f4"), xyz_c_all=(fv.xyz_c, (N, 3), "<f4"),
                 compact_to_global=(fv.compact_to_global, (M,), "<i4"), sigma=(fv.sigma, (M, 6), "<f4"),
                 conic=(fv.conic, (M, 3), "<f4"), J=(fv.J, (M, 6), "<f4"), rgb=(fv.precomputed_rgb, (M, 3), "<f4"),
                 radius=(fv.radius, (M, 4), "<f4"), uv=(fv.uv_selected, (M, 2), "<f4"),
                 xyz_c=(fv.xyz_c_selected, (M, 3), "<f4"), sorted=(fv.sorted_gaussians, (S,), "<i4"),
                 ranges=(fv.splat_start_end_idx_by_tile_idx, (T + 1,), "<i4"), image=(fv.image, (H, W, 3), "<f4"),
                 T=(fv.weight_per_pixel, (H, W), "<f4"), n=(fv.splats_per_pixel, (H, W), "<i4")), self)
        self._last = (g.num_gaussians, M, l_max)
        return out._all() if _EAGER_VIEWS else out

    def alloc_gradients(self, M, l_max, intermediates=False, device="cuda", factored_sh=False):
        """Leaf gradients in compacted order; intermediates=True adds the reference's six intermediate gradient arrays
        (cuda_data.cuh:28-36), an iterable of their names only those (the kernel skips the ones that are absent).
        factored_sh: no `sh` array -- the backward skips the SH gradients and AdamOptimizer.step rebuilds them from
        `precompute_rgb` (added) and the viewing direction (gsplat_optimizer_step_sh_factored)."""
        n_rest = (l_max + 1) ** 2 - 1
        z = lambda *s: torch.empty(*s, dtype=torch.float32, device=device)
        g = dict(xyz=z(M, 3), rgb=z(M, 3), sh=z(M, n_rest, 3), opacity=z(M), scale=z(M, 3), quaternion=z(M, 4))
        shapes = dict(conic=(M, 3), uv=(M, 2), J=(M, 6), sigma=(M, 6), xyz_c=(M, 3), precompute_rgb=(M, 3))
        wanted = shapes if intermediates is True else tuple(intermediates or ())  # True: all six; or an iterable of names
        if factored_sh and l_max > 0:
            g["sh"] = None
            if "precompute_rgb" not in wanted:
                wanted = tuple(wanted) + ("precompute_rgb",)
        g.update({k: z(*shapes[k]) for k in wanted})
        return g

    @staticmethod
    def _grad_struct(grads):
        gs = _lib.Gradients()
        gs.grad_xyz, gs.grad_rgb, gs.grad_sh = _ptr(grads["xyz"]), _ptr(grads.get("rgb")), _ptr(grads.get("sh"))
        gs.grad_opacity, gs.grad_scale = _ptr(grads.get("opacity")), _ptr(grads.get("scale"))
        gs.grad_quaternion = _ptr(grads.get("quaternion"))
        for k in ("conic", "uv", "J", "sigma", "xyz_c", "precompute_rgb"):
            setattr(gs, "grad_" + k, _ptr(grads.get(k)))
        return gs

    def backward_pass(self, params, cam, grad_image, bg_color, l_max, grads):
        """grads: dict from alloc_gradients (compacted order); every leaf gradient is overwritten."""
        g, c = self._structs(params, cam, l_max)
        gs = self._grad_struct(grads)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(self._lib.gsplat_backward_pass(self._h, ctypes.byref(g), ctypes.byref(c), _ptr(grad_image),
                                             float(bg_color), int(l_max), ctypes.byref(gs), st))
        return grads

    def backward_pass_adam(self, params, cam, grad_image, bg_color, l_max, adam, grads=None):
        """backward_pass with the optimizer step inside (single-GPU training): the compositing backward, then ONE
        per-gaussian kernel that differentiates every visible gaussian and applies the masked Adam step to its rows of the
        parameters (`params`, in place), the moments and the densification statistics (gsplat_backward_gaussians_adam).
        adam: an _lib.AdamFused (AdamOptimizer.fused_state); grads: optional gradient arrays to fill as well."""
        self.backward_render(grad_image, bg_color)
        self.backward_gaussians_adam(params, cam, l_max, adam, grads)

    def backward_gaussians_adam(self, params, cam, l_max, adam, grads=None):
        """Second half of backward_pass_adam: the per-gaussian chain with the optimizer step inside."""
        g, c = self._structs(params, cam, l_max)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        gs = ctypes.byref(self._grad_struct(grads)) if grads is not None else None
        check(self._lib.gsplat_backward_gaussians_adam(self._h, ctypes.byref(g), ctypes.byref(c), int(l_max),
                                                       ctypes.byref(adam), gs, st))

    def backward_render(self, grad_image, bg_color, rgb_global=None, common=None, uv_norm=None):
        """First half of backward_pass: compositing backward; optionally this view's g_rgb in global order [N,3].
        common [N,12] (and uv_norm [N]): the split exchange's rows -- the rows of the gaussians this view culled are
        cleared here, the others are written by backward_gaussians_split (gsplat_backward_render_split)."""
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(self._lib.gsplat_backward_render_split(self._h, _ptr(grad_image), float(bg_color), _ptr(rgb_global),
                                                     _ptr(common), _ptr(uv_norm), st))

    def backward_gaussians_split(self, params, cam, l_max, common, uv_norm=None, first=0, end=None):
        """Second half of backward_pass for a view-sharded step: the twelve direction-independent gradient columns go
        straight into common[N,12] at the gaussians' global indices (|grad_uv| into uv_norm[N]); no compacted gradient
        array is written, the colour gradients are rebuilt by the optimizer from the gathered g_rgb."""
        g, c = self._structs(params, cam, l_max)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(self._lib.gsplat_backward_gaussians_split(self._h, ctypes.byref(g), ctypes.byref(c), int(l_max), _ptr(common),
                                                        _ptr(uv_norm), int(first), int(g.num_gaussians if end is None else end), st))

    def backward_gaussians(self, params, cam, l_max, grads):
        """Second half of backward_pass: the per-gaussian operator chain."""
        g, c = self._structs(params, cam, l_max)
        gs = self._grad_struct(grads)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(self._lib.gsplat_backward_gaussians(self._h, ctypes.byref(g), ctypes.byref(c), int(l_max), ctypes.byref(gs), st))
        return grads

    def backward_gaussians_range(self, params, cam, l_max, grads, first, end):
        """The per-gaussian chain for the gaussians with global index in [first, end) (chunked exchange)."""
        g, c = self._structs(params, cam, l_max)
        gs = self._grad_struct(grads)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(self._lib.gsplat_backward_gaussians_range(self._h, ctypes.byref(g), ctypes.byref(c), int(l_max),
                                                        ctypes.byref(gs), int(first), int(end), st))
        return grads

    def pack_gradients_global(self, grads, l_max, num_gaussians, packed):
        gs = self._grad_struct(grads)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(self._lib.gsplat_pack_gradients_global(self._h, ctypes.byref(gs), int(l_max), int(num_gaussians),
                                                     _ptr(packed), st))
        return packed


def pack_gradients_factored(ctx, grads, num_gaussians, rank, world, factored):
    gs = RasterContext._grad_struct(grads)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    check(_lib.load().gsplat_pack_gradients_factored(ctx._h, ctypes.byref(gs), int(num_gaussians), int(rank), int(world),
                                                     _ptr(factored), st))
    return factored


def unpack_gradients_factored(xyz, campos_all, factored, l_max, num_gaussians, world, packed):
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    check(_lib.load().gsplat_unpack_gradients_factored(_ptr(xyz), _ptr(campos_all), _ptr(factored), int(l_max),
                                                       int(num_gaussians), int(world), _ptr(packed), st))
    return packed


def pack_gradients_split(ctx, grads, num_gaussians, common, rgb):
    gs = RasterContext._grad_struct(grads)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    check(_lib.load().gsplat_pack_gradients_split(ctx._h, ctypes.byref(gs), int(num_gaussians), _ptr(common), _ptr(rgb), st))


def pack_gradients_split_range(ctx, grads, num_gaussians, first, end, common, rgb=None):
    gs = RasterContext._grad_struct(grads)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    check(_lib.load().gsplat_pack_gradients_split_range(ctx._h, ctypes.byref(gs), int(num_gaussians), int(first), int(end),
                                                        _ptr(common), _ptr(rgb), st))


def pack_uv_grad_norm(ctx, grads, num_gaussians, uv_norm):
    """This view's |grad_uv| in global gaussian order (0 where culled): gsplat_pack_uv_grad_norm."""
    gs = RasterContext._grad_struct(grads)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    check(_lib.load().gsplat_pack_uv_grad_norm(ctx._h, ctypes.byref(gs), int(num_gaussians), _ptr(uv_norm), st))
    return uv_norm


def unpack_gradients_split(xyz, common, rgb_all, rank_stride, l_max, num_gaussians, world, packed):
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    check(_lib.load().gsplat_unpack_gradients_split(_ptr(xyz), _ptr(common), _ptr(rgb_all), int(rank_stride), int(l_max),
                                                    int(num_gaussians), int(world), _ptr(packed), st))
    return packed


def factored_gradient_width(world):
    return int(_lib.load().gsplat_factored_gradient_width(int(world)))


def packed_gradient_width(l_max):
    return int(_lib.load().gsplat_packed_gradient_width(int(l_max)))
