"""Host-side mirror of the reference's optimizer step (TrainerImpl::optimizer_step, cuda/trainer.cu:1027-1158;
constants include/gsplat_cuda/optimizer.cuh:9-11) on top of the C ABI's masked in-place Adam.

The reference compacts every parameter group and both of its moments by the view's mask, calls adam_step on the
copies and scatters them back.  Here the moments live in global order next to the parameters and one kernel updates
the visible rows in place; `step` takes the compacted gradients of one view, `step_packed` the all-reduced packed rows
of a view-sharded step (3dgs_amd/dist.py).  There is no CPU fallback: without the HIP library these calls raise.
"""
import ctypes

import torch

from . import _lib
from .dist import packed_layout
from .ops import check

B1, B2, EPS = 0.9, 0.999, 1e-8  # include/gsplat_cuda/optimizer.cuh:9-11

GROUPS = ("xyz", "rgb", "sh", "opacity", "scale", "quaternion")

# config/base.yaml of the reference (learning-rate multipliers of ConfigParameters)
DEFAULT_LR = dict(base_lr=1e-3, xyz_lr_multiplier_init=1.6e-1, xyz_lr_multiplier_final=1.6e-3, quat_lr_multiplier=1.0,
                  scale_lr_multiplier=5.0, opacity_lr_multiplier=25.0, rgb_lr_multiplier=2.5, sh_lr_multiplier=0.125,
                  num_iters=7000)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class AdamOptimizer:
    """Adam state (exp_avg / exp_avg_sq per group, global order) + densification accumulators.

    params: dict name -> device tensor [N, stride] (xyz rgb sh opacity scale quaternion); updated in place."""

    def __init__(self, params, l_max, lr_config=None, scene_extent=1.0):
        self.params, self.l_max = params, l_max
        self.cfg = dict(DEFAULT_LR, **(lr_config or {}))
        self.scene_extent = float(scene_extent)
        self.names = [g for g in GROUPS if not (g == "sh" and l_max == 0)]
        self.exp_avg = {g: torch.zeros_like(params[g]) for g in self.names}
        self.exp_avg_sq = {g: torch.zeros_like(params[g]) for g in self.names}
        n = params["xyz"].shape[0]
        self.uv_grad_accum = torch.zeros(n, device=params["xyz"].device)       # OptimizerAccumulators
        self.grad_accum_dur = torch.zeros(n, dtype=torch.int32, device=params["xyz"].device)
        self.cols, self.width = packed_layout(l_max)

    def learning_rates(self, it):
        """Per-group rates at iteration `it` (cuda/trainer.cu:1049-1071)."""
        c = self.cfg
        decay = (c["xyz_lr_multiplier_final"] / c["xyz_lr_multiplier_init"]) ** (float(it) / float(c["num_iters"]))
        return dict(xyz=self.scene_extent * c["base_lr"] * c["xyz_lr_multiplier_init"] * decay,
                    rgb=c["base_lr"] * c["rgb_lr_multiplier"], sh=c["base_lr"] * c["sh_lr_multiplier"],
                    opacity=c["base_lr"] * c["opacity_lr_multiplier"], scale=c["base_lr"] * c["scale_lr_multiplier"],
                    quaternion=c["base_lr"] * c["quat_lr_multiplier"])

    @staticmethod
    def bias_corrections(it):
        return 1.0 - B1 ** (it + 1), 1.0 - B2 ** (it + 1)  # cuda/trainer.cu:1046-1047

    def _groups(self, it, grads=None, names=None):
        lrs = self.learning_rates(it)
        names = self.names if names is None else names
        arr = (_lib.AdamGroup * len(names))()
        for k, g in enumerate(names):
            p = self.params[g]
            stride = self.cols[g][1] - self.cols[g][0]
            arr[k].param, arr[k].exp_avg, arr[k].exp_avg_sq = p.data_ptr(), self.exp_avg[g].data_ptr(), self.exp_avg_sq[g].data_ptr()
            arr[k].grad = grads[g].data_ptr() if grads is not None else None
            arr[k].stride, arr[k].packed_column, arr[k].lr = stride, self.cols[g][0], lrs[g]
        return arr

    def step(self, it, fwd, grads, campos=None):
        """One view: `fwd` is the dict RasterContext.rasterize_image returned, `grads` the dict filled by
        backward_pass (compacted order).  When grads carries the "uv" intermediate the densification statistics
        (uv_grad_accum, grad_accum_dur) are updated as the reference does.

        grads["sh"] is None (RasterContext.alloc_gradients(factored_sh=True)): the backward did not store the SH
        gradients; they are rebuilt from grads["precompute_rgb"] and the viewing direction -- `campos`, the camera
        position of the view, is needed then -- by gsplat_optimizer_step_sh_factored, BEFORE the xyz group moves."""
        b1c, b2c = self.bias_corrections(it)
        names = list(self.names)
        c2g = fwd["compact_to_global"].data_ptr() if fwd["num_culled"] else None
        if "sh" in names and grads.get("sh") is None:
            if campos is None or grads.get("precompute_rgb") is None:
                raise ValueError("factored SH gradients need campos and grads['precompute_rgb']")
            names.remove("sh")
            check(_lib.load().gsplat_optimizer_step_sh_factored(
                c2g, int(fwd["num_culled"]), int(self.l_max), self.params["sh"].data_ptr(), self.exp_avg["sh"].data_ptr(),
                self.exp_avg_sq["sh"].data_ptr(), self.learning_rates(it)["sh"], B1, B2, EPS, b1c, b2c,
                self.params["xyz"].data_ptr(), float(campos[0]), float(campos[1]), float(campos[2]),
                grads["precompute_rgb"].data_ptr(), _stream()))
        arr = self._groups(it, grads, names)
        uv = grads.get("uv")
        check(_lib.load().gsplat_optimizer_step(
            c2g, int(fwd["num_culled"]), arr, len(names), B1, B2, EPS, b1c, b2c, uv.data_ptr() if uv is not None else None,
            self.uv_grad_accum.data_ptr() if uv is not None else None,
            self.grad_accum_dur.data_ptr() if uv is not None else None, _stream()))

    def step_after_partial_backward(self, it, fwd, grads, campos):
        """What is left of `step` behind RasterContext.backward_pass_adam(..., fused_state(it, mode=1)): the SH group from
        the factored gradient (it takes its directions from the positions, so it goes first), then the position group."""
        b1c, b2c = self.bias_corrections(it)
        c2g = fwd["compact_to_global"].data_ptr() if fwd["num_culled"] else None
        if "sh" in self.names:
            check(_lib.load().gsplat_optimizer_step_sh_factored(
                c2g, int(fwd["num_culled"]), int(self.l_max), self.params["sh"].data_ptr(), self.exp_avg["sh"].data_ptr(),
                self.exp_avg_sq["sh"].data_ptr(), self.learning_rates(it)["sh"], B1, B2, EPS, b1c, b2c,
                self.params["xyz"].data_ptr(), float(campos[0]), float(campos[1]), float(campos[2]),
                grads["precompute_rgb"].data_ptr(), _stream()))
        arr = self._groups(it, grads, ["xyz"])
        check(_lib.load().gsplat_optimizer_step(c2g, int(fwd["num_culled"]), arr, 1, B1, B2, EPS, b1c, b2c, None, None, None,
                                                _stream()))

    def fused_state(self, it, with_stats=True, mode=0):
        """The optimizer's state at iteration `it` as the struct RasterContext.backward_pass_adam hands to the per-gaussian
        backward, which then applies this very step itself (single-GPU training: same parameters, moments and statistics
        as backward_pass + step, bit for bit, without the gradients' round trip through memory)."""
        b1c, b2c = self.bias_corrections(it)
        lrs = self.learning_rates(it)
        a = _lib.AdamFused()
        for k, g in enumerate(GROUPS):
            if g in self.exp_avg:
                a.exp_avg[k], a.exp_avg_sq[k] = self.exp_avg[g].data_ptr(), self.exp_avg_sq[g].data_ptr()
            a.lr[k] = lrs[g]
        a.b1, a.b2, a.eps, a.bias1, a.bias2 = B1, B2, EPS, b1c, b2c
        a.uv_grad_accum = self.uv_grad_accum.data_ptr() if with_stats else None
        a.grad_accum_dur = self.grad_accum_dur.data_ptr() if with_stats else None
        a.mode = int(mode)
        return a

    def step_packed(self, it, packed, uv_norm_sum=None):
        """All-reduced packed rows [N, width] (sum over the views of the step).  uv_norm_sum [N]: the all-reduced
        per-view |grad_uv| (ViewShardedStep(with_uv_norm=True)); when given, the densification statistics are updated
        once per view that saw the gaussian, as W single-view steps of the reference would."""
        assert packed.shape[1] == self.width and packed.is_contiguous()
        b1c, b2c = self.bias_corrections(it)
        arr = self._groups(it)
        stats = uv_norm_sum is not None
        check(_lib.load().gsplat_optimizer_step_packed(
            packed.data_ptr(), packed.shape[0], self.width, arr, len(self.names), B1, B2, EPS, b1c, b2c,
            uv_norm_sum.data_ptr() if stats else None, self.uv_grad_accum.data_ptr() if stats else None,
            self.grad_accum_dur.data_ptr() if stats else None, _stream()))

    def step_split(self, it, common, rgb_all, uv_norm_sum=None):
        """The W-view step on what the split exchange delivers (ViewShardedStep, exchange="split"): common[N,12] = the
        all-reduced direction-independent columns {xyz 3, opacity, scale 3, quaternion 4, views that saw the gaussian},
        rgb_all[W, N+1, 3] = every view's g_rgb in global order + its camera position.  The colour groups are rebuilt
        and applied by gsplat_optimizer_step_sh_views (first: the directions come from xyz), the other four groups by the
        packed kernel on common itself.  Bit-identical to step_packed on the materialised rows."""
        assert common.is_contiguous() and common.shape[1] == 12 and rgb_all.is_contiguous()
        N, W = int(common.shape[0]), int(rgb_all.shape[0])
        assert rgb_all.shape[1] == N + 1 and N == int(self.params["xyz"].shape[0])
        b1c, b2c = self.bias_corrections(it)
        lrs = self.learning_rates(it)
        has_sh = self.l_max > 0
        ptr = lambda t: t.data_ptr() if t is not None else None
        check(_lib.load().gsplat_optimizer_step_sh_views(
            int(self.l_max), N, W, self.params["xyz"].data_ptr(), rgb_all.data_ptr(), 3 * (N + 1), common.data_ptr(),
            self.params["rgb"].data_ptr(), self.exp_avg["rgb"].data_ptr(), self.exp_avg_sq["rgb"].data_ptr(), lrs["rgb"],
            ptr(self.params["sh"] if has_sh else None), ptr(self.exp_avg.get("sh")), ptr(self.exp_avg_sq.get("sh")),
            lrs["sh"], B1, B2, EPS, b1c, b2c, _stream()))
        names = ("xyz", "opacity", "scale", "quaternion")
        col = dict(xyz=0, opacity=3, scale=4, quaternion=7)  # columns of common (gsplat_hip.h)
        arr = (_lib.AdamGroup * len(names))()
        for k, g in enumerate(names):
            arr[k].param, arr[k].exp_avg, arr[k].exp_avg_sq = (self.params[g].data_ptr(), self.exp_avg[g].data_ptr(),
                                                               self.exp_avg_sq[g].data_ptr())
            arr[k].grad, arr[k].stride, arr[k].packed_column, arr[k].lr = None, self.cols[g][1] - self.cols[g][0], col[g], lrs[g]
        stats = uv_norm_sum is not None
        check(_lib.load().gsplat_optimizer_step_packed(
            common.data_ptr(), N, 12, arr, len(names), B1, B2, EPS, b1c, b2c,
            uv_norm_sum.data_ptr() if stats else None, self.uv_grad_accum.data_ptr() if stats else None,
            self.grad_accum_dur.data_ptr() if stats else None, _stream()))
