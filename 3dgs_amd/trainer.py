"""Host-side mirror of the reference's training loop (TrainerImpl, cuda/trainer.cu) -- SURVEY.md section 8f, row f4.

Only the SCHEDULE lives here (what runs at which iteration, config thresholds); every data-parallel step is a HIP
kernel behind the C ABI: rasterize / loss / backward / masked Adam (rows a-f2) and, for the density policy, the masks,
clone / split, compaction, SH band re-layout, Morton codes and the row gather of gs_density.hip.  torch is used for
device memory and for the key sort of the Morton re-order.  There is no CPU fallback.

Parity note: the reference's density control is not reproducible (cuRAND seeded from time(NULL), std::random_device for
the view order), so this loop is judged statistically (PSNR, gaussian counts), not bit for bit; with a fixed seed it is
itself deterministic.
"""
import math
import os

import numpy as np
import torch

from . import ops, raster
from .dist import TorchComm, ViewShardedStep
from .optimizer import AdamOptimizer, DEFAULT_LR

GROUPS = ("xyz", "rgb", "sh", "opacity", "scale", "quaternion")

# config/base.yaml of the reference: the keys the loop reads
DEFAULT_CONFIG = dict(
    DEFAULT_LR, near_thresh=0.3, mh_dist=3.0, cull_mask_padding=100, ssim_frac=0.2, use_background=True,
    use_background_end=2000, reset_opacity_interval=3000, reset_opacity_value=0.05, reset_opacity_start=1050,
    reset_opacity_end=5000, max_sh_band=3, add_sh_band_interval=1000, use_split=True, use_clone=True, use_delete=True,
    adaptive_control_start=500, adaptive_control_end=5000, adaptive_control_interval=100, max_gaussians=4250000,
    delete_opacity_threshold=0.02, uv_grad_threshold=0.0002, split_scale_factor=1.6)


def _logit(p):
    return math.log(p) - math.log(1.0 - p)


def draw_view_indices(rng, iteration, world, num_views):
    """View indices of one iteration, one per rank, identical on every rank that shares the generator's seed.  The
    reference walks the first two images in order and then draws uniformly (cuda/trainer.cu:1438-1444); with W ranks
    an iteration consumes W consecutive draws of that sequence (rank r trains on the r-th)."""
    draws = []
    for k in range(world):
        g = iteration * world + k
        draws.append(g if g < 2 else int(rng.integers(0, num_views)))
    return [d % num_views for d in draws]


class Trainer:
    """params: dict of device tensors (xyz rgb opacity scale quaternion, optionally sh) as gsplat_initialize_gaussians
    returns them; views: list of (camera dict for raster.device_camera, ground-truth image tensor [H,W,3] on device)."""

    def __init__(self, params, views, config=None, scene_extent=1.0, seed=0, exchange="split", comm=None):
        """With an initialised torch.distributed group of W > 1 ranks the loop is view-sharded: every
        rank holds the same parameters and the same seed, one iteration = W training views (rank r takes the r-th of
        the iteration's W draws), the per-gaussian gradients and |grad_uv| are summed over the ranks
        (ViewShardedStep) and every rank applies the identical packed optimizer step and density control, so the
        replicas stay bit-identical without ever exchanging parameters."""
        self.cfg = dict(DEFAULT_CONFIG, **(config or {}))
        self.views = views
        # the ranks of the group: torch.distributed's default group, or in-process rank threads (dist.ThreadGroup)
        self.comm = comm if comm is not None else TorchComm()
        self.world, self.rank = self.comm.world, self.comm.rank
        self.exchange = exchange
        # r06, one rank: where the optimizer step runs (GSPLAT_FUSED_ADAM; every form leaves parameters, moments and
        # statistics bit-identical -- tests/test_optimizer_gpu.py; kernel times at 1e6 gaussians, SH 3, from one trace:
        # profiles/r06_two_kernel_adam.txt):
        #   1 (default): band 0, opacity, scale, rotation and the statistics inside the per-gaussian backward, then
        #     gsplat_optimizer_step_sh_factored (which takes its directions from positions that must not have moved yet) and
        #     gsplat_optimizer_step on the position group alone: 122 + 183 + 24 us against 72 + 176 + 94 -- 20 us of an
        #     iteration's 1100, and only the position and colour gradients are stored;
        #   3: two kernels -- sh_adam_dir_kernel in front reads the SH rows once (the SH group's step and the sums over the
        #     rows that the position gradient needs), then the backward with the five small groups' steps and no SH rows:
        #     219 + 113 us, 10 us of the 1100, no gradient arrays at all;
        #   2: all six groups inside the backward (one fat kernel at three waves per SIMD: 356 us, 2 % slower);
        #   0: the backward stores its gradients, the two optimizer kernels read them (r01-r05).
        self.fused_adam = int(__import__("os").environ.get("GSPLAT_FUSED_ADAM", "1") or 0)
        self._sharded = None  # (key, ViewShardedStep) for the current gaussian count / SH degree
        self._grad_image = {}  # (H, W) -> dL/dimage buffer, allocated once per image size
        self._grads = None     # (capacity, l_max, dict): per-view gradient arrays, reused across iterations
        self.scene_extent = float(scene_extent)
        self.seed = int(seed)
        self.rng = np.random.default_rng(seed)
        self.iter = 0
        self.l_max = 0 if params.get("sh") is None or params["sh"].numel() == 0 else \
            int(round(math.sqrt(params["sh"].shape[1] + 1))) - 1
        self.params = {k: params[k].contiguous() for k in ("xyz", "rgb", "opacity", "scale", "quaternion")}
        n = self.num_gaussians
        self.params["sh"] = params["sh"].contiguous() if self.l_max > 0 else \
            torch.zeros(n, 0, 3, dtype=torch.float32, device=self.params["xyz"].device)
        W = max(int(c["width"]) for c, _ in views)
        H = max(int(c["height"]) for c, _ in views)
        self.ctx = raster.RasterContext(max(n, 1), W, H)
        self.ctx.set_lean_forward(True)  # the loop only runs the fused backward, which recomputes Sigma / J / conic
        self.ctx_capacity = n
        self._new_optimizer(None)
        self.history = []

    # ------------------------------------------------------------------ state
    @property
    def num_gaussians(self):
        return int(self.params["xyz"].shape[0])

    def _new_optimizer(self, moments):
        """Fresh AdamOptimizer over the current parameter tensors; `moments` = (exp_avg, exp_avg_sq) dicts to adopt."""
        self._sharded = None  # the parameter tensors were replaced: the exchange step binds to the new ones
        p = dict(self.params)
        if self.l_max == 0:
            p.pop("sh")
        self.opt = AdamOptimizer(p, self.l_max, lr_config=self.cfg, scene_extent=self.scene_extent)
        if moments is not None:
            for g in self.opt.names:
                self.opt.exp_avg[g], self.opt.exp_avg_sq[g] = moments[0][g], moments[1][g]

    def _context_for(self, n):
        if n > self.ctx_capacity:  # the workspace is sized for a gaussian count: grow it geometrically
            self.ctx_capacity = int(n * 1.5) + 1
            W, H = self.ctx.max_width, self.ctx.max_height
            self.ctx = raster.RasterContext(self.ctx_capacity, W, H)
            self.ctx.set_lean_forward(True)
        return self.ctx

    # ------------------------------------------------------------------ one iteration (cuda/trainer.cu:1338-1362)
    def _grad_image_for(self, H, W, device):
        buf = self._grad_image.get((H, W))
        if buf is None:
            buf = self._grad_image[(H, W)] = torch.empty(H, W, 3, dtype=torch.float32, device=device)
        return buf

    def _gradients_for(self, ctx, m):
        """Per-view gradient arrays in compacted order: allocated for the current gaussian count, sliced per view."""
        n = self.num_gaussians
        if self._grads is None or self._grads[0] < n or self._grads[1] != self.l_max:
            # density statistics need |grad_uv|; the SH gradients are not stored: the optimizer rebuilds them from the 24
            # bytes they are made of (AdamOptimizer.step, gsplat_optimizer_step_sh_factored) -- single-GPU training only,
            # the view-sharded step exchanges packed rows
            self._grads = (n, self.l_max, ctx.alloc_gradients(n, self.l_max, intermediates=("uv",), factored_sh=True))
        return {k: (v[:m] if v is not None else None) for k, v in self._grads[2].items()}

    def _partial_gradients_for(self, ctx, m):
        """The two arrays the optimizer kernels behind a partial backward_pass_adam read: grad_xyz, grad_precompute_rgb."""
        n = self.num_gaussians
        if getattr(self, "_grads2", None) is None or self._grads2[0] < n:
            dev = self.params["xyz"].device
            self._grads2 = (n, dict(xyz=torch.empty(n, 3, device=dev), precompute_rgb=torch.empty(n, 3, device=dev)))
        return {k: v[:m] for k, v in self._grads2[1].items()}

    def train_step(self, cam, gt_image, want_loss=True):
        if self.world > 1:
            return self._train_step_sharded(cam, gt_image, want_loss)
        c = self.cfg
        it = self.iter
        # cuda/trainer.cu:1341-1343: the background cycles for the whole run (use_background_end is parsed but never read)
        bg = (it % 255) / 255.0 if c["use_background"] else 0.0
        if it % c["add_sh_band_interval"] == 0 and it >= c["add_sh_band_interval"]:
            self.add_sh_band()
        ctx = self._context_for(self.num_gaussians)
        p = dict(self.params)
        H, W = int(cam["height"]), int(cam["width"])
        try:
            fwd = ctx.rasterize_image(p, cam, c, bg, self.l_max)
        except Exception as e:  # no gaussian in view: the reference warns and skips the iteration
            if getattr(e, "code", None) != -5:
                raise
            self.iter += 1
            return None
        grad_image = self._grad_image_for(H, W, gt_image.device)
        # the loss value is a blocking read-back: only fetched when the caller logs it
        loss = ops.fused_loss(fwd["image"], gt_image, H, W, float(c["ssim_frac"]), grad_image, blocking=want_loss)
        if self.fused_adam == 3:
            # r06: the SH group's step in a kernel that reads the coefficient rows once (update + the sums the position
            # gradient needs), then the per-gaussian backward with the five small groups' steps inside: no gradient arrays
            ctx.backward_pass_adam(p, cam, grad_image, bg, self.l_max, self.opt.fused_state(it, mode=2))
        elif self.fused_adam == 2:
            # r06: the per-gaussian backward applies the optimizer step itself (no gradient arrays at all)
            ctx.backward_pass_adam(p, cam, grad_image, bg, self.l_max, self.opt.fused_state(it))
        elif self.fused_adam == 1:
            g2 = self._partial_gradients_for(ctx, fwd["num_culled"])
            ctx.backward_pass_adam(p, cam, grad_image, bg, self.l_max, self.opt.fused_state(it, mode=1), g2)
            self.opt.step_after_partial_backward(it, fwd, g2, cam["campos"])
        else:
            grads = self._gradients_for(ctx, fwd["num_culled"])
            ctx.backward_pass(p, cam, grad_image, bg, self.l_max, grads)
            self.opt.step(it, fwd, grads, campos=cam["campos"])
        self.iter += 1
        return loss

    def _train_step_sharded(self, cam, gt_image, want_loss):
        """One iteration of a W-rank group: this rank's view -> loss -> backward -> exchange -> the same optimizer step on
        every rank, incl. the densification statistics (split exchange: AdamOptimizer.step_split on common + rgb_all;
        the other payloads: gsplat_optimizer_step_packed on the packed rows)."""
        c, it = self.cfg, self.iter
        bg = (it % 255) / 255.0 if c["use_background"] else 0.0
        if it % c["add_sh_band_interval"] == 0 and it >= c["add_sh_band_interval"]:
            self.add_sh_band()
        n = self.num_gaussians
        key = (n, self.l_max)
        if self._sharded is None or self._sharded[0] != key:
            ctx = self._context_for(n)
            p = dict(self.params)
            self._sharded = (key, ViewShardedStep(p, self.l_max, ctx.max_width, ctx.max_height, c, 0.0,
                                                  exchange=self.exchange, with_uv_norm=True, ctx=ctx, comm=self.comm))
        step = self._sharded[1]
        H, W = int(cam["height"]), int(cam["width"])
        out = {}

        def grad_fn(fwd):
            g = self._grad_image_for(H, W, gt_image.device)
            out["loss"] = ops.fused_loss(fwd["image"], gt_image, H, W, float(c["ssim_frac"]), g, blocking=want_loss)
            return g

        step.step(cam, grad_fn=grad_fn, bg=bg)
        if step.exchange == "split" and not step.materialize:
            # the exchange's own factored form feeds the optimizer: the colour groups rebuild sum_r g_rgb^r x Y_k(dir^r)
            # inside their Adam (gsplat_optimizer_step_sh_views), no packed[N, 12 + 3 n] rows are written or read
            self.opt.step_split(it, step.common, step.rgb_all, step.uv_norm_sum)
        else:
            self.opt.step_packed(it, step.packed, step.uv_norm_sum)
        self.iter += 1
        return out.get("loss")

    def maintenance(self):
        """Density control and opacity reset at the reference's cadence (cuda/trainer.cu:1394-1406); call after
        train_step.  Uses the iteration index the step just ran with."""
        c, it = self.cfg, self.iter - 1
        if it > c["adaptive_control_start"] and it % c["adaptive_control_interval"] == 0 and it < c["adaptive_control_end"]:
            self.adaptive_density_step()
            self.sort_gaussians()
            self.reset_grad_accum()
        if it > c["reset_opacity_start"] and it % c["reset_opacity_interval"] == 0 and it < c["reset_opacity_end"]:
            self.reset_opacity()
            self.reset_grad_accum()

    def draw_views(self):
        return draw_view_indices(self.rng, self.iter, self.world, len(self.views))

    def train(self, num_iters, log_every=0, loss_every=1, eval_every=0, eval_views=None, on_eval=None):
        """loss_every: read the loss value back every that many iterations (each read blocks the host; 0 = never);
        eval_every / eval_views / on_eval: `evaluate(eval_views)` at that cadence (the reference: every 3000
        iterations, cuda/trainer.cu:1388), reported through on_eval(iteration, psnr)."""
        for _ in range(num_iters):
            v = self.draw_views()[self.rank]
            cam, gt = self.views[v]
            want = bool(loss_every) and (self.iter % loss_every == 0 or (log_every and (self.iter + 1) % log_every == 0))
            loss = self.train_step(cam, gt, want_loss=want)
            if eval_every and (self.iter - 1) % eval_every == 0 and eval_views:
                psnr = self.evaluate(eval_views)
                if on_eval:
                    on_eval(self.iter - 1, psnr)
            self.maintenance()
            if loss is not None:
                self.history.append((self.iter, loss, self.num_gaussians))
            if log_every and self.iter % log_every == 0 and self.rank == 0:
                print(f"iter {self.iter}: loss {loss} gaussians {self.num_gaussians}", flush=True)
        return self.history

    def evaluate(self, views=None):
        """Mean PSNR over the views at background 0 (TrainerImpl::evaluate, cuda/trainer.cu:263-360)."""
        total, views = 0.0, (views or self.views)
        ctx = self._context_for(self.num_gaussians)
        ctx.set_render_only(True)  # nothing here runs a backward
        try:
            for cam, gt in views:
                fwd = ctx.rasterize_image(dict(self.params), cam, self.cfg, 0.0, self.l_max)
                total += ops.compute_psnr(fwd["image"], gt, int(cam["height"]), int(cam["width"]))
        finally:
            ctx.set_render_only(False)
        return total / len(views)

    # ------------------------------------------------------------------ policy steps
    def reset_grad_accum(self):  # cuda/trainer.cu:233-236
        self.opt.uv_grad_accum.zero_()
        self.opt.grad_accum_dur.zero_()

    def reset_opacity(self):  # cuda/trainer.cu:238-245
        self.params["opacity"].fill_(_logit(float(self.cfg["reset_opacity_value"])))
        self.opt.exp_avg["opacity"].zero_()
        self.opt.exp_avg_sq["opacity"].zero_()

    def add_sh_band(self):  # cuda/trainer.cu:377-413
        if self.l_max >= self.cfg["max_sh_band"]:
            return
        old = self.l_max
        m = ({g: self.opt.exp_avg[g] for g in self.opt.names}, {g: self.opt.exp_avg_sq[g] for g in self.opt.names})
        acc = (self.opt.uv_grad_accum, self.opt.grad_accum_dur)
        self.params["sh"] = ops.expand_sh(self.params["sh"], old)
        if old == 0:
            m[0]["sh"], m[1]["sh"] = torch.zeros_like(self.params["sh"]), torch.zeros_like(self.params["sh"])
        else:
            m[0]["sh"], m[1]["sh"] = ops.expand_sh(m[0]["sh"], old), ops.expand_sh(m[1]["sh"], old)
        self.l_max = old + 1
        self._new_optimizer(m)
        self.opt.uv_grad_accum, self.opt.grad_accum_dur = acc

    def adaptive_density_step(self):  # cuda/trainer.cu:518-779
        c, n = self.cfg, self.num_gaussians
        if os.environ.get("GSPLAT_DEBUG_DENSITY") and self.rank == 0:
            avg = (self.opt.uv_grad_accum / self.opt.grad_accum_dur.clamp(min=1).float())
            q = torch.quantile(avg[:: max(1, n // 500000)], torch.tensor([0.5, 0.9, 0.99, 0.999], device=avg.device)).tolist()
            print(f"[density] iter {self.iter}: n {n}, avg |grad_uv| quantiles 50/90/99/99.9 % = "
                  + " ".join(f"{v:.2e}" for v in q) + f", threshold {c['uv_grad_threshold']:.1e}, mean views seen "
                  f"{self.opt.grad_accum_dur.float().mean().item():.1f}", flush=True)
        max_scale = self.scene_extent * 0.1
        clone_thresh = self.scene_extent * 0.01
        prune, clone, split, keep, (n_prune, n_clone, n_split) = ops.density_masks(
            self.params["opacity"], self.params["scale"], self.opt.uv_grad_accum, self.opt.grad_accum_dur,
            _logit(float(c["delete_opacity_threshold"])), max_scale, float(c["uv_grad_threshold"]), clone_thresh)
        if not c["use_delete"]:
            n_prune = 0
            keep = (1 - split).to(torch.uint8)
        if not c["use_clone"]:
            n_clone, clone = 0, torch.zeros_like(clone)
        if not c["use_split"]:
            n_split, split = 0, torch.zeros_like(split)
            keep = (1 - prune).to(torch.uint8) if c["use_delete"] else torch.ones_like(keep)
        n_add = n_clone + 2 * n_split
        new_n = n - n_prune - n_split + n_add
        if new_n > c["max_gaussians"] or (n_add == 0 and n_prune == 0):
            return dict(pruned=0, cloned=0, split=0, skipped=new_n > c["max_gaussians"])
        nsh = (self.l_max + 1) ** 2 - 1 if self.l_max > 0 else 0
        dev = self.params["xyz"].device
        src = dict(self.params)
        src["sh"] = self.params["sh"].reshape(n, -1)

        def fresh(rows):
            return dict(xyz=torch.empty(rows, 3, device=dev), rgb=torch.empty(rows, 3, device=dev),
                        opacity=torch.empty(rows, device=dev), scale=torch.empty(rows, 3, device=dev),
                        quaternion=torch.empty(rows, 4, device=dev), sh=torch.empty(rows, nsh * 3, device=dev))

        def write_ids(mask):
            m = mask.to(torch.int32)
            return (torch.cumsum(m, 0, dtype=torch.int32) - m).contiguous()

        clones, splits = fresh(n_clone), fresh(2 * n_split)
        if n_clone:
            ops.clone_gaussians(n, nsh, clone, write_ids(clone), src, clones)
        if n_split:
            ops.split_gaussians(n, float(c["split_scale_factor"]), nsh, split, write_ids(split), src, splits,
                                seed=self.seed * 1000003 + self.iter)
        keep_n = n - n_prune - n_split

        def rebuild(t, stride, new_rows):
            if stride == 0:  # no SH coefficients yet
                return torch.empty(new_n, 0, device=dev)
            kept = ops.compact_masked_array(stride, t.reshape(-1), keep, keep_n)
            parts = [kept.reshape(keep_n, stride)] + [r.reshape(-1, stride) for r in new_rows]
            return torch.cat(parts, 0)

        shape = dict(xyz=3, rgb=3, opacity=1, scale=3, quaternion=4, sh=nsh * 3)
        new_params, m, v = {}, {}, {}
        for g in GROUPS:
            s = shape[g]
            new_params[g] = rebuild(self.params[g], s, [clones[g], splits[g]])
            if g in self.opt.names:  # kept rows keep their moments, new rows start at zero (cuda/trainer.cu:703-742)
                z = [torch.zeros(n_clone, s, device=dev), torch.zeros(2 * n_split, s, device=dev)]
                m[g] = rebuild(self.opt.exp_avg[g], s, z)
                v[g] = rebuild(self.opt.exp_avg_sq[g], s, z)
        self.params = dict(xyz=new_params["xyz"], rgb=new_params["rgb"], opacity=new_params["opacity"].reshape(-1),
                           scale=new_params["scale"], quaternion=new_params["quaternion"],
                           sh=new_params["sh"].reshape(new_n, nsh, 3))
        for g in m:
            tgt = self.params[g].shape
            m[g], v[g] = m[g].reshape(tgt).contiguous(), v[g].reshape(tgt).contiguous()
        self._new_optimizer((m, v))
        return dict(pruned=n_prune, cloned=n_clone, split=n_split, skipped=False)

    def sort_gaussians(self):  # cuda/trainer.cu:853-922: Morton order of the positions
        n = self.num_gaussians
        if n == 0:
            return
        xyz = self.params["xyz"]
        lo, hi = xyz.min(0).values.tolist(), xyz.max(0).values.tolist()
        codes = torch.empty(n, dtype=torch.int64, device=xyz.device)
        ops.compute_morton_codes(n, xyz, hi[0], hi[1], hi[2], lo[0], lo[1], lo[2], codes)
        order = torch.sort(codes, stable=True).indices.to(torch.int32).contiguous()  # codes use 63 bits: int64 order is unsigned order
        for g in GROUPS:
            if self.params[g].numel():
                self.params[g] = ops.gather_rows(self.params[g], order)
        m = ({g: ops.gather_rows(self.opt.exp_avg[g], order) for g in self.opt.names},
             {g: ops.gather_rows(self.opt.exp_avg_sq[g], order) for g in self.opt.names})
        acc = (self.opt.uv_grad_accum, self.opt.grad_accum_dur)
        self._new_optimizer(m)
        self.opt.uv_grad_accum = ops.gather_rows(acc[0], order)
        self.opt.grad_accum_dur = ops.gather_rows(acc[1].view(torch.float32), order).view(torch.int32)

    def save_to_ply(self, path):
        """TrainerImpl::save_to_ply (cuda/trainer.cu:1166-1196): the device quaternion (kernel order w,x,y,z) is
        memcpy'd into Eigen's (x,y,z,w) storage and normalised, so the file's rot_0..3 are the unit quaternion in
        DEVICE order.  gsplat_save_ply writes its (w,x,y,z) input as (x,y,z,w) (Gaussians' Eigen convention), so the
        columns are pre-rotated here to land unchanged."""
        from . import dataset
        p = {k: v.detach().cpu().numpy() for k, v in self.params.items()}
        q = p["quaternion"].astype(np.float32)
        norm = np.sqrt((q.astype(np.float64) ** 2).sum(1, keepdims=True))
        q = (q / np.where(norm > 0, norm, 1.0)).astype(np.float32)
        dataset.save_ply(path, p["xyz"], p["rgb"], p["opacity"], p["scale"], np.roll(q, 1, axis=1),
                         p["sh"].reshape(len(p["xyz"]), -1) if self.l_max > 0 else None)
