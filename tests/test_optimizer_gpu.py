"""GPU tests of the "next" row f2 choreography: the masked in-place optimizer step (gsplat_optimizer_step[_packed])
against the reference's compact -> adam_step -> scatter sequence (cuda/trainer.cu:1027-1158) restated with the
oracle's adam_step on numpy-compacted arrays, and a short end-to-end training run
(rasterize -> fused_loss -> backward_pass -> optimizer step) whose loss must fall."""
import numpy as np
import pytest

from conftest import pkg

pytestmark = pytest.mark.gpu
GROUPS = ("xyz", "rgb", "sh", "opacity", "scale", "quaternion")


def _np(t):
    return t.detach().cpu().numpy()


def _forward_backward(torch, scene, name="small", view_index=0):
    raster = pkg("raster")
    N, W, H, L, _ = scene.WORKLOADS[name]
    params = scene.make_gaussians(N, W, H, L)
    params["xyz"][::3, 2] *= -1  # a third of the scene sits behind the camera: those rows must not be touched
    cam = scene.make_camera(W, H, view_index)
    c = scene.CONFIG
    ctx = raster.RasterContext(N, W, H)
    dp, dc = raster.device_params(params), raster.device_camera(cam)
    fwd = ctx.rasterize_image(dp, dc, c, c["bg"], L)
    grads = ctx.alloc_gradients(fwd["num_culled"], L, intermediates=True)
    ctx.backward_pass(dp, dc, torch.as_tensor(scene.make_grad_image(W, H)).cuda(), c["bg"], L, grads)
    torch.cuda.synchronize()
    return dict(params=params, dp=dp, dc=dc, ctx=ctx, fwd=fwd, grads=grads, N=N, W=W, H=H, L=L)


def _reference_choreography(orc, opt_mod, r, it, m0, v0, lrs):
    """compact by mask -> adam_step -> scatter, per group, on the CPU oracle."""
    mask = _np(r["fwd"]["mask"]).astype(bool)
    b1c, b2c = opt_mod.AdamOptimizer.bias_corrections(it)
    out = {}
    for g in GROUPS:
        p = _np(r["dp"][g]).reshape(r["N"], -1).copy()
        m, v = m0[g].copy(), v0[g].copy()
        grad = _np(r["grads"][g]).reshape(int(mask.sum()), -1)
        pc, mc, vc = orc.adam_step(p[mask], grad, m[mask], v[mask], np.float32(lrs[g]), np.float32(opt_mod.B1),
                                   np.float32(opt_mod.B2), np.float32(opt_mod.EPS), np.float32(b1c), np.float32(b2c))
        p[mask], m[mask], v[mask] = pc.reshape(grad.shape), mc.reshape(grad.shape), vc.reshape(grad.shape)
        out[g] = (p, m, v)
    return out, mask


def test_optimizer_step_matches_compact_adam_scatter(gpu, scene, orc):
    torch, opt_mod = gpu, pkg("optimizer")
    r = _forward_backward(torch, scene, "small", view_index=1)
    opt = opt_mod.AdamOptimizer(r["dp"], r["L"], scene_extent=2.5)
    rng = np.random.default_rng(0)
    m0 = {g: (rng.standard_normal((r["N"], opt.cols[g][1] - opt.cols[g][0])) * 1e-4).astype(np.float32) for g in GROUPS}
    v0 = {g: (rng.random((r["N"], opt.cols[g][1] - opt.cols[g][0])) * 1e-8).astype(np.float32) for g in GROUPS}
    for g in GROUPS:
        opt.exp_avg[g].copy_(torch.from_numpy(m0[g]).reshape(opt.exp_avg[g].shape))
        opt.exp_avg_sq[g].copy_(torch.from_numpy(v0[g]).reshape(opt.exp_avg_sq[g].shape))
    it = 37
    ref, mask = _reference_choreography(orc, opt_mod, r, it, m0, v0, opt.learning_rates(it))
    uv_norm = np.sqrt((_np(r["grads"]["uv"]) ** 2).sum(1))
    opt.step(it, r["fwd"], r["grads"])
    torch.cuda.synchronize()
    assert 0 < mask.sum() < r["N"]  # the view must leave some rows untouched for the test to mean anything
    for g in GROUPS:
        p, m, v = ref[g]
        np.testing.assert_allclose(_np(opt.exp_avg[g]).reshape(p.shape), m, rtol=2e-6, atol=1e-12, err_msg=g)
        np.testing.assert_allclose(_np(opt.exp_avg_sq[g]).reshape(p.shape), v, rtol=2e-6, atol=1e-20, err_msg=g)
        np.testing.assert_allclose(_np(r["dp"][g]).reshape(p.shape), p, rtol=2e-6, atol=1e-7, err_msg=g)
        # rows the view did not see are bit-identical to the start
        assert (_np(opt.exp_avg[g]).reshape(p.shape)[~mask] == m0[g][~mask]).all()
    acc, dur = _np(opt.uv_grad_accum), _np(opt.grad_accum_dur)
    np.testing.assert_allclose(acc[mask], uv_norm, rtol=1e-6)
    assert (acc[~mask] == 0).all() and (dur == mask.astype(np.int32)).all()


def test_packed_step_equals_single_view_step(gpu, scene):
    """With one view the packed (all-reduced) route must update exactly what the compacted route updates."""
    torch, opt_mod = gpu, pkg("optimizer")
    a = _forward_backward(torch, scene, "small", view_index=2)
    twin = {k: v.clone() for k, v in a["dp"].items()}  # same gradients for both routes (atomics reorder between runs)
    oa, ob = opt_mod.AdamOptimizer(a["dp"], a["L"]), opt_mod.AdamOptimizer(twin, a["L"])
    packed = torch.empty(a["N"], ob.width, device="cuda")
    a["ctx"].pack_gradients_global(a["grads"], a["L"], a["N"], packed)
    oa.step(5, a["fwd"], a["grads"])
    ob.step_packed(5, packed)
    torch.cuda.synchronize()
    assert not torch.equal(oa.exp_avg["xyz"], torch.zeros_like(oa.exp_avg["xyz"]))
    for g in GROUPS:
        assert torch.equal(a["dp"][g], twin[g]), g
        assert torch.equal(oa.exp_avg[g], ob.exp_avg[g]) and torch.equal(oa.exp_avg_sq[g], ob.exp_avg_sq[g]), g


@pytest.mark.parametrize("name", ["small", "small_l1"])
def test_factored_sh_step_equals_the_stored_gradient_step(gpu, scene, name):
    """The SH gradients of one view are Y_{k+1}(direction) x grad_precompute_rgb (cuda/spherical_harmonics_backward.cu:168-209).
    A backward that is given NO grad_sh array plus gsplat_optimizer_step_sh_factored (which rebuilds them in the optimizer)
    must leave parameters and both moments of every group BIT-identical to the backward that stores them and the plain
    group that reads them back -- same basis function, same product, same Adam -- and must not touch a culled row."""
    torch, raster, opt_mod = gpu, pkg("raster"), pkg("optimizer")
    if name == "small_l1":
        N, W, H, L = 3000, 160, 96, 1
    else:
        N, W, H, L, _ = scene.WORKLOADS["small"]
    params = scene.make_gaussians(N, W, H, L)
    params["xyz"][::3, 2] *= -1
    cam = scene.make_camera(W, H, 1)
    c = scene.CONFIG
    ctx = raster.RasterContext(N, W, H)
    dp_a, dc = raster.device_params(params), raster.device_camera(cam)
    dp_b = {k: v.clone() for k, v in dp_a.items()}
    gi = torch.as_tensor(scene.make_grad_image(W, H)).cuda()
    fwd = ctx.rasterize_image(dp_a, dc, c, c["bg"], L)
    M = fwd["num_culled"]
    assert 0 < M < N
    g_plain = ctx.alloc_gradients(M, L, intermediates=("uv", "precompute_rgb"))
    g_fact = ctx.alloc_gradients(M, L, intermediates=("uv",), factored_sh=True)
    assert g_fact["sh"] is None and g_fact["precompute_rgb"] is not None
    ctx.backward_pass(dp_a, dc, gi, c["bg"], L, g_plain)
    # the same compositing gradients for both (float atomics reorder between two launches): only the per-gaussian half again
    ctx.backward_gaussians(dp_a, dc, L, g_fact)
    torch.cuda.synchronize()
    for k in ("xyz", "rgb", "opacity", "scale", "quaternion", "uv", "precompute_rgb"):
        assert torch.equal(g_plain[k], g_fact[k]), k  # leaving grad_sh out changes nothing else
    oa, ob = opt_mod.AdamOptimizer(dp_a, L, scene_extent=2.5), opt_mod.AdamOptimizer(dp_b, L, scene_extent=2.5)
    rng = np.random.default_rng(3)
    for g in GROUPS:
        m0 = torch.from_numpy((rng.standard_normal(tuple(oa.exp_avg[g].shape)) * 1e-4).astype(np.float32)).cuda()
        v0 = torch.from_numpy((rng.random(tuple(oa.exp_avg_sq[g].shape)) * 1e-8).astype(np.float32)).cuda()
        oa.exp_avg[g].copy_(m0); ob.exp_avg[g].copy_(m0)
        oa.exp_avg_sq[g].copy_(v0); ob.exp_avg_sq[g].copy_(v0)
    sh_before = dp_b["sh"].clone()
    for it in (11, 12):  # two steps: the second one sees the moved positions, as the next iteration's backward would
        if it == 12:
            fwd = ctx.rasterize_image(dp_a, dc, c, c["bg"], L)   # both twins hold the same parameters
            M = fwd["num_culled"]
            g_plain = ctx.alloc_gradients(M, L, intermediates=("uv", "precompute_rgb"))
            g_fact = ctx.alloc_gradients(M, L, intermediates=("uv",), factored_sh=True)
            ctx.backward_pass(dp_a, dc, gi, c["bg"], L, g_plain)
            ctx.backward_gaussians(dp_a, dc, L, g_fact)
        oa.step(it, fwd, g_plain)
        ob.step(it, fwd, g_fact, campos=cam["campos"])
        torch.cuda.synchronize()
        for g in GROUPS:
            assert torch.equal(dp_a[g], dp_b[g]), (it, g)
            assert torch.equal(oa.exp_avg[g], ob.exp_avg[g]) and torch.equal(oa.exp_avg_sq[g], ob.exp_avg_sq[g]), (it, g)
    mask = _np(fwd["mask"]).astype(bool)
    assert not torch.equal(dp_b["sh"], sh_before)
    assert torch.equal(dp_b["sh"][torch.from_numpy(~mask).cuda()], sh_before[torch.from_numpy(~mask).cuda()])
    with pytest.raises(ValueError):
        ob.step(13, fwd, g_fact)  # no camera position: the direction cannot be rebuilt


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("name", ["small", "small_l1", "small_l0", "mid_l2"])
def test_backward_with_adam_inside_equals_backward_then_optimizer(gpu, scene, name, mode):
    """r06: gsplat_backward_gaussians_adam -- the per-gaussian backward that applies the masked Adam step itself -- must
    leave parameters, both moments of all six groups and the densification statistics BIT-identical to the backward that
    stores its gradients followed by gsplat_optimizer_step_sh_factored + gsplat_optimizer_step (the same gradient values,
    gs::adam_values in both), must not touch a culled row, and fills the gradient arrays it is given exactly as the plain
    backward does.  Both twins differentiate the SAME compositing rows (one gsplat_backward_render: its float atomics
    reorder between launches).  Two steps: the second sees the moved parameters."""
    torch, raster, opt_mod = gpu, pkg("raster"), pkg("optimizer")
    N, W, H, L = {"small": (5000, 256, 144, 3), "small_l1": (3000, 160, 96, 1), "small_l0": (3000, 160, 96, 0),
                  "mid_l2": (40000, 320, 192, 2)}[name]
    params = scene.make_gaussians(N, W, H, L)
    params["xyz"][::3, 2] *= -1
    cam = scene.make_camera(W, H, 1)
    c = scene.CONFIG
    ctx = raster.RasterContext(N, W, H)
    dp_a, dc = raster.device_params(params), raster.device_camera(cam)
    dp_b = {k: v.clone() for k, v in dp_a.items()}
    gi = torch.as_tensor(scene.make_grad_image(W, H)).cuda()
    oa, ob = opt_mod.AdamOptimizer(dp_a, L, scene_extent=2.5), opt_mod.AdamOptimizer(dp_b, L, scene_extent=2.5)
    rng = np.random.default_rng(5)
    for g in oa.names:
        m0 = torch.from_numpy((rng.standard_normal(tuple(oa.exp_avg[g].shape)) * 1e-4).astype(np.float32)).cuda()
        v0 = torch.from_numpy((rng.random(tuple(oa.exp_avg_sq[g].shape)) * 1e-8).astype(np.float32)).cuda()
        oa.exp_avg[g].copy_(m0); ob.exp_avg[g].copy_(m0)
        oa.exp_avg_sq[g].copy_(v0); ob.exp_avg_sq[g].copy_(v0)
    before = {k: v.clone() for k, v in dp_b.items()}
    for it in (20, 21):
        fwd = ctx.rasterize_image(dp_a, dc, c, c["bg"], L)  # (both twins hold the same parameters)
        M = fwd["num_culled"]
        assert 0 < M < N
        ctx.backward_render(gi, c["bg"])
        g_a = ctx.alloc_gradients(M, L, intermediates=("uv",), factored_sh=True)
        g_b = ctx.alloc_gradients(M, L, intermediates=("uv",), factored_sh=True)
        for t in g_b.values():
            if t is not None:
                t.fill_(float("nan"))
        ctx.backward_gaussians(dp_a, dc, L, g_a)
        oa.step(it, fwd, g_a, campos=cam["campos"])
        if mode in (0, 2):  # twin B: one kernel (0), or the SH group's kernel in front of the backward with the small
            # groups' steps (2: sh_adam_dir_kernel reads the coefficient rows once and hands the position gradient through
            # the view direction on); on the second step without gradient arrays at all (what the trainer does)
            ctx.backward_gaussians_adam(dp_b, dc, L, ob.fused_state(it, mode=mode), g_b if it == 20 else None)
        else:  # mode 1: four groups + statistics in the kernel, then the SH and the position group behind it
            g_part = g_b if it == 20 else dict(xyz=g_b["xyz"], precompute_rgb=g_b.get("precompute_rgb"))
            ctx.backward_gaussians_adam(dp_b, dc, L, ob.fused_state(it, mode=1), g_part)
            ob.step_after_partial_backward(it, fwd, g_part, cam["campos"])
        torch.cuda.synchronize()
        if it == 20:
            for k in ("xyz", "rgb", "opacity", "scale", "quaternion", "uv", "precompute_rgb"):
                if g_a.get(k) is not None:  # (no precompute_rgb array at degree 0)
                    assert torch.equal(g_a[k], g_b[k]), k
        for g in oa.names:
            assert torch.equal(dp_a[g], dp_b[g]), (it, g)
            assert torch.equal(oa.exp_avg[g], ob.exp_avg[g]) and torch.equal(oa.exp_avg_sq[g], ob.exp_avg_sq[g]), (it, g)
        assert torch.equal(oa.uv_grad_accum, ob.uv_grad_accum) and torch.equal(oa.grad_accum_dur, ob.grad_accum_dur), it
    culled = torch.from_numpy(~_np(fwd["mask"]).astype(bool)).cuda()
    assert int(ob.grad_accum_dur.max()) == 2 and int(ob.grad_accum_dur[culled].max()) <= 1
    for g in oa.names:
        assert not torch.equal(dp_b[g], before[g]), g


def test_training_iterations_reduce_the_loss(gpu, scene):
    """rasterize -> fused_loss -> backward -> optimizer step, 30 iterations on one view toward a target rendered
    from the unperturbed scene: the L1+SSIM loss must drop and PSNR must rise (reference loop: trainer.cu:417-516)."""
    torch, raster, ops, opt_mod = gpu, pkg("raster"), pkg("ops"), pkg("optimizer")
    N, W, H, L, _ = scene.WORKLOADS["small"]
    c = scene.CONFIG
    truth = scene.make_gaussians(N, W, H, L)
    cam = raster.device_camera(scene.make_camera(W, H, 0))
    ctx = raster.RasterContext(N, W, H)
    target = ctx.rasterize_image(raster.device_params(truth), cam, c, c["bg"], L)["image"].clone()
    start = {k: np.array(v, copy=True) for k, v in truth.items()}
    rng = np.random.default_rng(1)
    start["rgb"] = start["rgb"] + rng.normal(0, 0.5, start["rgb"].shape).astype(np.float32)
    start["opacity"] = start["opacity"] + rng.normal(0, 0.5, start["opacity"].shape).astype(np.float32)
    dp = raster.device_params(start)
    opt = opt_mod.AdamOptimizer(dp, L, scene_extent=1.0)
    grad_image = torch.empty(H, W, 3, device="cuda")
    losses, psnrs = [], []
    for it in range(30):
        fwd = ctx.rasterize_image(dp, cam, c, c["bg"], L)
        losses.append(ops.fused_loss(fwd["image"], target, H, W, 0.2, grad_image))
        psnrs.append(ops.compute_psnr(fwd["image"], target, H, W))
        grads = ctx.alloc_gradients(fwd["num_culled"], L, intermediates=True)
        ctx.backward_pass(dp, cam, grad_image, c["bg"], L, grads)
        opt.step(it, fwd, grads)
    assert all(np.isfinite(losses))
    assert losses[-1] < 0.7 * losses[0], losses
    assert psnrs[-1] > psnrs[0] + 1.0, psnrs


def test_optimizer_argument_errors(gpu, scene):
    torch, opt_mod, lib_mod = gpu, pkg("optimizer"), pkg("_lib")
    lib = lib_mod.load()
    grp = (lib_mod.AdamGroup * 1)()
    assert lib.gsplat_optimizer_step(None, 0, grp, 0, 0.9, 0.999, 1e-8, 0.1, 0.001, None, None, None, None) == -3
    assert lib.gsplat_optimizer_step(None, 4, grp, 1, 0.9, 0.999, 1e-8, 0.1, 0.001, None, None, None, None) == -3  # stride 0
    grp[0].stride = 3
    assert lib.gsplat_optimizer_step(None, 4, grp, 1, 0.9, 0.999, 1e-8, 0.1, 0.001, None, None, None, None) == -1  # NULL param


@pytest.mark.parametrize("name,world", [("small", 3), ("small_l1", 2), ("tiny_l0", 2)])
def test_split_step_equals_packed_rows_step(gpu, scene, name, world):
    """A W-view step on the split exchange's own factored form -- common[N,12] summed over the views + every view's
    g_rgb and camera position (rgb_all) -- through AdamOptimizer.step_split (gsplat_optimizer_step_sh_views rebuilds
    sum_r g_rgb^r x Y_k(dir^r) inside the colour groups' Adam, the other four groups read common directly) must leave
    parameters, both moments and the densification statistics BIT-identical to materialising packed[N, 12 + 3 n]
    (gsplat_unpack_gradients_split) and running gsplat_optimizer_step_packed on it (cuda/trainer.cu:1027-1158 per view).
    W views on one GPU, two steps (the second on moved gaussians), rows no view saw untouched."""
    torch, raster, opt_mod = gpu, pkg("raster"), pkg("optimizer")
    N, W, H, L = {"small": (5000, 256, 144, 3), "small_l1": (5000, 256, 144, 1), "tiny_l0": (600, 96, 64, 0)}[name]
    params = scene.make_gaussians(N, W, H, L)
    params["xyz"][::3, 2] *= -1  # a third behind every camera
    c = scene.CONFIG
    ctx = raster.RasterContext(N, W, H)
    ctx.set_lean_forward(True)
    dp = raster.device_params(params)
    twin = {k: v.clone() for k, v in dp.items()}
    gi = torch.as_tensor(scene.make_grad_image(W, H)).cuda()
    cams = [raster.device_camera(scene.make_camera(W, H, v + 1)) for v in range(world)]
    oa, ob = opt_mod.AdamOptimizer(dp, L), opt_mod.AdamOptimizer(twin, L)
    wc = raster.packed_gradient_width(L)
    for it in (0, 1):
        common = torch.zeros(N, 12, device="cuda")
        uv_sum = torch.zeros(N, device="cuda")
        rgb_all = torch.full((world, N + 1, 3), float("nan"), device="cuda")
        for r, cam in enumerate(cams):  # the exchange by hand: sum of the views' common rows, gather of their g_rgb
            com_r, uv_r = torch.full((N, 12), float("nan"), device="cuda"), torch.full((N,), float("nan"), device="cuda")
            ctx.rasterize_image(dp, cam, c, c["bg"], L)
            ctx.backward_render(gi, c["bg"], rgb_all[r], com_r, uv_r)
            ctx.backward_gaussians_split(dp, cam, L, com_r, uv_r)
            rgb_all[r, N] = cam["campos_dev"]
            common += com_r
            uv_sum += uv_r
        assert torch.isfinite(common).all() and torch.isfinite(rgb_all).all()
        seen = common[:, 11] > 0
        assert 0 < int(seen.sum()) < N and int(common[:, 11].max()) == world
        packed = torch.full((N, wc), float("nan"), device="cuda")
        raster.unpack_gradients_split(dp["xyz"], common, rgb_all, 3 * (N + 1), L, N, world, packed)
        before = {g: twin[g].clone() for g in oa.names}
        oa.step_split(it, common, rgb_all, uv_sum)
        ob.step_packed(it, packed, uv_sum)
        torch.cuda.synchronize()
        for g in oa.names:
            assert torch.equal(dp[g], twin[g]), (it, g)
            assert torch.equal(oa.exp_avg[g], ob.exp_avg[g]) and torch.equal(oa.exp_avg_sq[g], ob.exp_avg_sq[g]), (it, g)
            moved = (twin[g].reshape(N, -1) != before[g].reshape(N, -1)).any(1)
            assert not moved[~seen].any() and moved[seen].any(), (it, g)
        assert torch.equal(oa.uv_grad_accum, ob.uv_grad_accum) and torch.equal(oa.grad_accum_dur, ob.grad_accum_dur)
        assert int(oa.grad_accum_dur.max()) == world * (it + 1)
