"""Two ranks on ONE GPU (gloo for the collectives, since RCCL wants one device per rank): the whole view-sharded step
-- rasterize, split backward with the g_rgb all-gather started behind the per-gaussian backward, all-reduce, local SH
rebuild -- must leave the same packed gradients as the plain full-row all-reduce, identically on both ranks."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), GSPLAT_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    import importlib
    import torch
    scene = importlib.import_module("3dgs_amd.scene")
    raster = importlib.import_module("3dgs_amd.raster")
    gdist = importlib.import_module("3dgs_amd.dist")
    gdist.init_from_env()
    N, W, H, L = 4000, 160, 96, 3
    params = scene.make_gaussians(N, W, H, L)
    params["xyz"][::9, 2] *= -1
    cam = raster.device_camera(scene.make_camera(W, H, view_index=rank + 1))
    dp = raster.device_params(params)
    gi = torch.as_tensor(scene.make_grad_image(W, H)).cuda()
    c = scene.CONFIG
    out = {}
    for ex in ("split", "factored", "full"):
        step = gdist.ViewShardedStep(dp, L, W, H, c, c["bg"], exchange=ex)
        step.step(cam, gi)
        step.step(cam, gi)  # twice: buffers are reused
        torch.cuda.synchronize()
        out[ex] = step.packed.cpu().numpy().copy()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **out)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_view_sharded_step_two_ranks_one_gpu(tmp_path):
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"rank{k}.npz") for k in range(world)]
    for ex in ("split", "factored", "full"):
        assert (r[0][ex] == r[1][ex]).all(), f"{ex}: ranks disagree"
    full = r[0]["full"]
    scale = np.abs(full).mean()
    for ex in ("split", "factored"):
        err = np.abs(r[0][ex] - full)
        assert err.max() <= 1e-4 * np.abs(full).max() + 1e-3 * scale, (ex, err.max())
        assert (r[0][ex][:, -1] == full[:, -1]).all()  # views that saw each gaussian
    assert (full[:, -1] == 2).any() and (full[:, -1] == 0).any()


def test_split_exchange_on_rccl_one_rank():
    """The same overlapped exchange on the real backend (nccl = RCCL), one rank: the collectives degenerate to copies
    but shapes, contiguity and async handles go through the calls of the multi-GPU run (tools/nccl_one_rank.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="1", RANK="0",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "nccl_one_rank.py")], env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0 and "nccl one-rank rehearsal: ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
