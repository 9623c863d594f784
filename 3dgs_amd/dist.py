"""View-sharded data parallelism: one process per GPU, one training view per rank, ONE all-reduce of the
per-gaussian gradients per step.  The reference has no distributed layer; this is the new
exchange step the north star asks for.

Per-view gradients live in compacted (post-cull) order with a different mask on every rank, so each rank first
scatters them into a global-order row-major buffer packed[N, width] (gsplat_pack_gradients_global), zero where
the gaussian was culled, plus a visibility count column; the buffer is summed across ranks with a single
torch.distributed all-reduce (backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests).
"""
import os

import torch
import torch.distributed as dist


def packed_layout(l_max):
    """Column slices of the packed row: name -> (start, stop)."""
    n = (l_max + 1) ** 2
    cols, off = {}, 0
    for name, w in (("xyz", 3), ("rgb", 3), ("sh", 3 * (n - 1)), ("opacity", 1), ("scale", 3), ("quaternion", 4),
                    ("visible", 1)):
        cols[name] = (off, off + w)
        off += w
    return cols, off


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def all_reduce_gradients(packed):
    """Sum the packed per-gaussian gradient rows over all ranks, in place (no-op for a single process)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(packed, op=dist.ReduceOp.SUM)
    return packed


def unpack(packed, l_max):
    """Views into the reduced buffer: dict name -> [N, ...] tensor (global gaussian order)."""
    cols, _ = packed_layout(l_max)
    n = (l_max + 1) ** 2
    out = {k: packed[:, a:b] for k, (a, b) in cols.items()}
    out["sh"] = out["sh"].reshape(packed.shape[0], n - 1, 3)
    out["opacity"] = out["opacity"][:, 0]
    out["visible"] = out["visible"][:, 0]
    return out


class ViewShardedStep:
    """forward + backward of this rank's view, then the gradient all-reduce.  Device tensors in, packed[N,width] out."""

    def __init__(self, params, l_max, width, height, config, bg):
        from . import raster
        self.params, self.l_max, self.config, self.bg = params, l_max, config, bg
        self.N = int(params["xyz"].shape[0])
        self.ctx = raster.RasterContext(self.N, width, height)
        self.width_cols = raster.packed_gradient_width(l_max)
        self.packed = torch.empty(self.N, self.width_cols, dtype=torch.float32, device=params["xyz"].device)
        self.grads = None
        self.world = dist.get_world_size() if dist.is_initialized() else 1

    def step(self, cam, grad_image):
        fwd = self.ctx.rasterize_image(self.params, cam, self.config, self.bg, self.l_max)
        M = fwd["num_culled"]
        if self.grads is None or self.grads["xyz"].shape[0] < M:
            self.grads = self.ctx.alloc_gradients(self.N, self.l_max)  # capacity N: never reallocated again
        self.ctx.backward_pass(self.params, cam, grad_image, self.bg, self.l_max, self.grads)
        if self.world > 1:
            self.ctx.pack_gradients_global(self.grads, self.l_max, self.N, self.packed)
            all_reduce_gradients(self.packed)
        return fwd
