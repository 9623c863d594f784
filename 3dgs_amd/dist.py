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
            # GSPLAT_DIST_BACKEND=gloo lets several ranks rehearse the exchange step on ONE GPU (RCCL needs one
            # device per rank); the default on GPUs is nccl = RCCL over xGMI
            backend = os.environ.get("GSPLAT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


class _Done:
    """Handle of a collective that has already been issued in stream order."""

    def wait(self):
        return True


class TorchComm:
    """The exchange's collectives on a torch.distributed process group (RCCL over xGMI on the GPUs, gloo in the CPU
    tests and the one-GPU rehearsals); a single process is a group of one.  group=None: the default group.
    TorchComm.own_group() makes a group -- a communicator -- of its own over all ranks: bench.py runs its headline on one
    and its optional payload sweep on another, so that an RCCL error in the sweep cannot reach the headline's
    communicator (close() destroys an owned group)."""

    def __init__(self, group=None, owned=False):
        self.on = dist.is_initialized()  # a group of ONE still goes through the backend (tools/nccl_one_rank.py)
        self.group, self.owned = group, bool(owned and group is not None)
        self.world = dist.get_world_size(group) if self.on else 1
        self.rank = dist.get_rank(group) if self.on else 0

    @classmethod
    def own_group(cls):
        """A fresh group over all ranks of the default group (collective: every rank calls it, in the same order)."""
        if not dist.is_initialized():
            return cls()
        return cls(dist.new_group(ranks=list(range(dist.get_world_size())), backend=dist.get_backend()), owned=True)

    def close(self):
        if self.owned and self.group is not None:
            dist.destroy_process_group(self.group)
        self.group, self.owned = None, False

    def all_reduce(self, t, async_op=False):
        if not self.on:
            return _Done()
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op) or _Done()

    def all_gather_blocks(self, out, block, async_op=False):
        """out[world, ...] <- every rank's `block` (which may BE out[rank]: the in-place form).  One collective;
        RCCL/NCCL gathers straight into the contiguous buffer, gloo (CPU tests, single-GPU rehearsal) takes the list form."""
        if not self.on:
            if out[0].data_ptr() != block.data_ptr():
                out[0].copy_(block)
            return _Done()
        if dist.get_backend(self.group) == "nccl":
            return dist.all_gather_into_tensor(out, block, group=self.group, async_op=async_op) or _Done()
        if out[self.rank].data_ptr() == block.data_ptr():
            block = block.clone()  # gloo's list form does not take an input that aliases its output
        return dist.all_gather(list(out.unbind(0)), block, group=self.group, async_op=async_op) or _Done()

    def all_to_all_blocks(self, out, inp, async_op=False):
        """out[r] <- rank r's inp[my rank]: every rank sends block j of `inp` to rank j -- W-1 point-to-point transfers
        per rank, one per xGMI link (the direct half of a reduce-scatter).  RCCL: all_to_all_single; gloo (CPU tests,
        one-GPU rehearsals) has no CUDA all-to-all: pairs of isend / irecv."""
        if not self.on:
            out[0].copy_(inp[0])
            return _Done()
        if dist.get_backend(self.group) == "nccl":
            return dist.all_to_all_single(out, inp, group=self.group, async_op=async_op) or _Done()
        out[self.rank].copy_(inp[self.rank])
        ops = []
        for r in range(self.world):
            if r != self.rank:
                peer = dist.get_global_rank(self.group, r) if self.group is not None else r
                ops.append(dist.P2POp(dist.isend, inp[r], peer, group=self.group))
                ops.append(dist.P2POp(dist.irecv, out[r], peer, group=self.group))
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        return _Done()

    def barrier(self):
        if self.on:
            dist.barrier(group=self.group)

    def all_reduce_max(self, t):
        if self.on:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return t

    def backend(self):
        return dist.get_backend(self.group) if self.world > 1 else "none"


class ThreadGroup:
    """W ranks as W THREADS of one process sharing one GPU: the rehearsal of rank counts that cannot be started as
    processes on a one-GPU box (the pool admits six GPU processes per box; the 8-rank shape of BASELINE config 5 is
    eight).  Every rank-thread owns its RasterContext, its parameter replica and its exchange buffers exactly as a
    rank process does; only the collectives differ: a rendezvous on a threading.Barrier, the reduction done once in
    rank order and copied to every rank (what a ring all-reduce also guarantees: bitwise the same sums everywhere).
    All threads issue on the device's default stream, so GPU order = host issue order and the barriers order it."""

    def __init__(self, world):
        import threading
        self.world = int(world)
        self._barrier = threading.Barrier(self.world)
        self._slots = [None] * self.world
        self._result = None
        self.shared = {}  # for the ranks' own bookkeeping (bench.py: votes on an exchange payload); guard it with barrier()

    def comm(self, rank):
        return ThreadComm(self, rank)

    def run(self, fn):
        """fn(comm) on every rank-thread; returns the list of results, re-raises the first failure (the barrier is
        broken so that the other ranks do not wait for a rank that died)."""
        import threading
        out, err = [None] * self.world, [None] * self.world

        def body(r):
            try:
                out[r] = fn(self.comm(r))
            except BaseException as e:  # noqa: BLE001
                err[r] = e
                self._barrier.abort()

        threads = [threading.Thread(target=body, args=(r,), name=f"rank{r}") for r in range(self.world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        failed = [e for e in err if e is not None]
        if failed:  # the rank that really failed, not the ranks its broken barrier released
            raise ([e for e in failed if not isinstance(e, threading.BrokenBarrierError)] or failed)[0]
        return out


class ThreadComm:
    def __init__(self, group, rank):
        self.group, self.rank, self.world = group, int(rank), group.world

    def all_reduce(self, t, async_op=False):
        g = self.group
        g._slots[self.rank] = t
        g._barrier.wait()                 # every rank's producers are queued on the stream
        if self.rank == 0:
            acc = g._slots[0].clone()
            for r in range(1, self.world):
                acc += g._slots[r]        # rank order: one well-defined sum
            g._result = acc
        g._barrier.wait()
        t.copy_(g._result)
        g._barrier.wait()                 # everyone has queued its copy before the next collective replaces _result
        return _Done()

    def all_gather_blocks(self, out, block, async_op=False):
        g = self.group
        g._slots[self.rank] = block
        g._barrier.wait()
        for r in range(self.world):
            out[r].copy_(g._slots[r])
        g._barrier.wait()
        return _Done()

    def all_to_all_blocks(self, out, inp, async_op=False):
        g = self.group
        g._slots[self.rank] = inp
        g._barrier.wait()
        for r in range(self.world):
            out[r].copy_(g._slots[r][self.rank])
        g._barrier.wait()
        return _Done()

    def barrier(self):
        self.group._barrier.wait()

    def all_reduce_max(self, t):
        g = self.group
        g._slots[self.rank] = t
        g._barrier.wait()
        if self.rank == 0:
            g._result = torch.stack([x.to(g._slots[0].device) for x in g._slots]).amax(0)
        g._barrier.wait()
        t.copy_(g._result)
        g._barrier.wait()
        return t

    def backend(self):
        return "threads"


def all_reduce_gradients(packed):
    """Sum the packed per-gaussian gradient rows over all ranks, in place (no-op for a single process)."""
    TorchComm().all_reduce(packed)
    return packed


def all_gather_blocks(out, block, async_op=False):
    return TorchComm().all_gather_blocks(out, block, async_op=async_op)


XGMI_LINK_GBS = 153.0  # MI355X_MICROARCH.md: 7 xGMI links per GPU, ~153 GB/s each, point to point


def collective_environment(comm=None):
    """What the exchange's collectives run on, for the bench line: the world size AS THE BACKEND'S GROUP SEES IT (not the
    launcher's WORLD_SIZE), the backend, and the RCCL algorithm / protocol overrides in effect (None: RCCL's own choice) --
    so that the first multi-GPU record can be read against exchange_model without a second run."""
    on = dist.is_available() and dist.is_initialized()
    group = getattr(comm, "group", None)
    return {"rccl_world": (dist.get_world_size(group) if on else None),
            "backend": (dist.get_backend(group) if on else None),
            "nccl_algo": os.environ.get("NCCL_ALGO"), "nccl_proto": os.environ.get("NCCL_PROTO"),
            "nccl_min_nchannels": os.environ.get("NCCL_MIN_NCHANNELS"), "rccl_msccl_enable": os.environ.get("RCCL_MSCCL_ENABLE")}


def time_collectives(comm, num_gaussians, device, reps=10, warmup=3):
    """ms of the split exchange's two collectives, each ALONE on otherwise idle links: the SUM all-reduce of
    common[N,12] (48 MB at 1e6 gaussians) and the in-place all-gather of every rank's g_rgb[N+1,3] block (12 MB per
    rank), bracketed by events on the stream the collective synchronises with; MAX over the ranks.  Collective: every rank
    of `comm` calls it.  A group of one measures the backend's own overhead (the collectives are copies)."""
    N, W = int(num_gaussians), comm.world
    common = torch.zeros(N, 12, device=device)
    rgb_all = torch.zeros(W, N + 1, 3, device=device)
    cuda = getattr(device, "type", str(device).split(":")[0]) == "cuda"

    def timed(fn):
        for _ in range(warmup):
            fn()
        if cuda:
            torch.cuda.synchronize()
        comm.barrier()
        if cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
        else:
            import time
            t = time.perf_counter()
            for _ in range(reps):
                fn()
            ms = (time.perf_counter() - t) / reps * 1e3
        t = torch.tensor([ms], dtype=torch.float64, device=device)
        comm.all_reduce_max(t)
        return round(float(t.item()), 4)

    out = {"all_reduce_common_ms": timed(lambda: comm.all_reduce(common).wait()),
           "all_gather_rgb_ms": timed(lambda: comm.all_gather_blocks(rgb_all, rgb_all[comm.rank]).wait()),
           "all_reduce_bytes": 4 * N * 12, "all_gather_bytes_per_rank": 4 * (N + 1) * 3, "reps": reps}
    del common, rgb_all
    return out


def exchange_model(world, num_gaussians, l_max, payload, with_uv_norm=False, chunks=1, link_gbs=XGMI_LINK_GBS):
    """What one step's exchange moves per rank and what that costs on xGMI, for checking a measured run against.
    Two bounds per collective: "ring" (RCCL's ring: every byte of a rank leaves through ONE link) and "direct" (every
    peer over its own link, what the fully connected 8-GPU mesh allows: reduce-scatter + all-gather with 1/W of the
    buffer per peer).  all-reduce of B bytes: a rank sends 2 (W-1)/W B; ring time = that / link, direct = 2 B / W / link.
    all-gather of b bytes per rank: sends (W-1) b; ring = that / link, direct = b / link.  Times in ms; launch and
    synchronisation latencies (tens of microseconds per collective, more with `chunks`) are not in the model."""
    W, N = int(world), int(num_gaussians)
    n = (l_max + 1) ** 2
    tail = 4 * N if with_uv_norm else 0
    if payload == "full":
        coll = [("all_reduce", 4 * N * (12 + 3 * n) + tail)]
    elif payload == "factored":
        coll = [("all_reduce", 4 * (N + 1) * (12 + 3 * W) + tail)]
    elif payload == "split":
        coll = [("all_gather", 4 * (N + 1) * 3), ("all_reduce", 4 * N * 12 + tail)]
    elif payload == "split_direct":
        # the all-reduce spelled out as its two direct halves: all-to-all of the W shards (reduce-scatter), local sum,
        # all-gather of the reduced shards; the same bytes as a ring all-reduce, one peer per link
        shard = (4 * N * 12 + tail + W - 1) // W
        coll = [("all_gather", 4 * (N + 1) * 3), ("all_to_all", shard * W), ("all_gather", shard)]
    else:
        raise ValueError(payload)
    out = dict(world=W, payload=payload, chunks=int(chunks), collectives=[], sent_bytes_per_rank=0, ring_ms=0.0, direct_ms=0.0)
    per_ms = link_gbs * 1e6  # bytes per millisecond on one link
    for kind, b in coll:
        if kind == "all_reduce":
            sent, ring, direct = 2 * (W - 1) / W * b, 2 * (W - 1) / W * b / per_ms, 2 * b / W / per_ms
        elif kind == "all_to_all":  # b = the whole buffer: a rank keeps 1/W of it and sends one shard to each peer
            sent, ring, direct = (W - 1) / W * b, (W - 1) / W * b / per_ms, (b / W / per_ms if W > 1 else 0.0)
        else:
            sent, ring, direct = (W - 1) * b, (W - 1) * b / per_ms, (b / per_ms if W > 1 else 0.0)
        out["collectives"].append(dict(kind=kind, buffer_bytes=int(b), sent_bytes_per_rank=int(sent), ring_ms=round(ring, 4),
                                       direct_ms=round(direct, 4)))
        out["sent_bytes_per_rank"] += int(sent)
    # the split payload's two collectives run concurrently (different buffers, the gather starts first)
    if payload == "split_direct":
        out["ring_ms"] = round(sum(c["ring_ms"] for c in out["collectives"]), 4)
        out["direct_ms"] = round(sum(c["direct_ms"] for c in out["collectives"]), 4)
        out["exposed_direct_ms_after_backward"] = round(sum(c["direct_ms"] for c in out["collectives"][1:]), 4)
        out["exposed_ring_ms_after_backward"] = round(sum(c["ring_ms"] for c in out["collectives"][1:]), 4)
    elif payload == "split":
        out["ring_ms"] = round(sum(c["ring_ms"] for c in out["collectives"]), 4)       # same links: they add up
        out["direct_ms"] = round(sum(c["direct_ms"] for c in out["collectives"]), 4)
        # with `chunks` ranges only the last range's share of the all-reduce is behind the backward
        ar = out["collectives"][1]
        out["exposed_direct_ms_after_backward"] = round(ar["direct_ms"] / max(1, int(chunks)), 4)
        out["exposed_ring_ms_after_backward"] = round(ar["ring_ms"] / max(1, int(chunks)), 4)
    else:
        out["ring_ms"], out["direct_ms"] = out["collectives"][0]["ring_ms"], out["collectives"][0]["direct_ms"]
    return out


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
    """forward + backward of this rank's view, then the gradient exchange.  Device tensors in; out: the summed
    per-gaussian gradients of the step's W views on every rank.

    exchange="split" (default): every SH-coefficient gradient of a view is the outer product g_rgb[3] x Y_k(view
    direction), so a rank ships only g_rgb.  Two concurrent collectives: the 12 direction-independent columns are SUM
    all-reduced (common[N,12] = xyz 3, opacity, scale 3, quaternion 4, views that saw the gaussian) and every rank's
    g_rgb[N,3] + camera position is all-gathered in place (rgb_all[world, N+1, 3]); at 8 ranks a rank moves 168 MB per
    step instead of 420 MB.  r05: the step's OUTPUT is that factored form -- `common` and `rgb_all` -- which
    AdamOptimizer.step_split consumes directly (gsplat_optimizer_step_sh_views rebuilds sum_r g_rgb^r x Y_k(dir^r) inside the
    colour groups' Adam); no compacted gradient array, no pack pass and no packed[N, 12 + 3 n] rows are written: the
    per-gaussian backward writes its twelve columns straight into common at the gaussians' global indices
    (gsplat_backward_gaussians_split), the compositing backward's g_rgb goes straight into this rank's block of rgb_all.
    `packed` materialises the rows on demand (tests, comparisons).
    exchange="split_packed": the same collectives, and the packed rows materialised in every step (r04's default).
    exchange="split_direct": the split payload with the all-reduce spelled out as its two direct halves (below).
    exchange="factored": ONE all-reduce whose buffer carries, per gaussian, the 12 non-SH gradient columns plus
    one 3-float g_rgb slot per rank (12 + 3*world floats instead of 12 + 3*n_coeffs = 60 at SH degree 3), and one
    extra row with every rank's camera position; the SH-coefficient gradients are rebuilt locally afterwards
    (gsplat_unpack_gradients_factored).  Still ONE all-reduce per step.
    exchange="full": all-reduce the complete packed[N, 12+3*n_coeffs] rows (the north star's single all-reduce).

    with_uv_norm=True (training): the all-reduced buffer carries N more floats, the per-view |grad_uv| in global
    order, so that after the exchange `uv_norm_sum[N]` holds the sum over the step's views: what the densification
    statistics of cuda/trainer.cu:1136-1157 need on a view-sharded step.  Same collective, no extra launch on the wire.
    """

    def __init__(self, params, l_max, width, height, config, bg, exchange="split", with_uv_norm=False, ctx=None,
                 comm=None, chunks=None, exchange_at_world_one=False):
        from . import raster
        self.raster = raster
        self.params, self.l_max, self.config, self.bg = params, l_max, config, bg
        self.N = N = int(params["xyz"].shape[0])
        # A context this step creates is lean (only the fused backward follows: Sigma / J / conic / colour are not
        # materialised).  A context the CALLER owns is left as it is: its later forwards may feed the stand-alone
        # backward operators, which read those arrays -- the caller switches it itself (Trainer and bench.py do).
        if ctx is None:
            ctx = raster.RasterContext(N, width, height)
            ctx.set_lean_forward(True)
        self.ctx = ctx
        self.width_cols = wc = raster.packed_gradient_width(l_max)
        self.dev = dev = params["xyz"].device
        self.comm = comm if comm is not None else TorchComm()  # ThreadComm: in-process ranks (ThreadGroup)
        self.world, self.rank = self.comm.world, self.comm.rank
        if exchange not in ("split", "split_packed", "split_direct", "factored", "full"):
            raise ValueError(f"unknown exchange {exchange!r}")
        # "split_direct": the split payload with the all-reduce of the twelve common columns spelled out as its two
        # direct halves -- an all-to-all of the W shards (every peer over its own xGMI link), a local sum, an all-gather of
        # the reduced shards -- instead of leaving the algorithm to RCCL (a single-link ring is per-link bound).
        self.direct = exchange == "split_direct"
        self.materialize = exchange == "split_packed"  # unpack into packed[N, 12 + 3 n] in every step
        self.exchange = exchange = "split" if exchange.startswith("split") else exchange
        # a group of ONE normally skips the exchange; exchange_at_world_one runs it anyway (the collectives degenerate to
        # copies through the real backend): what tools/nccl_one_rank.py uses to rehearse and to time the host's share
        self.always_exchange = bool(exchange_at_world_one)
        self.host_s = 0.0  # host seconds spent in the exchange's own Python + collective calls (exchange_gradients etc.)
        # chunks > 1 (split exchange, more than one rank): the per-gaussian backward runs in that many ranges of global
        # indices and the all-reduce of one range's twelve common columns is in flight while the next range is computed
        # (GSPLAT_EXCHANGE_CHUNKS; default 1: one all-reduce behind the whole backward).  Same rows either way.
        self.chunks = max(1, int(chunks if chunks is not None else os.environ.get("GSPLAT_EXCHANGE_CHUNKS", "1")))
        if self.direct or exchange != "split":
            self.chunks = 1  # the shards of the direct exchange cut across the ranges: one exchange behind the whole backward
        self.with_uv_norm = bool(with_uv_norm)
        tail = N if with_uv_norm else 0
        z = lambda n: torch.zeros(n, dtype=torch.float32, device=dev)
        self.uv_norm_sum = None
        self._packed = None       # split: materialised on demand (property `packed`)
        self._packed_fresh = False
        self._grads = None        # compacted gradient arrays: only the packed payloads and the no-exchange step need them
        # the buffer that goes through the all-reduce is flat: rows first, then (training) the N uv norms
        if exchange == "full":
            self._reduce_buf = z(N * wc + tail)
            self._packed = self._reduce_buf[:N * wc].view(N, wc)
        elif exchange == "factored":
            self._packed = torch.empty(N, wc, dtype=torch.float32, device=dev)
        self.fw = raster.factored_gradient_width(self.world)
        self.factored = None
        if exchange == "factored":
            self._reduce_buf = z((N + 1) * self.fw + tail)
            self.factored = self._reduce_buf[:(N + 1) * self.fw].view(N + 1, self.fw)
        self._rgb_gather = None
        if exchange == "split":
            W = self.world
            shard = (N * 12 + tail + W - 1) // W  # split_direct: the buffer is cut into W equal shards
            self._reduce_buf = z(shard * W if self.direct else N * 12 + tail)
            if self.direct:
                self._shards_in = z(shard * W).view(W, shard)   # what the peers sent: their copies of this rank's shard
                self._shard_sum = z(shard)
            self.common = self._reduce_buf[:N * 12].view(N, 12)
            self.rgb_all = z(self.world * (N + 1) * 3).view(self.world, N + 1, 3)
            self.rgb = self.rgb_all[self.rank]                 # this rank's block, gathered IN PLACE; row N: campos
        if with_uv_norm:
            body = N * 12 if exchange == "split" else self._reduce_buf.numel() - N  # (split_direct pads behind the tail)
            self.uv_norm_sum = self._reduce_buf[body:body + N]
        self._campos_of = None  # the camera whose position is already in the exchange buffers
        self._blind = False     # this rank's view saw no gaussian in the current step
        self._chunk_reduces = None  # chunked step: the all-reduces of the common rows started during the backward

    # ---- buffers that only some payloads need
    @property
    def grads(self):
        """Compacted per-view gradient arrays (capacity N: never reallocated): what the `full` / `factored` payloads pack
        from and what a step without an exchange (one rank) writes.  The split payload never touches them."""
        if self._grads is None:
            g = self.ctx.alloc_gradients(self.N, self.l_max)
            g["precompute_rgb"] = torch.empty(self.N, 3, dtype=torch.float32, device=self.dev)
            if self.with_uv_norm:
                g["uv"] = torch.empty(self.N, 2, dtype=torch.float32, device=self.dev)
            self._grads = g
        return self._grads

    @property
    def packed(self):
        """packed[N, 12 + 3 n]: the summed rows in the packed layout (packed_layout).  `full` / `factored` produce them
        in every step; the split payload rebuilds them here, on demand, from common + rgb_all
        (gsplat_unpack_gradients_split) -- the optimizer does not need them (AdamOptimizer.step_split)."""
        if self.exchange == "split" and not self._packed_fresh:
            self.materialize_packed()
        return self._packed

    def materialize_packed(self):
        N = self.N
        if self._packed is None:
            self._packed = torch.empty(N, self.width_cols, dtype=torch.float32, device=self.dev)
        self.raster.unpack_gradients_split(self.params["xyz"], self.common, self.rgb_all, 3 * (N + 1), self.l_max, N,
                                           self.world, self._packed)
        self._packed_fresh = True
        return self._packed

    def describe_exchange(self):
        mb = lambda t: f"{t.numel() * 4 / 1e6:.0f} MB"
        if self.exchange == "split" and self.direct:
            return (f"split_direct: all-to-all of {mb(self._reduce_buf)} in {self.world} shards + local sum + all-gather of "
                    f"{mb(self._shard_sum)} per rank (the all-reduce's two direct halves) + all-gather of {mb(self.rgb)} per rank")
        if self.exchange == "split":
            how = f" in {self.chunks} ranges behind the ranges of the per-gaussian backward" if self.chunks > 1 else ""
            out = ("; packed[N, 12 + 3 n] rows materialised every step" if self.materialize else
                   "; output = common[N,12] + rgb_all[W,N+1,3] (what gsplat_optimizer_step_sh_views consumes), written in "
                   "global order by the backward kernels themselves: no pack / unpack pass")
            return (f"split{'_packed' if self.materialize else ''}: all-reduce of {mb(self._reduce_buf)}{how} + in-place "
                    f"all-gather of {mb(self.rgb)} per rank{out}")
        return f"{self.exchange}: one all-reduce of {mb(self._reduce_buf)}"

    def exchange_gradients(self, cam):
        """Sum this step's gradients over the ranks.  split: leaves common (and uv_norm_sum) reduced and rgb_all gathered;
        full / factored: leaves the packed rows."""
        N = self.N
        self._packed_fresh = False
        if self.exchange == "split":
            gather = self._rgb_gather  # started by step() behind the compositing backward, or None (blind rank)
            self._rgb_gather = None
            if self._blind:  # nothing in view on this rank: exact zeros, but every collective is still joined
                self._reduce_buf.zero_()
                self.rgb[:N].zero_()
                self._set_campos(cam)
                gather = self.comm.all_gather_blocks(self.rgb_all, self.rgb, async_op=True)
                if self.chunks > 1:  # the other ranks reduce range by range: join every one of those collectives
                    self._chunk_reduces = [self.comm.all_reduce(self.common[lo:hi], async_op=True)
                                           for lo, hi in self.chunk_bounds() if hi > lo]
            if self._chunk_reduces is not None:
                # chunked step: the common rows are already on their way range by range; what is left is the tail
                # (the |grad_uv| norms, which need every range's uv gradient)
                reduces = self._chunk_reduces
                self._chunk_reduces = None
                if self.with_uv_norm:
                    reduces.append(self.comm.all_reduce(self.uv_norm_sum, async_op=True))
            elif self.direct:
                reduces = [self.comm.all_to_all_blocks(self._shards_in, self._reduce_buf.view(self.world, -1), async_op=True)]
            else:
                reduces = [self.comm.all_reduce(self._reduce_buf, async_op=True)]
            gather.wait()
            for r in reduces:
                r.wait()
            if self.direct:
                # second half: this rank owns the sum of its shard (formed in rank order: one rank computes each element,
                # so every replica receives the same bits), and gathers everybody's
                torch.sum(self._shards_in, dim=0, out=self._shard_sum)
                self.comm.all_gather_blocks(self._reduce_buf.view(self.world, -1), self._shard_sum)
            if self.materialize:
                self.materialize_packed()
            return None
        if self._blind:
            self._reduce_buf.zero_()
        elif self.with_uv_norm:
            self.raster.pack_uv_grad_norm(self.ctx, self.grads, N, self.uv_norm_sum)
        if self.exchange == "factored":
            f = self.factored
            if not self._blind:
                self.raster.pack_gradients_factored(self.ctx, self.grads, N, self.rank, self.world, f)
            f[N].zero_()
            f[N, 12 + 3 * self.rank: 15 + 3 * self.rank] = self._campos_tensor(cam)
            self.comm.all_reduce(self._reduce_buf)
            self.raster.unpack_gradients_factored(self.params["xyz"], f[N, 12:], f, self.l_max, N,
                                                  self.world, self._packed)
        else:
            if not self._blind:
                self.ctx.pack_gradients_global(self.grads, self.l_max, N, self._packed)
            self.comm.all_reduce(self._reduce_buf)
        return self._packed

    def chunk_bounds(self):
        """Global-index ranges of the chunked exchange: the same on every rank (they depend on N only)."""
        N, K = self.N, self.chunks
        return [(N * k // K, N * (k + 1) // K) for k in range(K)]

    def backward_gaussians_chunked(self, cam):
        """Per-gaussian backward range by range; each range writes its rows of `common` in place and their all-reduce
        is started behind it, so that only the last range's exchange is exposed.  Every rank issues the same collectives
        in the same order (the rows of gaussians a rank culled were zeroed by the compositing backward's pass)."""
        self._chunk_reduces = []
        for lo, hi in self.chunk_bounds():
            if hi == lo:
                continue
            self.ctx.backward_gaussians_split(self.params, cam, self.l_max, self.common, self.uv_norm_sum, lo, hi)
            self._chunk_reduces.append(self.comm.all_reduce(self.common[lo:hi], async_op=True))

    def _campos_tensor(self, cam):
        dev_pos = cam.get("campos_dev")  # raster.device_camera uploads it once per view
        if dev_pos is not None:
            return dev_pos
        return torch.as_tensor([float(c) for c in cam["campos"]], dtype=torch.float32, device=self.dev)

    def _set_campos(self, cam):
        if self._campos_of is cam:  # same view as the last step (the benchmark): already in place
            return
        self.rgb[self.N] = self._campos_tensor(cam)
        self._campos_of = cam

    def step(self, cam, grad_image=None, grad_fn=None, bg=None):
        """One view-sharded step.  grad_image: dL/dimage [H,W,3]; or grad_fn(fwd) -> dL/dimage computed from this
        rank's rendering (the training loop's loss).  Returns the forward dict (None when nothing was in view)."""
        import time
        bg = self.bg if bg is None else bg
        exchanging = self.world > 1 or self.always_exchange
        split = exchanging and self.exchange == "split"
        if split:
            self._set_campos(cam)  # device-to-device (or cached): issued before the GPU has work queued
        self._blind = False
        try:
            fwd = self.ctx.rasterize_image(self.params, cam, self.config, bg, self.l_max)
        except Exception as e:  # no gaussian in view: the reference warns and skips the view (cuda/trainer.cu:1352)
            if getattr(e, "code", None) != -5:
                raise
            self._blind, fwd = True, None
        if not self._blind:
            if grad_fn is not None:
                grad_image = grad_fn(fwd)
            if split:
                # g_rgb is final after the compositing backward: it lands in this rank's block of rgb_all and its
                # all-gather runs behind the per-gaussian backward; the same pass clears the common rows of culled gaussians
                self.ctx.backward_render(grad_image, bg, self.rgb, self.common, self.uv_norm_sum)
                t0 = time.perf_counter()
                self._rgb_gather = self.comm.all_gather_blocks(self.rgb_all, self.rgb, async_op=True)
                self.host_s += time.perf_counter() - t0
                if self.chunks > 1:
                    self.backward_gaussians_chunked(cam)
                else:
                    self.ctx.backward_gaussians_split(self.params, cam, self.l_max, self.common, self.uv_norm_sum)
            else:
                self.ctx.backward_pass(self.params, cam, grad_image, bg, self.l_max, self.grads)
        if exchanging:
            t0 = time.perf_counter()
            self.exchange_gradients(cam)
            self.host_s += time.perf_counter() - t0
        return fwd
