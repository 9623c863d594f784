"""Multi-process (gloo, world_size 2, CPU) test of the view-sharded exchange step: every rank owns one training
view, scatters its compacted per-view gradients into the global-order packed layout and the ranks sum them with
ONE all-reduce (3dgs_amd/dist.py).  The per-view gradients come from the CPU oracle (test infrastructure); what is
under test is the host logic: packed layout, view assignment, the collective, and unpacking."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT, pkg


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _pack_cpu(bwd, mask, l_max):
    """numpy restatement of gsplat_pack_gradients_global (csrc/gs_fused.hip pack_global_kernel)."""
    gdist = pkg("dist")
    cols, width = gdist.packed_layout(l_max)
    N = len(mask)
    packed = np.zeros((N, width), np.float32)
    idx = np.nonzero(mask)[0]
    for name, key in (("xyz", "xyz"), ("rgb", "band0"), ("sh", "sh"), ("opacity", "opacity"), ("scale", "scale"),
                      ("quaternion", "quaternion")):
        a, b = cols[name]
        packed[idx, a:b] = np.asarray(bwd[key], np.float32).reshape(len(idx), b - a)
    a, _ = cols["visible"]
    packed[idx, a] = 1.0
    return packed


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import importlib
    scene = importlib.import_module("3dgs_amd.scene")
    gdist = importlib.import_module("3dgs_amd.dist")
    from oracle import oracle as orc
    r, w, _ = gdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    N, W, H, L = 400, 64, 48, 2
    params = scene.make_gaussians(N, W, H, L)
    params["xyz"][rank::5, 2] *= -1.0  # different cull mask on every rank
    params_all = scene.make_gaussians(N, W, H, L)
    cam = scene.make_camera(W, H, view_index=rank + 1)  # one view per rank
    c = scene.CONFIG
    # every rank must hold identical parameters: use the unmodified set for the check, the per-rank set for culling
    fwd = orc.rasterize(params, cam, c["near_thresh"], c["mh_dist"], c["cull_mask_padding"], c["bg"], L)
    bwd = orc.backward_pass(fwd, cam, scene.make_grad_image(W, H), c["bg"], L)
    packed = torch.from_numpy(_pack_cpu(bwd, fwd["mask"], L))
    local = packed.clone()
    gdist.all_reduce_gradients(packed)
    np.save(os.path.join(out_dir, f"local{rank}.npy"), local.numpy())
    np.save(os.path.join(out_dir, f"reduced{rank}.npy"), packed.numpy())
    # the split exchange's second collective: every rank's [N+1,3] block lands in out[rank]
    blocks = torch.zeros(world, N + 1, 3)
    mine = torch.full((N + 1, 3), float(rank + 1))
    works = [torch.distributed.all_reduce(torch.ones(4), async_op=True), gdist.all_gather_blocks(blocks, mine, async_op=True)]
    for wk in works:
        wk.wait()
    for rr in range(world):
        assert (blocks[rr] == float(rr + 1)).all()
    # the direct exchange (split_direct): all-to-all of the W shards, local sum, all-gather of the reduced shards must
    # give the all-reduce's sums (bit for bit at two ranks: a two-term sum has one order), identical on every rank
    comm = gdist.TorchComm()
    flat = local[:, :12].contiguous().reshape(-1)
    shard = (flat.numel() + world - 1) // world
    buf = torch.zeros(shard * world)
    buf[:flat.numel()] = flat
    want = buf.clone()
    torch.distributed.all_reduce(want)
    got_in, got_sum = torch.zeros(world, shard), torch.zeros(shard)
    comm.all_to_all_blocks(got_in, buf.view(world, shard), async_op=True).wait()
    torch.sum(got_in, dim=0, out=got_sum)
    comm.all_gather_blocks(buf.view(world, shard), got_sum).wait()
    assert torch.equal(buf, want), "direct exchange differs from the all-reduce"
    np.save(os.path.join(out_dir, f"direct{rank}.npy"), buf.numpy())
    # r05: the in-place all-gather (every rank's block IS its slot of the output) on a communicator of its own -- what
    # the split exchange's headline step does; closing that group leaves the default group usable
    own = gdist.TorchComm.own_group()
    assert own.owned and own.world == world and own.rank == rank
    slots = torch.zeros(world, N + 1, 3)
    slots[rank] = float(10 + rank)
    own.all_gather_blocks(slots, slots[rank], async_op=True).wait()
    for rr in range(world):
        assert (slots[rr] == float(10 + rr)).all()
    t = torch.tensor([float(rank)])
    own.all_reduce_max(t)
    assert t.item() == world - 1
    own.barrier()
    own.close()
    assert own.group is None
    torch.distributed.all_reduce(torch.ones(2))  # the default group still works
    # r06: what bench.py reports about the collectives -- the world size as the backend's group sees it, and the split
    # exchange's two collectives timed each alone (MAX over the ranks: the same figures on every rank)
    env = gdist.collective_environment(comm)
    assert env["rccl_world"] == world and env["backend"] == "gloo" and "nccl_algo" in env and "nccl_proto" in env
    tc = gdist.time_collectives(comm, 2000, torch.device("cpu"), reps=2, warmup=1)
    assert tc["all_reduce_common_ms"] > 0 and tc["all_gather_rgb_ms"] > 0
    assert tc["all_reduce_bytes"] == 4 * 2000 * 12 and tc["all_gather_bytes_per_rank"] == 4 * 2001 * 3
    both = torch.tensor([tc["all_reduce_common_ms"], tc["all_gather_rgb_ms"]], dtype=torch.float64)
    lo, hi = both.clone(), both.clone()
    torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
    torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
    assert torch.equal(lo, hi), "the collective times are the MAX over the ranks: identical everywhere"
    un = gdist.unpack(packed, L)
    assert un["sh"].shape == (N, (L + 1) ** 2 - 1, 3) and un["visible"].shape == (N,)
    assert params_all["xyz"].shape == (N, 3)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_view_sharded_all_reduce_gloo(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    local = [np.load(tmp_path / f"local{r}.npy") for r in range(world)]
    reduced = [np.load(tmp_path / f"reduced{r}.npy") for r in range(world)]
    np.testing.assert_allclose(reduced[0], local[0] + local[1], rtol=1e-6, atol=1e-12)
    assert (reduced[0] == reduced[1]).all(), "all ranks must end with identical gradients"
    direct = [np.load(tmp_path / f"direct{r}.npy") for r in range(world)]
    assert (direct[0] == direct[1]).all() and np.abs(direct[0]).max() > 0
    gdist = pkg("dist")
    cols, width = gdist.packed_layout(2)
    vis = reduced[0][:, cols["visible"][0]]
    assert set(np.unique(vis)).issubset({0.0, 1.0, 2.0}) and (vis == 2).any() and (vis < 2).any()
    # culled rows contribute exact zeros
    assert (local[0][local[0][:, cols["visible"][0]] == 0] == 0).all()
    assert width == 12 + 3 * 9


def test_packed_layout_matches_library_width():
    gdist, lib = pkg("dist"), pkg("_lib").load()
    for l in range(4):
        cols, width = gdist.packed_layout(l)
        assert width == lib.gsplat_packed_gradient_width(l)
        assert cols["visible"] == (width - 1, width)


def test_single_process_all_reduce_is_identity():
    gdist = pkg("dist")
    t = torch.arange(12, dtype=torch.float32).reshape(3, 4)
    assert gdist.all_reduce_gradients(t.clone()).equal(t)


def _draw_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import importlib
    gdist = importlib.import_module("3dgs_amd.dist")
    trainer_mod = importlib.import_module("3dgs_amd.trainer")
    gdist.init_from_env(backend="gloo")
    rng = np.random.default_rng(11)  # every rank: the same seed
    mine, everyone = [], []
    for it in range(25):
        draws = trainer_mod.draw_view_indices(rng, it, world, 7)
        everyone.append(draws)
        mine.append(draws[rank])
    got = [torch.zeros(25, dtype=torch.int64) for _ in range(world)]
    torch.distributed.all_gather(got, torch.tensor(mine))
    np.save(os.path.join(out_dir, f"draws{rank}.npy"), np.array(everyone))
    np.save(os.path.join(out_dir, f"picked{rank}.npy"), torch.stack(got).numpy())
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_view_schedule_is_shared_and_sharded(tmp_path):
    """The view-sharded trainer's schedule: every rank draws the SAME W views per iteration from the shared seed and
    trains on its own one; the first two samples are views 0 and 1 as in the reference (cuda/trainer.cu:1438-1444)."""
    world = 2
    mp.spawn(_draw_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    d = [np.load(tmp_path / f"draws{r}.npy") for r in range(world)]
    p = [np.load(tmp_path / f"picked{r}.npy") for r in range(world)]
    assert (d[0] == d[1]).all() and (p[0] == p[1]).all()
    assert d[0].shape == (25, world) and list(d[0][0]) == [0, 1]
    for r in range(world):
        assert (p[0][r] == d[0][:, r]).all()
    assert d[0].min() >= 0 and d[0].max() <= 6 and len(np.unique(d[0])) > 3


def test_thread_ranks_eight_way_collectives_and_schedule():
    """The 8-rank shape of BASELINE config 5 with in-process rank threads (dist.ThreadGroup), on CPU tensors: the
    all-reduce leaves bitwise the same rank-ordered sum on every rank, the block all-gather puts rank r's block in
    out[r], MAX-reduce and barrier work, a failing rank releases the others, and the 8-way view schedule hands every
    rank its own draw of the shared sequence."""
    gdist, trainer_mod = pkg("dist"), pkg("trainer")
    world, N = 8, 257
    grp = gdist.ThreadGroup(world)
    rngs = [np.random.default_rng(100 + r) for r in range(world)]
    locals_ = [torch.from_numpy(rngs[r].normal(size=(N, 12)).astype(np.float32)) for r in range(world)]

    def body(comm):
        assert comm.world == world and comm.backend() == "threads"
        t = locals_[comm.rank].clone()
        comm.all_reduce(t).wait()
        blocks = torch.zeros(world, N + 1, 3)
        comm.all_gather_blocks(blocks, torch.full((N + 1, 3), float(comm.rank + 1)), async_op=True).wait()
        mx = comm.all_reduce_max(torch.tensor([float(comm.rank)], dtype=torch.float64))
        comm.barrier()
        rng = np.random.default_rng(11)
        draws = [trainer_mod.draw_view_indices(rng, it, world, 185) for it in range(10)]
        return t, blocks, float(mx.item()), draws

    out = grp.run(body)
    want = locals_[0].clone()
    for r in range(1, world):
        want += locals_[r]
    for r in range(world):
        t, blocks, mx, draws = out[r]
        assert t.equal(want) and mx == world - 1.0
        for q in range(world):
            assert (blocks[q] == float(q + 1)).all()
        assert draws == out[0][3] and len(draws[0]) == world and draws[0][:2] == [0, 1]
    assert len({tuple(d) for d in out[0][3]}) == 10

    def failing(comm):
        if comm.rank == 5:
            raise ValueError("rank 5 broke")
        comm.all_reduce(torch.zeros(3))

    with pytest.raises(ValueError, match="rank 5 broke"):
        gdist.ThreadGroup(world).run(failing)
    lib = pkg("_lib").load()
    assert lib.gsplat_factored_gradient_width(world) == 12 + 3 * world


def test_exchange_model_bytes():
    """The per-rank bytes of DESIGN section 6 (8 ranks, 1e6 gaussians, SH 3): full 420 MB, factored 252 MB, split 168 MB."""
    gdist = pkg("dist")
    m = {p: gdist.exchange_model(8, 1_000_000, 3, p) for p in ("full", "factored", "split")}
    assert abs(m["full"]["sent_bytes_per_rank"] / 1e6 - 420) < 1
    assert abs(m["factored"]["sent_bytes_per_rank"] / 1e6 - 252) < 1
    assert abs(m["split"]["sent_bytes_per_rank"] / 1e6 - 168) < 1
    assert m["split"]["direct_ms"] < m["factored"]["direct_ms"] < m["full"]["direct_ms"]
    assert m["split"]["ring_ms"] > 5 * m["split"]["direct_ms"]
    four = gdist.exchange_model(8, 1_000_000, 3, "split", chunks=4)
    assert abs(four["exposed_direct_ms_after_backward"] * 4 - four["collectives"][1]["direct_ms"]) < 1e-3
    assert gdist.exchange_model(1, 1000, 3, "split")["sent_bytes_per_rank"] == 0
