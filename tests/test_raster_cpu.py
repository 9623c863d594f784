"""Host-side behaviour of the forward's result object (3dgs_amd/raster.py) that needs no GPU."""
import importlib

import pytest


def test_forward_result_builds_views_on_demand(monkeypatch):
    raster = importlib.import_module("3dgs_amd.raster")
    built = []

    def fake_view(ptr, shape, typestr, owner):
        built.append((ptr, shape))
        return ("view", ptr, shape)

    monkeypatch.setattr(raster, "_view", fake_view)
    out = raster._LazyViews(dict(num_culled=3, num_splats=7),
                            dict(image=(0x1000, (2, 2, 3), "<f4"), sorted=(0x2000, (7,), "<i4"), rgb=(None, (3, 3), "<f4")),
                            owner=None)
    assert out["num_culled"] == 3 and built == []                      # counts are plain entries, nothing built yet
    assert "image" in out and "sorted" in out and "nope" not in out and built == []
    assert out["image"] == ("view", 0x1000, (2, 2, 3)) and built == [(0x1000, (2, 2, 3))]
    assert out["image"] is out["image"] and len(built) == 1              # built once
    assert out.get("nope", 5) == 5 and out.get("sorted")[1] == 0x2000 and len(built) == 2
    with pytest.raises(KeyError):
        out["nope"]
    assert set(out.keys()) == {"num_culled", "num_splats", "image", "sorted", "rgb"} and len(built) == 3
    assert dict(out)["rgb"] == ("view", None, (3, 3)) and len(out) == 5


def test_chunk_ownership_arithmetic():
    import numpy as np
    """gs_common.h (r03): the cull's 256 slices are runs of whole 64-entry chunks, slice s = chunks
    [C*s/256, C*(s+1)/256); bin_slice_of_chunk(C, c) = ((c+1)*256 - 1) / C must invert that for every chunk, for chunk
    counts below, at and far above the slice count (restated here; the device code uses the same integer formulas)."""
    K = 256
    first = lambda C, s: C * s // K
    slice_of = lambda C, c: ((c + 1) * K - 1) // C
    for C in (1, 2, 3, 255, 256, 257, 511, 1000, 15625, 19688, 62500, 1 << 20):
        owners = np.full(C, -1)
        for s in range(K):
            lo, hi = first(C, s), first(C, s + 1)
            assert 0 <= lo <= hi <= C
            owners[lo:hi] = s
        assert (owners >= 0).all() and first(C, K) == C          # every chunk in exactly one slice, in order
        got = np.array([slice_of(C, c) for c in range(0, C, max(1, C // 4096))])
        assert (got == owners[::max(1, C // 4096)]).all()
        assert slice_of(C, C - 1) == K - 1 and slice_of(C, 0) == owners[0]
    # round-robin deal: chunk c -> workgroup c % 256, wave (c / 256) % 16: every chunk exactly once
    C = 15625
    seen = np.zeros(C, int)
    for b in range(K):
        for w in range(16):
            seen[b + K * w::K * 16] += 1
    assert (seen == 1).all()


def test_segment_slot_arithmetic():
    """gs_render.h (r05): the checkpoint of boundary k >= 1 of a tile whose list starts at instance r lives in slot
    r // S + k, S = 496 -- no table.  Restated here: over random tile lists no two boundaries of lists beyond the split
    threshold share a slot and every slot is below instances // S + 2, the pool's size.  And the forward's block table:
    with the long lists ranked by their number of segments (most first), segment (t, k) is block base[k] + rank[t] with
    base[k] = L_0 + .. + L_(k-1), L_k = lists of more than k segments -- a bijection onto [0, sum of the segments), every
    block of a list behind the list's earlier ones."""
    import numpy as np
    S, split_min = 496, 1488
    rng = np.random.default_rng(5)
    for trial in range(40):
        T = int(rng.integers(1, 400))
        lens = rng.choice([0, 1, 37, 495, 496, 497, 1487, 1488, 1489, 1984, 1985, 5000, 9800, 20000], size=T) + \
            rng.integers(0, 3, size=T)
        starts = np.concatenate([[0], np.cumsum(lens)])
        total = int(starts[-1])
        slots = []
        for t in range(T):
            if lens[t] > split_min:
                m = -(-int(lens[t]) // S)
                slots += [int(starts[t]) // S + k for k in range(1, m)]   # boundaries 1 .. m - 1
        assert len(slots) == len(set(slots))
        assert not slots or max(slots) < total // S + 2
        # the forward's table
        m = np.where(lens > split_min, -(-lens // S), 0)
        order = np.argsort(-m, kind="stable")
        rank = np.empty(T, int)
        rank[order] = np.arange(T)
        L = np.array([(m > k).sum() for k in range(int(m.max()) + 1)])
        base = np.concatenate([[0], np.cumsum(L)])
        blocks = {}
        for t in range(T):
            for k in range(int(m[t])):
                assert rank[t] < L[k]
                b = int(base[k] + rank[t])
                assert b not in blocks
                blocks[b] = (t, k)
                assert k == 0 or base[k - 1] + rank[t] < b      # the segment in front has the smaller block index
        assert sorted(blocks) == list(range(int(m.sum())))
