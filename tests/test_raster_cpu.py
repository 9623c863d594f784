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
