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
