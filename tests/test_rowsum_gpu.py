"""The row-level nine-value reductions of the backward (gs::row_sum9 and gs::row_moments9, masked DPP in inline
assembly) checked
lane by lane on the GPU: tools/dpp_rowsum_test.hip compares every row total with a host sum."""
import os
import subprocess

import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "tools", "dpp_rowsum_test.hip")
EXE = os.path.join(ROOT, "tools", "dpp_rowsum_test")


def _build():
    hdr = os.path.join(ROOT, "3dgs_amd", "csrc", "gs_render.h")
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(SRC), os.path.getmtime(hdr)):
        subprocess.check_call(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-value", SRC, "-o", EXE])
    return EXE


def test_rowsum_program_builds():
    assert os.path.exists(_build())


@pytest.mark.gpu
def test_row_sum9_on_gpu():
    out = subprocess.run([_build()], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "row_sum9: ok" in out.stdout and "row_moments9: ok" in out.stdout \
        and "row_moments9q: ok" in out.stdout and "row_moments9r: ok" in out.stdout, out.stdout + out.stderr


BH_SRC = os.path.join(ROOT, "tools", "block_hits_test.hip")
BH_EXE = os.path.join(ROOT, "tools", "block_hits_test")


def _build_block_hits():
    hdr = os.path.join(ROOT, "3dgs_amd", "csrc", "gs_render.h")
    if not os.path.exists(BH_EXE) or os.path.getmtime(BH_EXE) < max(os.path.getmtime(BH_SRC), os.path.getmtime(hdr)):
        subprocess.check_call(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-value", BH_SRC, "-o", BH_EXE])
    return BH_EXE


def test_block_hits_program_builds():
    assert os.path.exists(_build_block_hits())


@pytest.mark.gpu
def test_block_hits_is_conservative_on_gpu():
    """gs::block_hits (the ellipse-vs-block test of the compositing kernels) never drops a block that holds a pixel with
    alpha >= 1/255: 262144 random gaussians against a per-pixel evaluation."""
    out = subprocess.run([_build_block_hits()], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "block_hits: ok" in out.stdout, out.stdout + out.stderr
