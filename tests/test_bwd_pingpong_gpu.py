"""r06 experiment: the backward with two half-batches in flight (gs_render.hip: render_bwd_pp_kernel, switched on by
GSPLAT_BWD_PINGPONG=1 per launch).  The same parity rows as the default kernel's: every test below is the test of
tests/test_fused_gpu.py of the same name, run with the switch set -- small scenes against the oracle (all twelve gradient
arrays), odd sizes, the cuda/render_backward.cu:170 gate, lists of 1 025 .. 10 000 entries, the segmented backward, large
splats, and BASELINE configs[2] at full size."""
import pytest

import test_fused_gpu as F

pytestmark = pytest.mark.gpu


@pytest.fixture
def pingpong(monkeypatch):
    monkeypatch.setenv("GSPLAT_BWD_PINGPONG", "1")


@pytest.mark.parametrize("name", ["tiny", "small"])
def test_fused_matches_oracle(gpu, scene, orc, pingpong, name):
    F.test_fused_matches_oracle(gpu, scene, orc, name)


@pytest.mark.parametrize("l_max", [0, 2])
def test_lower_sh_degrees(gpu, scene, orc, pingpong, l_max):
    F.test_lower_sh_degrees(gpu, scene, orc, l_max)


@pytest.mark.parametrize("shape", [(1, 17, 9, 0), (37, 100, 7, 2), (257, 130, 66, 3)])
def test_odd_sizes_match_oracle(gpu, scene, orc, pingpong, shape):
    F.test_odd_sizes_match_oracle(gpu, scene, orc, shape)


def test_gate_saturation_and_long_lists(gpu, scene, orc, pingpong):
    F.test_saturated_pixels_stop_early(gpu, scene, orc)
    F.test_backward_gate_opaque_gaussians_and_zero_gradient_tiles(gpu, scene, orc)
    F.test_very_long_tile_lists(gpu, scene, orc)
    F.test_large_splats_match_oracle(gpu, scene, orc, 25.0)


def test_long_lists_split_into_segments_for_the_backward(gpu, scene, orc, pingpong):
    F.test_long_lists_split_into_segments_for_the_backward(gpu, scene, orc)


def test_full_size_properties_and_parity(gpu, scene, orc, config3_case, pingpong):
    F.test_full_size_properties_and_parity(gpu, scene, orc, config3_case)
