#!/bin/bash
# r03 slope experiment: is render_bwd bound by VALU issue?  Library variants prebuilt by
#   for k in 4 8 16: tools/ab/build_variant.sh fma$k 3dgs_amd/csrc/gs_render.hip -DGS_EXTRA_FMA=$k   (k extra plain v_fma_f32 per trip)
#   tools/ab/build_variant.sh abl2 3dgs_amd/csrc/gs_render.hip -DGS_ABLATE=2                        (row_moments9 + atomic removed)
# run alternately on ONE box (tools/ab/run.sh: two rounds, 300 steps); fit  time = a + b * (VALU per trip).
bash tools/ab/run.sh 300 base fma4 fma8 fma16 abl2
