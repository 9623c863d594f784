#!/bin/bash
# r05: which layers of the forward's segment blocks run side by side (phase A): tools/ab/libthin{128,512,2048}.so
# (as run: variants built with -DGS_FWD_THIN_LAYER=<blocks>, a macro of gs_render.hip at that commit (89488e6); the threshold
# has since become a per-context option, gsplat_context_set_segment_options / RasterContext.set_segment_options)
cd $GRAFT_REPO_ROOT
export GSPLAT_NO_BUILD=1 GSPLAT_NO_RENDER_DUMPS=1
python tools/make_colmap_dataset.py /tmp/ds --points 1200000 > /tmp/dataset.log 2>&1 && python tools/write_config.py /tmp/garden.yaml > /dev/null 2>&1 || exit 1
for tag in ${TAGS:-thin128 thin512 thin2048}; do
  GSPLAT_LIB=tools/ab/lib$tag.so GSPLAT_DEBUG_STAGES=1 python train.py /tmp/garden.yaml /tmp/ds > /tmp/train_$tag.log 2>&1 || { tail -20 /tmp/train_$tag.log; exit 1; }
  echo "== $tag"
  grep -E "stages|training done" /tmp/train_$tag.log | tail -2 | cut -c1-300
done
