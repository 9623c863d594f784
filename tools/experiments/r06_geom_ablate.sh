#!/bin/bash
# r06: where preprocess_geom_kernel's time goes: diagnostic builds (-DGS_GEOM_ABLATE: 1 no tile loop, 5 no conic / radius math, 6 neither (2: no main stores, 4: no stores and no tile loop, earlier run)
# against the product library, GSPLAT_PRE_SPLIT=1 (the stage = sh_colour_kernel + preprocess_geom_kernel)
cd $GRAFT_REPO_ROOT
export GSPLAT_NO_BUILD=1 GSPLAT_PRE_SPLIT=1
for round in 1 2; do
python tools/time_preprocess.py config3 30 2>&1 | tail -1
for a in 1 5 6; do GSPLAT_LIB=tools/ab/libgeomab$a.so python tools/time_preprocess.py config3 30 2>&1 | tail -1; done
done
