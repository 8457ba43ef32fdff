#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for p in f16x3 f16mx; do
  R3D_SR_PRECISION=$p python scripts/prof_sr.py 30 2>/dev/null | grep "SR 128" | sed "s/^/$p full: /"
  for b in ${R3D_ABLATE_SET:-1 2 4 5}; do R3D_SR_PRECISION=$p R3D_DBG=$b R3D_LIB=$PWD/real3dportrait_amd/lib/libr3d_hip_ablate$b.so python scripts/prof_sr.py 30 2>/dev/null | grep "SR 128" | sed "s/^/$p ablate$b: /"; done
done
