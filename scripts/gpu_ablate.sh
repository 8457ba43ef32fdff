#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python scripts/prof_sr.py 20 2>/dev/null | grep "SR 128"
for b in 1 2 3; do R3D_DBG=$b R3D_LIB=$PWD/real3dportrait_amd/lib/libr3d_hip_ablate$b.so python scripts/prof_sr.py 20 2>/dev/null | grep "SR 128"; done
