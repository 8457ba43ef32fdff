#!/bin/bash
# Experiment builds of the ray kernel (-DR3D_ABLATE=<bits>, wrong results, timing only) next to the product build:
#   256   round 1's gather mapping (the 4 lanes of a sample 16 lanes apart: every lane quad of a load touches 4 lines)
#   64    half the load instructions      128  the second load of a tap re-reads the first one's 16 bytes      65536  every tap in one L1-resident window
# build (CPU, no GPU needed):  bash scripts/gpu_ray_experiments.sh build      run on the GPU box:  bash scripts/gpu_ray_experiments.sh
cd "$(dirname "$0")/.." || exit 1
V="256 64 128 65536 320"
if [ "$1" = build ]; then
  make -s -C real3dportrait_amd/csrc
  for b in $V; do
    ( cd real3dportrait_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function -DR3D_ABLATE=$b -c r3d_render.hip -o ../lib/obj/x_render_$b.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libr3d_hip_ray$b.so ../lib/obj/r3d_api.o ../lib/obj/x_render_$b.o ../lib/obj/r3d_sr.o ../lib/obj/r3d_comm.o ../lib/obj/r3d_sr_f16x3.o -ldl && rm -f ../lib/obj/x_render_$b.o ) &
  done; wait; ls real3dportrait_amd/lib; exit 0
fi
name() { case $1 in 256) echo "round-1 gather mapping (lanes of a sample 16 apart)";; 64) echo "half the load instructions";; 128) echo "second load of a tap re-reads the same 16 B";;
  65536) echo "every tap L1-resident";; 320) echo "round-1 mapping + half the loads";; *) echo "product";; esac; }
for rep in 1 2; do
  python scripts/prof_render.py 128 48 48 60 2>/dev/null | tail -1 | sed "s/^/[product] /"
  for b in $V; do R3D_LIB=$PWD/real3dportrait_amd/lib/libr3d_hip_ray$b.so python scripts/prof_render.py 128 48 48 60 2>/dev/null | tail -1 | sed "s/^/[$(name $b)] /"; done
done
