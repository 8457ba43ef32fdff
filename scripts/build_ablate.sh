#!/bin/bash
# experiment builds of the conv kernel: libr3d_hip_ablate{1,2,3}.so with -DR3D_ABLATE=<bits> (bit 0: no MFMAs in conv3x3_dma_block, bit 1: no
# epilogue); select with R3D_LIB.  Only r3d_sr_f16x3.hip differs; the other objects are the product's.
cd "$(dirname "$0")/../real3dportrait_amd/csrc" && make -s && for b in ${R3D_ABLATE_SET:-1 2 3}; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function -DR3D_ABLATE=$b -c r3d_sr_f16x3.hip -o ../lib/obj/ablate$b.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libr3d_hip_ablate$b.so ../lib/obj/r3d_api.o ../lib/obj/r3d_render.o ../lib/obj/r3d_sr.o ../lib/obj/r3d_comm.o ../lib/obj/ablate$b.o -ldl ) & done; wait; ls ../lib
