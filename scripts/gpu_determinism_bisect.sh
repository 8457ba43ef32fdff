#!/bin/bash
# Round-3 bisect of the "result depends on what else is resident" anomaly.
#   build (CPU):  bash scripts/gpu_determinism_bisect.sh build        run (GPU box):  bash scripts/gpu_determinism_bisect.sh [reps]
# Victims  = ray-kernel experiment builds (-DR3D_ABLATE bits: 4096 quad-shared taps, +131072 lane predicates recomputed per call,
#            +262144 bit selects on VGPR masks instead of v_cndmask on SGPR lane masks, +524288 s_setprio 3).
# Aggressors = the f16x3 SR (default), the SR without MFMAs (libr3d_hip_ablate1), synthetic loads (scripts/probes/aggressor.hip).
cd "$(dirname "$0")/.." || exit 1
L=real3dportrait_amd/lib
VICT="4096 135168 266240 528384"
if [ "$1" = build ]; then
  make -s -C real3dportrait_amd/csrc
  for b in $VICT; do
    ( cd real3dportrait_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function -DR3D_ABLATE=$b -c r3d_render.hip -o ../lib/obj/x_render_$b.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libr3d_hip_ray$b.so ../lib/obj/r3d_api.o ../lib/obj/x_render_$b.o ../lib/obj/r3d_sr.o ../lib/obj/r3d_comm.o ../lib/obj/r3d_sr_f16x3.o -ldl && rm -f ../lib/obj/x_render_$b.o ) &
  done
  mkdir -p scripts/probes/bin
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -shared -o scripts/probes/bin/libaggressor.so scripts/probes/aggressor.hip &
  wait; ls $L scripts/probes/bin; exit 0
fi
REPS=${1:-60}
run() { # lib-suffix, load, extra env
  local lib=$1 load=$2; shift 2
  env "$@" R3D_LIB=$PWD/$L/libr3d_hip_ray$lib.so timeout 300 python scripts/gpu_debug_determinism.py $REPS $load 2>&1 | grep -E "^lib|Error|error" | sed "s/^/[victim $lib | load $load | $*] /"
}
echo "== control: product library"; timeout 300 python scripts/gpu_debug_determinism.py $REPS 1 2>&1 | grep -E "^lib" | sed "s/^/[product] /"
echo "== victim variants next to the f16x3 SR"
for v in $VICT; do run $v 1 X=1; done
echo "== one hardware queue"
run 4096 1 GPU_MAX_HW_QUEUES=1
echo "== victim 4096 next to synthetic loads (mode bits: 1 MFMA, 2 LDS-DMA, 4 ds_read, 8 barrier, 16 76KB LDS, 32 VALU)"
for m in 31 29 30 25 17 1 28 22 48 49 16; do run 4096 agg:$m X=1; done
echo "== victim 4096 next to nothing / gemm / copies"
run 4096 0 X=1; run 4096 gemm X=1; run 4096 copy X=1
