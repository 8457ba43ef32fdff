# SR-only conv timing of experiment builds real3dportrait_amd/lib/libr3d_x_*.so next to the product (f16mx, Winograd on), and the direct kernel
for l in real3dportrait_amd/lib/libr3d_hip.so real3dportrait_amd/lib/libr3d_x_*.so; do echo "== $l"; R3D_LIB=$PWD/$l R3D_CONV_WINO=1 python scripts/prof_sr.py 20 2>&1 | grep "SR 128"; done
echo "== direct"; R3D_CONV_WINO=0 python scripts/prof_sr.py 20 2>&1 | grep "SR 128"
