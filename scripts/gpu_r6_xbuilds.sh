# timing-only experiment builds of the Winograd conv (results wrong by construction): conv kernel time per SR frame
for x in "" stamps_skip x_NOPIECE x_NORAW x_NOA x_ALL; do
  [ "$x" = stamps_skip ] && continue
  lib=$PWD/real3dportrait_amd/lib/libr3d_hip.so; [ -n "$x" ] && lib=$PWD/real3dportrait_amd/lib/libr3d_$x.so
  for p in f16mx; do echo "== ${x:-product} $p"; R3D_LIB=$lib R3D_CONV_WINO=1 R3D_SR_PRECISION=$p python scripts/prof_sr.py 20 2>&1 | grep "SR 128"; done
done
echo "== direct"; R3D_CONV_WINO=0 python scripts/prof_sr.py 20 2>&1 | grep "SR 128"
