# round 5, call 15: the torso forward without head_torso_block's tail fold (R3D_HB_TAIL_FOLD A/B): Warp goldens + torso frame time
for tf in 0 1; do
  echo "== R3D_HB_TAIL_FOLD=$tf"
  R3D_HB_TAIL_FOLD=$tf timeout 600 python -m pytest tests/test_gpu_warp_sr.py tests/test_gpu_f16x3.py -m gpu -q -k "warp or torso" 2>&1 | tail -1
  for i in 1 2; do R3D_HB_TAIL_FOLD=$tf timeout 300 python scripts/torso_frames.py 30 2>&1 | tail -1; done
done
