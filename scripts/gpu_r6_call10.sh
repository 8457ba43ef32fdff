timeout 900 python -m pytest tests/test_gpu_warp_sr.py tests/test_gpu_blend_conv.py -x -q -m gpu 2>&1 | tail -4
