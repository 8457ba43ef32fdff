export R=$PWD; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sr_ or synthesis or fusion or conv2d" 2>&1 | tail -5
timeout 300 python scripts/prof_fusion.py 20
timeout 300 python scripts/prof_sr.py 20
