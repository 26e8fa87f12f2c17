#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "x3_every_configuration or vae" 2>&1 | tail -12 | cut -c1-300
timeout 900 python -m pytest tests/test_vae_gpu.py -m gpu -q -x 2>&1 | tail -4 | cut -c1-300
VAE_MODE=x3 timeout 500 python3 tools/vae_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_vae_time.txt
