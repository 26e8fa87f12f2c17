#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/membound_x3.py 72 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_membound_x3.txt
timeout 600 python -m pytest tests/test_pipeline_gpu.py -m gpu -q -x -k "from_pretrained_folder or checkpoint_folder" 2>&1 | tail -12 | tee gpurun_out/r6_ckpt_tests.txt
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_unet_gpu.py -m gpu -q 2>&1 | tail -8 | tee gpurun_out/r6_gputests.txt
