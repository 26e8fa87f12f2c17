#!/bin/bash
# GPU box: the GPU test suite (all failures listed), the TCA kernel test under both attention kernels
mkdir -p gpurun_out
for x in 0 1; do FFN_ATTN_X3W=$x timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -s -k "test_x3_attention_tca or test_attention_tca_production_shapes" 2>&1 | grep "x3 TCA\|TCA\|passed\|failed" | sed "s/^/X3W=$x: /"; done 2>&1 | tee gpurun_out/r6_tca_err.txt
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee gpurun_out/r6_gputests.txt
