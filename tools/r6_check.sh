#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "kv64" 2>&1 | tail -30 | cut -c1-400 | tee gpurun_out/r6_kv64_tests.txt
