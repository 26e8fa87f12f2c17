mkdir -p gpurun_out
python -m pytest tests/test_pipeline_gpu.py -m gpu -q -s -k "768" > gpurun_out/r4h_gputests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r4h_gputests.log
grep -n "768\|passed\|failed" gpurun_out/r4h_gputests.log | tail -8
