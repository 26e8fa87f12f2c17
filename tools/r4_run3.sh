mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s --deselect "tests/test_pipeline_gpu.py::test_full_size_n50_schedules_vs_oracle_fixture[fs_edit_s0]" --deselect "tests/test_pipeline_gpu.py::test_full_size_n50_schedules_vs_oracle_fixture[fs_edit_n20]" > gpurun_out/r4c_gputests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r4c_gputests.log
FREEFINE_HIP_LIB=$PWD/freefine_amd/libfreefine_hip_base.so python tools/bench_x3.py > gpurun_out/r4c_x3_base.txt 2>&1
python tools/bench_x3.py > gpurun_out/r4c_x3_v2.txt 2>&1
tail -5 gpurun_out/r4c_gputests.log
