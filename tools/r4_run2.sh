mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -s --deselect "tests/test_pipeline_gpu.py::test_full_size_n50_schedules_vs_oracle_fixture[fs_edit_s0]" --deselect "tests/test_pipeline_gpu.py::test_full_size_n50_schedules_vs_oracle_fixture[fs_edit_n20]" > gpurun_out/r4b_gputests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r4b_gputests.log
FREEFINE_HIP_LIB=$PWD/freefine_amd/libfreefine_hip_base.so python tools/bench_x3.py > gpurun_out/r4b_x3_base.txt 2>&1
python tools/bench_x3.py > gpurun_out/r4b_x3_new.txt 2>&1
FREEFINE_HIP_LIB=$PWD/freefine_amd/libfreefine_hip_base.so python tools/bench_x3.py > gpurun_out/r4b_x3_base2.txt 2>&1
for cfg in "16 2" "32 1" "32 2"; do set -- $cfg
  python bench.py --batch $1 --concurrent $2 --steps 2 --warmup 1 --no-parity --no-fast-modes --no-cpu-baseline --no-roofline --no-ref-layout > gpurun_out/r4b_bench_$1x$2.json 2> gpurun_out/r4b_bench_$1x$2.err
done
FREEFINE_HIP_LIB=$PWD/freefine_amd/libfreefine_hip_base.so python bench.py --batch 16 --concurrent 2 --steps 2 --warmup 1 --no-parity --no-fast-modes --no-cpu-baseline --no-roofline --no-ref-layout > gpurun_out/r4b_bench_base_16x2.json 2> gpurun_out/r4b_bench_base_16x2.err
tail -3 gpurun_out/r4b_gputests.log; cat gpurun_out/r4b_bench_*.json | cut -c1-160
