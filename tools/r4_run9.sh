mkdir -p gpurun_out
python tools/bench_other.py > gpurun_out/r4_other_entry_points.txt 2>&1
python bench.py --start-step 35 --steps 3 --warmup 1 --no-parity --no-fast-modes --no-cpu-baseline --no-roofline > gpurun_out/r4_bench_geobench2d_schedule.json 2> gpurun_out/r4_bench_geobench2d_schedule.err
python bench.py --start-step 15 --steps 2 --warmup 1 --no-parity --no-fast-modes --no-cpu-baseline --no-roofline --no-ref-layout > gpurun_out/r4_bench_geobench3d_schedule.json 2> gpurun_out/r4_bench_geobench3d_schedule.err
cat gpurun_out/r4_other_entry_points.txt | grep -v amdgpu; cut -c1-200 gpurun_out/r4_bench_geobench2d_schedule.json; cut -c1-200 gpurun_out/r4_bench_geobench3d_schedule.json
