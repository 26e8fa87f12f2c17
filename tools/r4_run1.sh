mkdir -p gpurun_out
python -m pytest tests/test_pipeline_gpu.py -x -q -s -k "metric_schedules_n50" > gpurun_out/r4a_n50.log 2>&1
echo "pytest rc $?" >> gpurun_out/r4a_n50.log
for cfg in "8 2" "16 2" "16 1"; do set -- $cfg
  python bench.py --batch $1 --concurrent $2 --steps 2 --warmup 1 --no-parity --no-fast-modes --no-cpu-baseline --no-roofline --no-ref-layout > gpurun_out/r4a_bench_$1x$2.json 2> gpurun_out/r4a_bench_$1x$2.err
done
python bench.py --steps 3 --warmup 1 > gpurun_out/r4a_bench_full.json 2> gpurun_out/r4a_bench_full.err
tail -3 gpurun_out/r4a_n50.log; cat gpurun_out/r4a_bench_*x*.json | cut -c1-200
