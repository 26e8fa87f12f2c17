mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r4_final_gputests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r4_final_gputests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4_final_smoke.log 2>&1
echo "smoke rc $?" >> gpurun_out/r4_final_smoke.log
python bench.py > gpurun_out/r4_final_bench.json 2> gpurun_out/r4_final_bench.err
tail -3 gpurun_out/r4_final_gputests.log; tail -2 gpurun_out/r4_final_smoke.log; cut -c1-200 gpurun_out/r4_final_bench.json
