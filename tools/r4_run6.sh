mkdir -p gpurun_out
FFN_BENCH_SHARE_DEVICE=1 timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 1 --warmup 1 --batch 2 --concurrent 1 > gpurun_out/r4f_bench_2rank.json 2> gpurun_out/r4f_bench_2rank.err
echo "2-rank rc $?" >> gpurun_out/r4f_bench_2rank.err
for cfg in "8 3" "8 4" "4 4"; do set -- $cfg
  python bench.py --batch $1 --concurrent $2 --steps 2 --warmup 1 --no-parity --no-fast-modes --no-cpu-baseline --no-roofline --no-ref-layout > gpurun_out/r4f_bench_$1x$2.json 2> gpurun_out/r4f_bench_$1x$2.err
done
cut -c1-220 gpurun_out/r4f_bench_2rank.json; tail -3 gpurun_out/r4f_bench_2rank.err; cat gpurun_out/r4f_bench_*x*.json | cut -c1-160
