#!/bin/bash
# usage (GPU box, repo root; binaries built into build/ beforehand, see the header of each tools/native/*.hip):
#   tools/run_probes.sh <tag>  -> gpurun_out/<tag>_probes.txt : the micro-benchmarks behind the design decisions of round 2
out=gpurun_out/${1:-r2}_probes.txt
mkdir -p gpurun_out
cd build
{
echo "== store_bw: per-CU store throughput by lane layout (16 workgroups: the memory system is not the limit) =="; ./store_bw 24576 2560 10 16
echo "== store_bw: the same with every CU writing (1 GB) =="; ./store_bw 196608 2560 10 256 | head -5
echo "== dma_rate: LDS-DMA loads per CU, alone and with stores mixed in (32 workgroups: source in L2; 256: source in the Infinity Cache) =="; ./dma_rate 32; ./dma_rate 256
echo "== valu_rate: issue cost of the softmax instructions =="; ./valu_rate | head -12
echo "== pingpong_rate: MFMA chain beside the partner's VALU block =="; ./pingpong_rate
echo "== pingpong_load: MFMA chain beside the partner's load section =="; ./pingpong_load
echo "== pp_bench: igemm_pp_kernel, 3x3 conv 320->320 at 64x64 x 48 rows and FF-out GEMM (M=196608, N=320, K=1280): full / no global loads / A always L2-hot / no store instructions =="
for b in pp_bench pp_bench1 pp_bench10 pp_bench8; do echo $b; ./$b conv 48 64 320 320 10 | tail -1; ./$b dense 196608 320 1280 10 n | tail -1; ./$b dense 196608 2560 320 10 n | tail -1; done
} > ../$out 2>&1
cd ..
wc -l $out
