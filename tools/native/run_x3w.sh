#!/bin/bash
# GPU box: correctness of attn_x3w_kernel vs fp64 + timing vs attn_x3p_kernel (binaries cross-compiled by tools/native/build_x3w.sh); then the PMC pass
mkdir -p gpurun_out
out=gpurun_out/x3w_run.txt
{
timeout 300 build/native/x3w_test check
for cfg in "16 4096 5 1" "16 4096 5 2" "24 4096 5 2" "16 1024 10 1" "16 1024 10 2" "8 4096 5 2" "8 1024 10 2"; do timeout 120 build/native/x3w_test time $cfg; done
for a in $ABLS; do
  [ -x build/native/x3w_test_abl$a ] || continue
  for cfg in "16 4096 5 1" "16 4096 5 2" "16 1024 10 2"; do timeout 120 build/native/x3w_test_abl$a time $cfg; done
done
} 2>&1 | tee $out
bash tools/native/pmc_x3w.sh 2>&1 | tee gpurun_out/x3w_pmc.txt
