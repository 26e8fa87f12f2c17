#!/bin/bash
mkdir -p gpurun_out
export ONE_MODE=x3
ONE_B=24 bash tools/pmc.sh r6_x3w_attn_S4096_24rows_2pass_masked attn 4096 320 5 2 > /dev/null
ONE_B=24 FFN_ATTN_X3W=0 bash tools/pmc.sh r6_x3p_attn_S4096_24rows_2pass_masked attn 4096 320 5 2 > /dev/null
ONE_B=24 bash tools/pmc.sh r6_x3w_attn_S4096_24rows_1pass attn 4096 320 5 1 > /dev/null
ONE_B=24 FFN_ATTN_X3W=0 bash tools/pmc.sh r6_x3p_attn_S4096_24rows_1pass attn 4096 320 5 1 > /dev/null
ONE_B=24 bash tools/pmc.sh r6_x3w_attn_S1024_24rows_2pass_masked attn 1024 640 10 2 > /dev/null
ONE_B=24 FFN_ATTN_X3W=0 bash tools/pmc.sh r6_x3p_attn_S1024_24rows_2pass_masked attn 1024 640 10 2 > /dev/null
for f in gpurun_out/pmc_r6_x3*.txt; do echo "== $f"; cat $f; done
for x in 1 0; do for a in "4096 320 5 2" "4096 320 5 1" "1024 640 10 2" "1024 640 10 1" "256 1280 20 2"; do ONE_B=24 ONE_TIME=50 FFN_ATTN_X3W=$x python tools/one_kernel.py attn $a 2>&1 | grep "us per call" | sed "s/^/X3W=$x /"; done; done | tee gpurun_out/r6_attn_event_times.txt
