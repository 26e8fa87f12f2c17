#!/bin/bash
# usage: [NAME=x3w_test_abl1] tools/native/build_x3w.sh [extra hipcc flags]   -> build/native/$NAME (+ the .s and resource-usage remarks under build/native/$NAME.d/)
# attn_x3w_kernel needs -mllvm -amdgpu-mfma-vgpr-form (VGPR-destination MFMAs at one wave per SIMD) and -fno-slp-vectorize (no v_pk_*_f32 beside MFMAs)
R=$(cd "$(dirname "$0")/../.." && pwd)
NAME=${NAME:-x3w_test}
mkdir -p $R/build/native/$NAME.d && cd $R/build/native/$NAME.d
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize -save-temps -Rpass-analysis=kernel-resource-usage "$@" -o $R/build/native/$NAME $R/tools/native/x3w_test.hip 2> remarks.txt
rc=$?
grep -A9 "Function Name: _Z15attn_x3w" remarks.txt | grep "Function Name\|VGPRs\|AGPRs\|Scratch\|Spill" | sed 's/.*attention_x3w.h:[0-9]*:0: *//' | tr '\n' ' '; echo
grep -B2 -A6 "error" remarks.txt | head -40
exit $rc
