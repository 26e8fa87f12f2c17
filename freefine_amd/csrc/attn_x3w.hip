// Second translation unit of libfreefine_hip.so: attn_x3w_kernel (attention_x3w.h) runs ONE wave per SIMD on the whole 512-register file, and at that
// occupancy hipcc's default code generation works against it:
//   -mllvm -amdgpu-mfma-vgpr-form   MFMA results in the architectural VGPRs.  By default a kernel that may use more than 256 registers gets the
//                                   AGPR-destination MFMA forms, and every softmax instruction on the scores then goes through v_accvgpr_read / _write
//                                   copies (measured on the first version: 190 copies per 64 keys, and spills);
//   -fno-slp-vectorize              no v_pk_add_f32 / v_pk_mul_f32 in the gaps between MFMAs (a packed f32 instruction costs more issue time there
//                                   than the two scalar ones it replaces: MI355X_MICROARCH.md, price of one filler beside MFMAs).
// Built by __graft_entry__.build() into its own object and linked with capi.o.  No exported symbol: fx3w_launch has hidden visibility.
#include <hip/hip_runtime.h>

#include "attention_x3w.h"

extern "C" __attribute__((visibility("hidden"))) int fx3w_lds_bytes(void) { return X3W_KRING * 8192 + 2 * 16384 + 1024 + 65536; }

// launches attn_x3w_kernel<masks> for a descriptor whose k / vt are the pre-split images (kv_pair); returns a hipError_t
extern "C" __attribute__((visibility("hidden"))) int fx3w_launch(hipStream_t s, const ffn_attn_desc* d, int masks) {
    const int lds = fx3w_lds_bytes();
    static bool opted[2] = {false, false};
    if (!opted[masks ? 1 : 0]) {      // (idempotent: a race between two host threads sets the same attribute twice)
        const hipError_t e = masks ? hipFuncSetAttribute(reinterpret_cast<const void*>(attn_x3w_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)
                                   : hipFuncSetAttribute(reinterpret_cast<const void*>(attn_x3w_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        opted[masks ? 1 : 0] = true;
    }
    const dim3 grid(((d->S + 255) / 256) * d->heads * d->Bo);
    (void)hipGetLastError();
    if (masks) hipLaunchKernelGGL(attn_x3w_kernel<true>, grid, dim3(256), lds, s, *d);
    else hipLaunchKernelGGL(attn_x3w_kernel<false>, grid, dim3(256), lds, s, *d);
    return (int)hipGetLastError();
}
