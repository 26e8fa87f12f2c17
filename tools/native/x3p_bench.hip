// Timing harness for attn_x3p_kernel (attention_x3p.h: split-bf16 self attention, ping-pong schedule) with ablation builds -- no torch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DX3P_ABL=1] -o build/native/x3p_bench tools/native/x3p_bench.hip
//   ./x3p_bench rows S heads passes(1|2) [reps]          random fp32 operands; TIMING ONLY (correctness: tests/test_ops_gpu.py)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <random>
#include "../../freefine_amd/csrc/attention_x3p.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 16, S = argc > 2 ? atoi(argv[2]) : 4096, heads = argc > 3 ? atoi(argv[3]) : 5;
    const int passes = argc > 4 ? atoi(argv[4]) : 1, reps = argc > 5 ? atoi(argv[5]) : 10;
    const int D = 64, C = heads * D;
    std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, 1.f);
    const size_t n = (size_t)B * S * C;
    std::vector<float> h(n);
    float *dq, *dk, *dv, *dout; uint8_t *dm, *dsel; float* dw;
    CK(hipMalloc(&dq, n * 4)); CK(hipMalloc(&dk, n * 4)); CK(hipMalloc(&dv, n * 4)); CK(hipMalloc(&dout, n * 4));
    for (float** pp : {&dq, &dk, &dv}) { for (auto& v : h) v = nd(rng); CK(hipMemcpy(*pp, h.data(), n * 4, hipMemcpyHostToDevice)); }
    std::vector<uint8_t> hm(S), hs(S);
    for (int i = 0; i < S; ++i) { hm[i] = (rng() % 10) < 3; hs[i] = (rng() % 2); }
    CK(hipMalloc(&dm, S)); CK(hipMalloc(&dsel, S)); CK(hipMalloc(&dw, 4));
    CK(hipMemcpy(dm, hm.data(), S, hipMemcpyHostToDevice)); CK(hipMemcpy(dsel, hs.data(), S, hipMemcpyHostToDevice));
    const float cg = 0.4f; CK(hipMemcpy(dw, &cg, 4, hipMemcpyHostToDevice));
    const bool masks = passes > 1;
    const bool pairkv = getenv("X3P_PAIRKV") && atoi(getenv("X3P_PAIRKV"));       // pre-split K / V^T images + LDS-DMA staging (the shipped path)
    bf16 *dkp = nullptr, *dvp = nullptr;
    if (pairkv) {
        CK(hipMalloc(&dkp, n * 4)); CK(hipMalloc(&dvp, n * 4));
        const long nk = (long)B * S * heads * 8;
        hipLaunchKernelGGL(attn_presplit_k_kernel, dim3(4096), dim3(256), 0, 0, dk, dkp, nk, heads, C);
        hipLaunchKernelGGL(attn_presplit_vt_kernel, dim3(4096), dim3(256), 0, 0, dv, dvp, nk, S / 64, S);
        CK(hipDeviceSynchronize());
    }
    auto run = [&](int b0, int nb) {
        ffn_attn_desc d; memset(&d, 0, sizeof(d));
        d.q = dq; d.k = pairkv ? (const void*)dkp : (const void*)dk; d.vt = pairkv ? (const void*)dvp : (const void*)dv; d.out = dout + (size_t)b0 * S * C; d.w_dev = dw;
        d.kv_pair = pairkv;
        d.Bo = nb; d.S = S; d.Sk = S; d.heads = heads; d.D = D; d.ldq = C; d.ldk = C; d.ldvt = S; d.ldo = C; d.scale = 0.125f; d.npass = passes;
        for (int b = 0; b < nb; ++b) {
            if (passes == 1) { d.e[b].q_row = b0 + b; d.e[b].kv_row = b0 + b; d.e[b].w_const = 1.f; }
            else {
                ffn_attn_entry& e = d.e[b]; e.q_row = b0 + b; e.kv_row = (b0 + b) | 1; e.w_const = 0.f; e.w_slope = 1.f; e.kmask = dm; e.qsel = dsel; e.flags = FFN_ATT_HEAD_RULE; e.hr_row = b0 + b + 1;
                ffn_attn_entry& f = d.e[FFN_ATT_MAXB + b]; f.q_row = b0 + b; f.kv_row = b0 + b; f.w_const = 1.f; f.w_slope = -1.f;
            }
        }
        constexpr int lds = 5 * (2 * 8192) + 8 * 4 * 2 * 64 * 16;
        dim3 grid(((S + 255) / 256) * heads * nb);
        if (pairkv) {
            if (masks) { CK(hipFuncSetAttribute((const void*)attn_x3p_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds + 512)); hipLaunchKernelGGL((attn_x3p_kernel<true, true>), grid, dim3(512), lds + 512, 0, d); }
            else { CK(hipFuncSetAttribute((const void*)attn_x3p_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds + 512)); hipLaunchKernelGGL((attn_x3p_kernel<false, true>), grid, dim3(512), lds + 512, 0, d); }
            return;
        }
        if (masks) { CK(hipFuncSetAttribute((const void*)attn_x3p_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); hipLaunchKernelGGL(attn_x3p_kernel<true>, grid, dim3(512), lds, 0, d); }
        else { CK(hipFuncSetAttribute((const void*)attn_x3p_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); hipLaunchKernelGGL(attn_x3p_kernel<false>, grid, dim3(512), lds, 0, d); }
    };
    auto run_all = [&]() { for (int b0 = 0; b0 < B; b0 += 16) run(b0, B - b0 < 16 ? B - b0 : 16); };
    run_all(); CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double flops = 4.0 * passes * B * (double)S * S * C;
    float best = 1e30f;
    for (int round = 0; round < 3; ++round) {
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) run_all();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1)); t = t * 1e3f / reps;
        if (t < best) best = t;
    }
    printf("pairkv=%d X3P_ABL=%d rows %d S %d heads %d passes %d: %.1f us  %.0f TFLOP/s nominal (%.2f of 833)\n", (int)pairkv, X3P_ABL, B, S, heads, passes, best, flops / best * 1e-6, flops / best * 1e-6 / 833.3);
    return 0;
}
