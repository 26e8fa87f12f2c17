// Standalone A/B harness: attn_pp_kernel (attention_pp.h) against attn_kernel<bf16, 64, 2, 64, 2, MASKS> (attention.h) on the same
// operands and pass table; output comparison + interleaved timing in one process.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o build/attn_bench tools/native/attn_bench.hip
//   ./attn_bench rows S heads passes(1|2) [reps]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <random>
#include "../../freefine_amd/csrc/attention_pp.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 16, S = argc > 2 ? atoi(argv[2]) : 4096, heads = argc > 3 ? atoi(argv[3]) : 5;
    const int passes = argc > 4 ? atoi(argv[4]) : 1, reps = argc > 5 ? atoi(argv[5]) : 10;
    const int D = 64, C = heads * D;
    std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, 1.f);
    const size_t n = (size_t)B * S * C;
    std::vector<uint16_t> hq(n), hk(n), hv(n);
    for (auto& v : hq) v = f2bf(nd(rng)); for (auto& v : hk) v = f2bf(nd(rng)); for (auto& v : hv) v = f2bf(nd(rng));
    std::vector<uint8_t> hm(S), hs(S);
    for (int i = 0; i < S; ++i) { hm[i] = (rng() % 10) < 3; hs[i] = (rng() % 2); }
    void *dq, *dk, *dv, *do0, *do1; uint8_t *dm, *dsel; float* dw;
    CK(hipMalloc(&dq, n * 2)); CK(hipMalloc(&dk, n * 2)); CK(hipMalloc(&dv, n * 2)); CK(hipMalloc(&do0, n * 2)); CK(hipMalloc(&do1, n * 2));
    CK(hipMalloc(&dm, S)); CK(hipMalloc(&dsel, S)); CK(hipMalloc(&dw, 4));
    CK(hipMemcpy(dq, hq.data(), n * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dk, hk.data(), n * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dv, hv.data(), n * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dm, hm.data(), S, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsel, hs.data(), S, hipMemcpyHostToDevice));
    const float cg = 0.4f; CK(hipMemcpy(dw, &cg, 4, hipMemcpyHostToDevice));
    CK(hipMemset(do0, 0xff, n * 2)); CK(hipMemset(do1, 0xff, n * 2));
    const bool masks = passes > 1;
    auto run = [&](int which, int b0, int nb, void* out) {
        ffn_attn_desc d; memset(&d, 0, sizeof(d));
        d.q = dq; d.k = dk; d.vt = dv; d.out = (char*)out + (size_t)b0 * S * C * 2; d.w_dev = dw;
        d.Bo = nb; d.S = S; d.Sk = S; d.heads = heads; d.D = D; d.ldq = C; d.ldk = C; d.ldvt = S; d.ldo = C; d.scale = 0.125f; d.npass = passes;
        for (int b = 0; b < nb; ++b) {
            if (passes == 1) { d.e[b].q_row = b0 + b; d.e[b].kv_row = b0 + b; d.e[b].w_const = 1.f; }
            else {
                ffn_attn_entry& e = d.e[b]; e.q_row = b0 + b; e.kv_row = (b0 + b) | 1; e.w_const = 0.f; e.w_slope = 1.f; e.kmask = dm; e.qsel = dsel; e.flags = FFN_ATT_HEAD_RULE; e.hr_row = b0 + b + 1;
                ffn_attn_entry& f = d.e[FFN_ATT_MAXB + b]; f.q_row = b0 + b; f.kv_row = b0 + b; f.w_const = 1.f; f.w_slope = -1.f;
            }
        }
        if (which == 0) {
            const int lds = 2 * (64 * 128 + 64 * 128) + 4 * 4 * 2 * 64 * 16 + 2 * 64;
            dim3 grid(((S + 127) / 128) * heads * nb);
            if (masks) { CK(hipFuncSetAttribute((const void*)attn_kernel<bf16, 64, 2, 64, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); hipLaunchKernelGGL((attn_kernel<bf16, 64, 2, 64, 2, true>), grid, dim3(256), lds, 0, d); }
            else { CK(hipFuncSetAttribute((const void*)attn_kernel<bf16, 64, 2, 64, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); hipLaunchKernelGGL((attn_kernel<bf16, 64, 2, 64, 2, false>), grid, dim3(256), lds, 0, d); }
        } else {
            const int lds = 4 * 8192 + 4 * 8192 + 4 * 256 + 8 * 4 * 2 * 64 * 16;
            dim3 grid(((S + 255) / 256) * heads * nb);
            if (masks) { CK(hipFuncSetAttribute((const void*)attn_pp_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); hipLaunchKernelGGL(attn_pp_kernel<true>, grid, dim3(512), lds, 0, d); }
            else { CK(hipFuncSetAttribute((const void*)attn_pp_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); hipLaunchKernelGGL(attn_pp_kernel<false>, grid, dim3(512), lds, 0, d); }
        }
    };
    auto run_all = [&](int which, void* out) { for (int b0 = 0; b0 < B; b0 += 16) run(which, b0, B - b0 < 16 ? B - b0 : 16, out); };
    run_all(0, do0); CK(hipDeviceSynchronize()); run_all(1, do1); CK(hipDeviceSynchronize());
    std::vector<uint16_t> o0(n), o1(n);
    CK(hipMemcpy(o0.data(), do0, n * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(o1.data(), do1, n * 2, hipMemcpyDeviceToHost));
    double maxd = 0, maxv = 0; size_t nd2 = 0;
    for (size_t i = 0; i < n; ++i) { double a = bf2f(o0[i]), b = bf2f(o1[i]); double dd = fabs(a - b); if (!(dd <= maxd)) maxd = dd; if (fabs(a) > maxv) maxv = fabs(a); if (o0[i] != o1[i]) ++nd2; }
    printf("rows %d S %d heads %d passes %d: max |old - pp| = %g (max |old| %g), %zu / %zu elements differ\n", B, S, heads, passes, maxd, maxv, nd2, n);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double flops = 4.0 * passes * B * (double)S * S * C;
    for (int round = 0; round < 3; ++round) {
        float t[2];
        for (int v = 0; v < 2; ++v) {
            CK(hipEventRecord(e0, 0));
            for (int r = 0; r < reps; ++r) run_all(v, v ? do1 : do0);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t[v], e0, e1)); t[v] = t[v] * 1e3f / reps;
        }
        printf("round %d: attn_kernel %.1f us (%.0f TF)   attn_pp %.1f us (%.0f TF)\n", round, t[0], flops / t[0] * 1e-6, t[1], flops / t[1] * 1e-6);
    }
    {   // determinism = race detector: two runs of the ping-pong kernel must agree bit for bit (20 pairs)
        std::vector<uint16_t> r0(n), r1(n);
        long bad = 0;
        for (int it = 0; it < 20; ++it) {
            CK(hipMemset(do0, 0xff, n * 2)); CK(hipMemset(do1, 0xff, n * 2));
            run_all(1, do0); run_all(1, do1); CK(hipDeviceSynchronize());
            CK(hipMemcpy(r0.data(), do0, n * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(r1.data(), do1, n * 2, hipMemcpyDeviceToHost));
            if (memcmp(r0.data(), r1.data(), n * 2)) ++bad;
        }
        printf("determinism: %ld of 20 run pairs differ\n", bad);
    }
    return 0;
}
