// Timing harness for the split-bf16 (FFN_BF16X3, blocked operands) ping-pong kernels of igemm_p8.h -- no torch, starts in seconds.  Operands are random
// bf16 in the blocked pair geometry (values are not a split of anything: TIMING ONLY; correctness lives in tests/test_ops_gpu.py).  Ablation builds:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DPP_ABL=n] -o gpurun_out/x3_bench[_abln] tools/native/x3_bench.hip
//   ./x3_bench dense M N K [opt] [BM] | conv B HW Cin Cout [opt] [BM]       opt: r = fp32 residual, g = GEGLU (pair output), t = row bias, - = none
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <random>
#include "../../freefine_amd/csrc/igemm_p8.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }

template <typename K>
static void launch_pp(K kern, int grid, int lds, const ffn_igemm_desc& d, int splitk) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, d, splitk);
}

int main(int argc, char** argv) {
    if (argc < 5) { printf("usage: dense M N K [opt] [BM] | conv B HW Cin Cout [opt] [BM]\n"); return 1; }
    const bool conv = !strcmp(argv[1], "conv");
    ffn_igemm_desc d;
    memset(&d, 0, sizeof(d));
    long a_elems;
    int Kr;
    if (conv) {
        const int B = atoi(argv[2]), HW = atoi(argv[3]), Cin = atoi(argv[4]), Cout = atoi(argv[5]);
        Kr = 9 * Cin;
        d.M = B * HW * HW; d.N = Cout; d.conv = 1; d.lda = 2 * Cin;
        d.Hin = d.Win = d.Hout = d.Wout = HW; d.Cin = Cin; d.stride = 1; d.pad = 1; d.rows_per_batch = HW * HW;
        a_elems = (long)d.M * 2 * Cin;
    } else {
        d.M = atoi(argv[2]); d.N = atoi(argv[3]); Kr = atoi(argv[4]); d.lda = 2 * Kr; d.rows_per_batch = 4096;
        a_elems = (long)d.M * 2 * Kr;
    }
    d.K = 3 * Kr; d.Kpad = 2 * Kr; d.x3 = 2; d.a_lo = 32; d.alpha = 1.f; d.splitk = 1; d.flags = FFN_IG_OUT_F32;
    const int na = conv ? 6 : 5;
    const char* opt = argc > na ? argv[na] : "-";
    const int bm = argc > na + 1 ? atoi(argv[na + 1]) : 256;
    const bool use_res = strchr(opt, 'r'), use_geglu = strchr(opt, 'g'), use_rb = strchr(opt, 't');
    const int bn = (d.N % 320 == 0 && !use_geglu) ? 320 : 256;
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<uint16_t> hA(a_elems), hW((long)d.N * d.Kpad);
    for (auto& v : hA) v = f2bf(nd(rng));
    const float ws = 1.f / sqrtf((float)Kr);
    for (auto& v : hW) v = f2bf(nd(rng) * ws);
    std::vector<float> hb(d.N);
    for (auto& v : hb) v = nd(rng);
    void *dA, *dW, *dO; float* dB;
    d.ldo = use_geglu ? d.N : d.N;      // GEGLU: pair rows of N/2 columns = N bf16; else fp32 [M][N]
    const long obytes = (long)d.M * d.ldo * (use_geglu ? 2 : 4);
    CK(hipMalloc(&dA, a_elems * 2)); CK(hipMalloc(&dW, hW.size() * 2)); CK(hipMalloc(&dO, obytes)); CK(hipMalloc(&dB, d.N * 4));
    CK(hipMemcpy(dA, hA.data(), a_elems * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hb.data(), d.N * 4, hipMemcpyHostToDevice));
    d.A = dA; d.W = dW; d.bias = dB; d.out = dO;
    if (use_rb) {
        const int nbatch = (d.M + d.rows_per_batch - 1) / d.rows_per_batch;
        std::vector<float> hrb((long)nbatch * d.N, 0.5f);
        float* dRb; CK(hipMalloc(&dRb, hrb.size() * 4)); CK(hipMemcpy(dRb, hrb.data(), hrb.size() * 4, hipMemcpyHostToDevice));
        d.rowbias = dRb; d.ldrb = d.N;
    }
    if (use_res) {
        void* dR; CK(hipMalloc(&dR, (long)d.M * d.N * 4)); CK(hipMemset(dR, 0, (long)d.M * d.N * 4));
        d.residual = dR; d.ldr = d.N;
    }
    if (use_geglu) d.flags |= FFN_IG_GEGLU | FFN_IG_OUT_PAIR;
    const int nt = ((d.M + bm - 1) / bm) * (d.N / bn);
    const int grid = nt < 256 ? nt : 256;
#ifdef PP_STAMP
    const int lds = 2 * (bm + bn) * 128 + 12288 + 4096;
#else
    const int lds = 2 * (bm + bn) * 128 + 12288;
#endif
    auto run = [&]() {
#define PPL(BM_, BN_, AM_, R_, G_) launch_pp(igemm_pp_kernel<BM_, BN_, AM_, R_, G_, false, false, true, false>, grid, lds, d, 1)
#define PPB(BM_)                                                                                                                     \
        if (bn == 320) {                                                                                                             \
            if (conv) { if (use_res) PPL(BM_, 320, AMODE_CONV3, true, false); else PPL(BM_, 320, AMODE_CONV3, false, false); }      \
            else { if (use_res) PPL(BM_, 320, AMODE_DENSE, true, false); else PPL(BM_, 320, AMODE_DENSE, false, false); }           \
        } else {                                                                                                                     \
            if (conv) { if (use_res) PPL(BM_, 256, AMODE_CONV3, true, false); else PPL(BM_, 256, AMODE_CONV3, false, false); }      \
            else if (use_geglu) PPL(BM_, 256, AMODE_DENSE, false, true);                                                            \
            else { if (use_res) PPL(BM_, 256, AMODE_DENSE, true, false); else PPL(BM_, 256, AMODE_DENSE, false, false); }           \
        }
        if (bm == 256) { PPB(256) } else { PPB(192) }
    };
    run();
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double flops = 2.0 * d.M * d.N * Kr;
    const int reps = 20;
    float best = 1e30f;
    for (int round = 0; round < 4; ++round) {
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) run();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        t = t * 1e3f / reps;
        if (t < best) best = t;
    }
#ifdef PP_STAMP
    {   // one launch's stamps of workgroup 7, stages 24-27: per wave the segment boundaries (0-7) and three points inside phase L's load section (8-10)
        run(); CK(hipDeviceSynchronize());
        std::vector<unsigned long long> st(512);
        CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(pp_stamp_out), 512 * 8));
        printf("segments (cycles), mean over stages 24-27:   load L [reads issued | DMA issued | vmcnt wait | lgkm wait]  barrier   mfma L   barrier   load H   barrier   mfma H   barrier\n");
        for (int w = 0; w < 8; ++w) {
            double seg[8] = {0}, sub[4] = {0};
            for (int g = 0; g < 4; ++g) {
                const unsigned long long* t = &st[w * 64 + g * 16];
                for (int k = 0; k < 7; ++k) seg[k] += (double)(t[k + 1] - t[k]) / 4;
                if (g < 3) seg[7] += (double)(t[16] - t[7]) / 3;
                sub[0] += (double)(t[8] - t[0]) / 4; sub[1] += (double)(t[9] - t[8]) / 4; sub[2] += (double)(t[10] - t[9]) / 4; sub[3] += (double)(t[1] - t[10]) / 4;
            }
            printf("wave %d: %8.0f [%5.0f %5.0f %5.0f %5.0f] %6.0f %8.0f %6.0f %8.0f %6.0f %8.0f %6.0f\n", w, seg[0], sub[0], sub[1], sub[2], sub[3], seg[1], seg[2], seg[3], seg[4], seg[5], seg[6], seg[7]);
        }
    }
#endif
    printf("PP_ABL=%d %s M=%d N=%d K=%d opt=%s tile %dx%d: %.1f us  %.0f TFLOP/s (%.2f of 833)\n", PP_ABL, conv ? "conv" : "dense", d.M, d.N, Kr, opt, bm, bn, best,
           flops / best * 1e-6, flops / best * 1e-6 / 833.3);
    return 0;
}
