// Standalone A/B harness for the ping-pong igemm kernel (igemm_p8.h) against the 2-stage kernels of igemm.h: same operands,
// bitwise output comparison, interleaved timing in one process (cdna_hip_programming.md rule 24).  No torch: starts in seconds.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gpurun_out/pp_bench tools/native/pp_bench.hip
//   ./pp_bench dense M N K | conv B HW Cin Cout
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
static void launch(K kern, int grid, int threads, int lds, const ffn_igemm_desc& d) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, 0, d);
}

template <typename K>
static void launch_pp(K kern, int grid, int threads, int lds, const ffn_igemm_desc& d) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, 0, d, 1);
}

int main(int argc, char** argv) {
    if (argc < 5) { printf("usage\n"); return 1; }
    const bool conv = !strcmp(argv[1], "conv");
    ffn_igemm_desc d;
    memset(&d, 0, sizeof(d));
    long a_elems;
    if (conv) {
        const int B = atoi(argv[2]), HW = atoi(argv[3]), Cin = atoi(argv[4]), Cout = atoi(argv[5]);
        d.M = B * HW * HW; d.N = Cout; d.K = 9 * Cin; d.Kpad = d.K; d.conv = 1;
        d.Hin = d.Win = d.Hout = d.Wout = HW; d.Cin = Cin; d.stride = 1; d.pad = 1; d.rows_per_batch = HW * HW;
        a_elems = (long)d.M * Cin;
    } else {
        d.M = atoi(argv[2]); d.N = atoi(argv[3]); d.K = atoi(argv[4]); d.Kpad = d.K; d.lda = d.K; d.rows_per_batch = d.M;
        a_elems = (long)d.M * d.K;
    }
    d.ldo = d.N; d.alpha = 1.f; d.splitk = 1;
    const int na = conv ? 6 : 5;
    const int reps = argc > na ? atoi(argv[na]) : 20;
    const char* opt = argc > na + 1 ? argv[na + 1] : "";
    const bool use_res = strchr(opt, 'r'), use_rb = strchr(opt, 't'), use_geglu = strchr(opt, 'g');
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<uint16_t> hA(a_elems), hW((long)d.N * d.K);
    for (auto& v : hA) v = f2bf(nd(rng));
    const float ws = 1.f / sqrtf((float)d.K);
    for (auto& v : hW) v = f2bf(nd(rng) * ws);
    std::vector<float> hb(d.N);
    for (auto& v : hb) v = nd(rng);
    void *dA, *dW, *dO0, *dO1; float* dB;
    CK(hipMalloc(&dA, a_elems * 2)); CK(hipMalloc(&dW, hW.size() * 2));
    CK(hipMalloc(&dO0, (long)d.M * d.N * 2)); CK(hipMalloc(&dO1, (long)d.M * d.N * 2)); CK(hipMalloc(&dB, d.N * 4));
    CK(hipMemcpy(dA, hA.data(), a_elems * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hb.data(), d.N * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dO0, 0xff, (long)d.M * d.N * 2)); CK(hipMemset(dO1, 0xff, (long)d.M * d.N * 2));
    d.A = dA; d.W = dW; d.bias = dB;
    const int nbatch = d.M / d.rows_per_batch;
    if (use_rb) {
        std::vector<float> hrb((long)nbatch * d.N);
        for (auto& v : hrb) v = nd(rng);
        float* dRb; CK(hipMalloc(&dRb, hrb.size() * 4)); CK(hipMemcpy(dRb, hrb.data(), hrb.size() * 4, hipMemcpyHostToDevice));
        d.rowbias = dRb; d.ldrb = d.N;
    }
    if (use_res) {
        std::vector<uint16_t> hr((long)d.M * d.N);
        for (auto& v : hr) v = f2bf(nd(rng));
        void* dR; CK(hipMalloc(&dR, hr.size() * 2)); CK(hipMemcpy(dR, hr.data(), hr.size() * 2, hipMemcpyHostToDevice));
        d.residual = dR; d.ldr = d.N;
    }
    if (use_geglu) { d.flags |= FFN_IG_GEGLU; d.ldo = d.N / 2; }

    auto run_ref = [&](void* out) {
        ffn_igemm_desc e = d; e.out = out;
        const int nt = ((d.M + 127) / 128) * ((d.N + 319) / 320);
        const int grid = nt < 256 ? nt : 256;
        if (conv) launch(igemm_glds_kernel<bf16, 128, 320, AMODE_CONV3, true, 2, 4, 4, true>, grid, 1024, 2 * (128 + 320) * 128, e);
        else launch(igemm_glds_kernel<bf16, 128, 320, AMODE_DENSE, true, 2, 4, 4, true>, grid, 1024, 2 * (128 + 320) * 128, e);
    };
    auto run_ref256 = [&](void* out) {
        ffn_igemm_desc e = d; e.out = out;
        const int nt = ((d.M + 255) / 256) * ((d.N + 255) / 256);
        const int grid = nt < 256 ? nt : 256;
        if (conv) launch(igemm_glds_kernel<bf16, 256, 256, AMODE_CONV3, true, 2, 4, 4, true>, grid, 1024, 2 * (256 + 256) * 128, e);
        else launch(igemm_glds_kernel<bf16, 256, 256, AMODE_DENSE, true, 2, 4, 4, true>, grid, 1024, 2 * (256 + 256) * 128, e);
    };
    const int bm = getenv("PP_BM") ? atoi(getenv("PP_BM")) : 256;
    auto run_pp = [&](void* out, int bn) {
        ffn_igemm_desc e = d; e.out = out;
        const int nt = ((d.M + bm - 1) / bm) * (d.N / bn);
        const int gmax = getenv("PP_GRID") ? atoi(getenv("PP_GRID")) : 256; const int grid = nt < gmax ? nt : gmax;
        const int lds = 2 * (bm + bn) * 128 + 12288;
#define PPL(BM_, BN_, AM_, R_, G_) launch_pp(igemm_pp_kernel<BM_, BN_, AM_, R_, G_>, grid, 512, lds, e)
#define PPB(BM_)                                                                                                   \
        if (bn == 320) {                                                                                           \
            if (conv) { if (e.residual) PPL(BM_, 320, AMODE_CONV3, true, false); else PPL(BM_, 320, AMODE_CONV3, false, false); } \
            else { if (e.residual) PPL(BM_, 320, AMODE_DENSE, true, false); else PPL(BM_, 320, AMODE_DENSE, false, false); }      \
        } else {                                                                                                   \
            if (conv) { if (e.residual) PPL(BM_, 256, AMODE_CONV3, true, false); else PPL(BM_, 256, AMODE_CONV3, false, false); } \
            else if (e.flags & FFN_IG_GEGLU) PPL(BM_, 256, AMODE_DENSE, false, true);                              \
            else { if (e.residual) PPL(BM_, 256, AMODE_DENSE, true, false); else PPL(BM_, 256, AMODE_DENSE, false, false); }      \
        }
        if (bm == 256) { PPB(256) } else { PPB(192) }
    };
    const int bn = (d.N % 320 == 0 && !use_geglu) ? 320 : 256;
    if (use_geglu) run_ref256(dO0); else run_ref(dO0);
    CK(hipDeviceSynchronize());
    run_pp(dO1, bn);
    CK(hipDeviceSynchronize());
#ifdef PP_TRACE
    {
        run_pp(dO1, bn); CK(hipDeviceSynchronize());
        std::vector<unsigned long long> tr(256 * 64);
        CK(hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(pp_trace), tr.size() * 8));
        unsigned long long t0 = ~0ull;
        for (int w = 0; w < 256; ++w) if (tr[w * 64] && tr[w * 64] < t0) t0 = tr[w * 64];
        for (int k = 0; k < 24; ++k) {
            double mn = 1e30, mx = 0, dur = 0, dmx = 0; int n = 0;
            for (int w = 0; w < 256; ++w) {
                if (!tr[w * 64 + 2 * k]) continue;
                const double a = (tr[w * 64 + 2 * k] - t0) * 0.01, b = (tr[w * 64 + 2 * k + 1] - t0) * 0.01;
                mn = a < mn ? a : mn; mx = a > mx ? a : mx; dur += b - a; dmx = (b - a) > dmx ? b - a : dmx; ++n;
            }
            if (n) printf("tile %2d: low-row store section starts %.1f .. %.1f us after the first (spread %.1f), lasts %.2f us on average (max %.2f), %d workgroups\n", k, mn, mx, mx - mn, dur / n, dmx, n);
        }
        for (int w : {0, 1, 8, 9, 64, 65, 200}) { printf("wg %3d starts:", w); for (int k = 0; k < 8; ++k) printf(" %.1f", (tr[w * 64 + 2 * k] - t0) * 0.01); printf("\n"); }
    }
#endif
    std::vector<uint16_t> o0((long)d.M * d.N), o1((long)d.M * d.N);
    CK(hipMemcpy(o0.data(), dO0, o0.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(o1.data(), dO1, o1.size() * 2, hipMemcpyDeviceToHost));
    long ndiff = 0, nbig = 0; double maxd = 0;
    const size_t nout = (size_t)d.M * d.ldo;
    for (size_t i = 0; i < nout; ++i) {
        if (o0[i] != o1[i]) {
            ++ndiff;
            { int da = (int)o0[i] - (int)o1[i]; if (da < -1 || da > 1) ++nbig; }
            uint32_t a = (uint32_t)o0[i] << 16, b = (uint32_t)o1[i] << 16; float fa, fb; memcpy(&fa, &a, 4); memcpy(&fb, &b, 4);
            double dd = fabs((double)fa - fb); if (!(dd <= maxd)) maxd = dd;
            if (dd > 0.05 * (fabs(fa) + 1e-2) && nbig <= 5) printf("  diff at m=%ld n=%ld: ref %g pp %g\n", (long)(i / d.ldo), (long)(i % d.ldo), fa, fb);
        }
    }
    printf("%s M=%d N=%d K=%d opt=%s  pp BN=%d: %ld / %zu outputs differ from the 2-stage kernel, %ld by more than 1 bf16 ulp (max |diff| %g)\n", conv ? "conv" : "dense", d.M,
           d.N, d.K, opt, bn, ndiff, nout, nbig, maxd);

    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double flops = 2.0 * d.M * d.N * d.K;
    for (int round = 0; round < 3; ++round) {
        float t[3];
        for (int v = 0; v < 3; ++v) {
            CK(hipEventRecord(e0, 0));
            for (int r = 0; r < reps; ++r) { if (v == 0) run_ref(dO0); else if (v == 1) run_ref256(dO0); else run_pp(dO1, bn); }
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&t[v], e0, e1));
            t[v] = t[v] * 1e3f / reps;
        }
        printf("round %d: 128x320 %.1f us (%.0f TF)   256x256/16w %.1f us (%.0f TF)   pp%dx%d %.1f us (%.0f TF)\n", round, t[0], flops / t[0] * 1e-6, t[1],
               flops / t[1] * 1e-6, bm, bn, t[2], flops / t[2] * 1e-6);
    }
    {   // determinism = race detector: the kernel's arithmetic order is fixed, so two runs must agree bit for bit (20 pairs, back to back)
        std::vector<uint16_t> r0((long)d.M * d.ldo), r1((long)d.M * d.ldo);
        long bad = 0;
        for (int it = 0; it < 20; ++it) {
            CK(hipMemset(dO0, 0xff, (long)d.M * d.N * 2)); CK(hipMemset(dO1, 0xff, (long)d.M * d.N * 2));
            run_pp(dO0, bn); run_pp(dO1, bn); CK(hipDeviceSynchronize());
            CK(hipMemcpy(r0.data(), dO0, r0.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(r1.data(), dO1, r1.size() * 2, hipMemcpyDeviceToHost));
            if (memcmp(r0.data(), r1.data(), r0.size() * 2)) ++bad;
        }
        printf("determinism: %ld of 20 run pairs differ\n", bad);
    }
    return 0;
}
