// Correctness + timing harness for attn_x3w_kernel (attention_x3w.h) against an fp64 statement of the pass table and against attn_x3p_kernel -- no torch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DX3W_ABL=n] -o build/native/x3w_test tools/native/x3w_test.hip
//   ./x3w_test check                      small cases (ragged S, odd tile counts, masks + selectors + head rule, multi-pass, pair output) vs fp64
//   ./x3w_test time rows S heads passes   random fp32 operands, both kernels interleaved in one process
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <random>
#include <algorithm>
#include "../../freefine_amd/csrc/attention_x3p.h"
#include "../../freefine_amd/csrc/attention_x3w.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// fp64 statement: out[b, q, head*64 + d] = sum_p w_p wq_p[q] softmax_k(scale <Q, K> over allowed keys) V
__global__ void ref_kernel(const ffn_attn_desc p, double* out, const float* vnat /* V natural [B][Sk][C] */) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x, head = blockIdx.y, b = blockIdx.z;
    if (q >= p.S) return;
    const float* Q = (const float*)p.q;
    const float* K = (const float*)p.k;
    double acc[64];
    for (int d = 0; d < 64; ++d) acc[d] = 0.0;
    for (int pass = 0; pass < p.npass; ++pass) {
        const ffn_attn_entry& e = p.e[pass * FFN_ATT_MAXB + b];
        if (e.w_const == 0.f && e.w_slope == 0.f) continue;
        double w = e.w_const;
        if (p.w_dev) w += (double)e.w_slope * (double)(*p.w_dev);
        const int hb = e.hr_row > 0 ? e.hr_row - 1 : b;
        const bool masked = e.kmask && (!(e.flags & FFN_ATT_HEAD_RULE) || (((hb * p.heads + head) & 1) == 0));
        const int sel = e.qsel ? (e.qsel[q] != 0) : 1;
        const float* qv = Q + ((long)e.q_row * p.S + q) * p.ldq + head * 64;
        double m = -1e300;
        for (int k = 0; k < p.Sk; ++k) {
            if (masked && ((e.kmask[k] != 0) != (sel != 0))) continue;
            const float* kv = K + ((long)e.kv_row * p.Sk + k) * p.ldk + head * 64;
            double s = 0.0;
            for (int d = 0; d < 64; ++d) s += (double)qv[d] * (double)kv[d];
            s *= p.scale;
            m = s > m ? s : m;
        }
        double l = 0.0, o[64];
        for (int d = 0; d < 64; ++d) o[d] = 0.0;
        for (int k = 0; k < p.Sk; ++k) {
            if (masked && ((e.kmask[k] != 0) != (sel != 0))) continue;
            const float* kv = K + ((long)e.kv_row * p.Sk + k) * p.ldk + head * 64;
            double s = 0.0;
            for (int d = 0; d < 64; ++d) s += (double)qv[d] * (double)kv[d];
            const double pe = exp(s * p.scale - m);
            l += pe;
            const float* vv = vnat + ((long)e.kv_row * p.Sk + k) * p.ldk + head * 64;
            for (int d = 0; d < 64; ++d) o[d] += pe * (double)vv[d];
        }
        const double wqv = e.wq ? (double)e.wq[q] : 1.0;
        if (l > 0.0)
            for (int d = 0; d < 64; ++d) acc[d] += w * wqv * o[d] / l;
    }
    for (int d = 0; d < 64; ++d) out[((long)b * p.S + q) * (p.heads * 64) + head * 64 + d] = acc[d];
}

struct Problem {
    int B, S, heads, passes, C;
    float *dq, *dk, *dv, *dvt, *dout_p, *dout_w, *dw, *dwq;
    double* dref;
    bf16 *dkp, *dvp;
    uint8_t *dm, *dsel;
    size_t n;
};

static Problem make(int B, int S, int heads, int passes, unsigned seed, bool spikes) {
    Problem P; P.B = B; P.S = S; P.heads = heads; P.passes = passes; P.C = heads * 64;
    const int C = P.C;
    std::mt19937 rng(seed); std::normal_distribution<float> nd(0.f, 1.f);
    P.n = (size_t)B * S * C;
    std::vector<float> hq(P.n), hk(P.n), hv(P.n), hvt(P.n);
    for (auto& v : hq) v = nd(rng);
    for (auto& v : hk) v = nd(rng);
    for (auto& v : hv) v = nd(rng);
    if (spikes)                                             // a few dominant keys late in the sequence: forces the re-referencing path after the first tile
        for (int b = 0; b < B; ++b)
            for (int k : {S / 2 + 3, S - 5, S / 4 + 1})
                for (int c = 0; c < C; ++c) hk[((size_t)b * S + k) * C + c] *= 4.f;
    for (int b = 0; b < B; ++b)
        for (int k = 0; k < S; ++k)
            for (int c = 0; c < C; ++c) hvt[((size_t)b * C + c) * S + k] = hv[((size_t)b * S + k) * C + c];
    CK(hipMalloc(&P.dq, P.n * 4)); CK(hipMalloc(&P.dk, P.n * 4)); CK(hipMalloc(&P.dv, P.n * 4)); CK(hipMalloc(&P.dvt, P.n * 4));
    CK(hipMalloc(&P.dout_p, P.n * 4)); CK(hipMalloc(&P.dout_w, P.n * 4)); CK(hipMalloc(&P.dref, P.n * 8));
    CK(hipMalloc(&P.dkp, P.n * 4)); CK(hipMalloc(&P.dvp, P.n * 4));
    CK(hipMemcpy(P.dq, hq.data(), P.n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(P.dk, hk.data(), P.n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(P.dv, hv.data(), P.n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(P.dvt, hvt.data(), P.n * 4, hipMemcpyHostToDevice));
    std::vector<uint8_t> hm(S), hs(S);
    std::vector<float> hwq(S);
    for (int i = 0; i < S; ++i) { hm[i] = (rng() % 10) < 3; hs[i] = (rng() % 2); hwq[i] = 0.25f + (rng() % 100) * 0.01f; }
    for (int i = 64; i < 128 && i < S; ++i) hm[i] = 0;       // one tile with no allowed key for sel = 1 queries
    CK(hipMalloc(&P.dm, S)); CK(hipMalloc(&P.dsel, S)); CK(hipMalloc(&P.dw, 4)); CK(hipMalloc(&P.dwq, S * 4));
    CK(hipMemcpy(P.dm, hm.data(), S, hipMemcpyHostToDevice)); CK(hipMemcpy(P.dsel, hs.data(), S, hipMemcpyHostToDevice));
    CK(hipMemcpy(P.dwq, hwq.data(), S * 4, hipMemcpyHostToDevice));
    const float cg = 0.4f; CK(hipMemcpy(P.dw, &cg, 4, hipMemcpyHostToDevice));
    const long nk = (long)B * S * heads * 8;
    hipLaunchKernelGGL(attn_presplit_k_kernel, dim3(4096), dim3(256), 0, 0, P.dk, P.dkp, nk, heads, C);
    hipLaunchKernelGGL(attn_presplit_vt_kernel, dim3(4096), dim3(256), 0, 0, P.dvt, P.dvp, nk, S / 64, S);
    CK(hipDeviceSynchronize());
    return P;
}
static void destroy(Problem& P) {
    for (void* q : {(void*)P.dq, (void*)P.dk, (void*)P.dv, (void*)P.dvt, (void*)P.dout_p, (void*)P.dout_w, (void*)P.dref, (void*)P.dkp, (void*)P.dvp, (void*)P.dm, (void*)P.dsel, (void*)P.dw, (void*)P.dwq}) CK(hipFree(q));
}

// the TCA-shaped pass table of x3p_bench (passes = 2) or plain self attention (1); variant adds per-query weights / no head rule / a skipped entry
static ffn_attn_desc desc_for(const Problem& P, int b0, int nb, bool pair, int variant) {
    ffn_attn_desc d; memset(&d, 0, sizeof(d));
    const int C = P.C;
    d.q = P.dq; d.k = pair ? (const void*)P.dkp : (const void*)P.dk; d.vt = pair ? (const void*)P.dvp : (const void*)P.dvt; d.w_dev = P.dw;
    d.kv_pair = pair;
    d.Bo = nb; d.S = P.S; d.Sk = P.S; d.heads = P.heads; d.D = 64; d.ldq = C; d.ldk = C; d.ldvt = P.S; d.ldo = C; d.scale = 0.125f; d.npass = P.passes;
    for (int b = 0; b < nb; ++b) {
        if (P.passes == 1) { d.e[b].q_row = b0 + b; d.e[b].kv_row = b0 + b; d.e[b].w_const = 1.f; if (variant == 1) d.e[b].wq = P.dwq; }
        else {
            ffn_attn_entry& e = d.e[b]; e.q_row = b0 + b; e.kv_row = (b0 + b) | 1; if (e.kv_row >= P.B) e.kv_row = P.B - 1;
            e.w_const = 0.f; e.w_slope = 1.f; e.kmask = P.dm; e.qsel = P.dsel; e.flags = variant == 2 ? 0 : FFN_ATT_HEAD_RULE; e.hr_row = b0 + b + 1;
            if (variant == 1) e.wq = P.dwq;
            ffn_attn_entry& f = d.e[FFN_ATT_MAXB + b]; f.q_row = b0 + b; f.kv_row = b0 + b; f.w_const = 1.f; f.w_slope = -1.f;
            if (variant == 1) f.wq = P.dwq;
            if (variant == 3 && (b & 1)) { f.w_const = 0.f; f.w_slope = 0.f; }      // a skipped second pass
        }
    }
    return d;
}
constexpr int LDS_P = 5 * (2 * 8192) + 8 * 4 * 2 * 64 * 16 + 512;
constexpr int LDS_W = 6 * 8192 + 2 * 16384 + 1024 + 65536;
static void run_p(const Problem& P, int variant, float* out) {
    for (int b0 = 0; b0 < P.B; b0 += 16) {
        const int nb = std::min(16, P.B - b0);
        ffn_attn_desc d = desc_for(P, b0, nb, true, variant); d.out = out + (size_t)b0 * P.S * P.C;
        dim3 grid(((P.S + 255) / 256) * P.heads * nb);
        if (P.passes > 1) { CK(hipFuncSetAttribute((const void*)attn_x3p_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_P)); hipLaunchKernelGGL((attn_x3p_kernel<true, true>), grid, dim3(512), LDS_P, 0, d); }
        else { CK(hipFuncSetAttribute((const void*)attn_x3p_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_P)); hipLaunchKernelGGL((attn_x3p_kernel<false, true>), grid, dim3(512), LDS_P, 0, d); }
    }
}
static void run_w(const Problem& P, int variant, float* out) {
    for (int b0 = 0; b0 < P.B; b0 += 16) {
        const int nb = std::min(16, P.B - b0);
        ffn_attn_desc d = desc_for(P, b0, nb, true, variant); d.out = out + (size_t)b0 * P.S * P.C;
        dim3 grid(((P.S + 255) / 256) * P.heads * nb);
        if (P.passes > 1) { CK(hipFuncSetAttribute((const void*)attn_x3w_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_W)); hipLaunchKernelGGL((attn_x3w_kernel<true>), grid, dim3(256), LDS_W, 0, d); }
        else { CK(hipFuncSetAttribute((const void*)attn_x3w_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_W)); hipLaunchKernelGGL((attn_x3w_kernel<false>), grid, dim3(256), LDS_W, 0, d); }
    }
}
static void run_ref(const Problem& P, int variant) {
    for (int b0 = 0; b0 < P.B; b0 += 16) {
        const int nb = std::min(16, P.B - b0);
        ffn_attn_desc d = desc_for(P, b0, nb, false, variant);
        hipLaunchKernelGGL(ref_kernel, dim3((P.S + 63) / 64, P.heads, nb), dim3(64), 0, 0, d, P.dref + (size_t)b0 * P.S * P.C, P.dv);
    }
}
static int check_case(int B, int S, int heads, int passes, int variant, bool spikes) {
    Problem P = make(B, S, heads, passes, 7 + S + heads + passes, spikes);
    CK(hipMemset(P.dout_p, 0xff, P.n * 4)); CK(hipMemset(P.dout_w, 0xff, P.n * 4));
    run_ref(P, variant); run_p(P, variant, P.dout_p); run_w(P, variant, P.dout_w);
    CK(hipDeviceSynchronize());
    std::vector<float> hp(P.n), hw(P.n); std::vector<double> hr(P.n);
    CK(hipMemcpy(hp.data(), P.dout_p, P.n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hw.data(), P.dout_w, P.n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hr.data(), P.dref, P.n * 8, hipMemcpyDeviceToHost));
    double ep = 0, ew = 0, mx = 0; size_t bad = 0, worst = 0;
    for (size_t i = 0; i < P.n; ++i) {
        mx = std::max(mx, fabs(hr[i]));
        ep = std::max(ep, fabs(hp[i] - hr[i]));
        const double e = fabs(hw[i] - hr[i]);
        if (!(e <= 1e30)) ++bad;
        if (e > ew || !(e <= 1e30)) { ew = e; worst = i; }
    }
    const bool ok = bad == 0 && ew <= 1e-4 * std::max(1.0, mx);
    printf("%s B %d S %d heads %d passes %d variant %d spikes %d: |ref| max %.3f  x3p err %.2e  x3w err %.2e  (nan %zu; worst at b %zu q %zu c %zu)\n", ok ? "ok  " : "FAIL", B, S, heads, passes,
           variant, (int)spikes, mx, ep, ew, bad, worst / ((size_t)S * P.C), (worst / P.C) % S, worst % P.C);
    destroy(P);
    return ok ? 0 : 1;
}
int main(int argc, char** argv) {
    const char* mode = argc > 1 ? argv[1] : "check";
    if (!strcmp(mode, "check")) {
        int fails = 0;
        fails += check_case(2, 256, 5, 1, 0, false);
        fails += check_case(2, 512, 5, 1, 1, true);
        fails += check_case(3, 320, 5, 1, 0, true);          // ragged query count, odd tile count
        fails += check_case(3, 512, 5, 2, 0, true);
        fails += check_case(3, 576, 10, 2, 1, true);
        fails += check_case(2, 1024, 10, 2, 2, false);
        fails += check_case(4, 512, 5, 2, 3, true);
        fails += check_case(2, 128, 5, 2, 0, false);
        fails += check_case(1, 64, 5, 1, 0, false);          // one tile
        fails += check_case(18, 256, 2, 2, 0, false);        // more rows than one launch
        printf(fails ? "CHECK FAILED (%d)\n" : "CHECK OK\n", fails);
        return fails ? 1 : 0;
    }
    const int B = argc > 2 ? atoi(argv[2]) : 16, S = argc > 3 ? atoi(argv[3]) : 4096, heads = argc > 4 ? atoi(argv[4]) : 5, passes = argc > 5 ? atoi(argv[5]) : 1;
    const int reps = argc > 6 ? atoi(argv[6]) : 10;
    Problem P = make(B, S, heads, passes, 1, false);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double flops = 4.0 * passes * B * (double)S * S * P.C;
    float best[2] = {1e30f, 1e30f};
    run_p(P, 0, P.dout_p); run_w(P, 0, P.dout_w); CK(hipDeviceSynchronize());
    for (int round = 0; round < 4; ++round)
        for (int which = 0; which < 2; ++which) {
            CK(hipEventRecord(e0, 0));
            for (int rr = 0; rr < reps; ++rr) { if (which) run_w(P, 0, P.dout_w); else run_p(P, 0, P.dout_p); }
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1)); t = t * 1e3f / reps;
            best[which] = std::min(best[which], t);
        }
    // agreement of the two kernels on the timed problem
    std::vector<float> hp(P.n), hw(P.n);
    CK(hipMemcpy(hp.data(), P.dout_p, P.n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hw.data(), P.dout_w, P.n * 4, hipMemcpyDeviceToHost));
    double dd = 0; for (size_t i = 0; i < P.n; ++i) dd = std::max(dd, (double)fabs(hp[i] - hw[i]));
    printf("X3W_ABL=%d rows %d S %d heads %d passes %d: x3p %.1f us %.0f TFLOP/s (%.2f) | x3w %.1f us %.0f TFLOP/s (%.2f of 833) | max |x3p - x3w| %.2e\n", X3W_ABL, B, S, heads, passes,
           best[0], flops / best[0] * 1e-6, flops / best[0] * 1e-6 / 833.3, best[1], flops / best[1] * 1e-6, flops / best[1] * 1e-6 / 833.3, dd);
    destroy(P);
    return 0;
}
