// Probe (round 6): does a single wave's VALU work hide in the gaps of its own 32x32x16 MFMAs, and does it matter whether the MFMA accumulator
// lives in the architectural VGPRs (the -amdgpu-mfma-vgpr-form code attn_x3w_kernel uses) or in the accumulator half of the register file?
//   hipcc --offload-arch=gfx950 -O3 -o build/native/mfma_valu_overlap tools/native/mfma_valu_overlap.hip ; ./mfma_valu_overlap
// One wave per SIMD (256 threads per workgroup, 512 registers), 256 workgroups; per loop iteration 8 MFMAs (two chains of four) with F filler
// instructions after each.  Prints cycles per MFMA (s_memtime) for F = 0 .. 8, filler kinds fma / exp+fma / cvt_pk, accumulator in v or a.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int ACC_A, int F, int KIND, int B_A = 0>
__global__ __launch_bounds__(256) void probe(float* out, unsigned long long* cyc, int iters, float seed) {
    f32x16 c0, c1;
    for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; }
    u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = seed + threadIdx.x * 1e-3f + i;
    if (ACC_A) { asm volatile("" : "+a"(c0), "+a"(c1)); } else { asm volatile("" : "+v"(c0), "+v"(c1)); }
    asm volatile("" : "+v"(a));
    if (B_A) { asm volatile("" : "+a"(b)); } else { asm volatile("" : "+v"(b)); }
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#define FILL(k)                                                                                                           \
    if (F > k) {                                                                                                         \
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[k & 7]) : "v"(seed));                            \
        else if (KIND == 1) { if ((k & 3) == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(f[k & 7])); else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[k & 7]) : "v"(seed)); } \
        else asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(f[k & 7]) : "v"(seed));                                  \
    }
#define MM(c)                                                                                                            \
    if (ACC_A) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));                        \
    else if (B_A) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b));                     \
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));                              \
    FILL(0) FILL(1) FILL(2) FILL(3) FILL(4) FILL(5) FILL(6) FILL(7)
    for (int it = 0; it < iters; ++it) {
        MM(c0) MM(c0) MM(c0) MM(c0) MM(c1) MM(c1) MM(c1) MM(c1)
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (ACC_A) { asm volatile("s_nop 15\n\ts_nop 15" : "+a"(c0), "+a"(c1)); }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    for (int i = 0; i < 8; ++i) s += f[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int ACC_A, int F, int KIND, int B_A = 0>
static void run(float* out, unsigned long long* cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL((probe<ACC_A, F, KIND, B_A>), dim3(256), dim3(256), 0, 0, out, cyc, iters, 1.0001f);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((probe<ACC_A, F, KIND, B_A>), dim3(256), dim3(256), 0, 0, out, cyc, iters, 1.0001f);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(1024);
    CK(hipMemcpy(h.data(), cyc, 1024 * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double med = (double)h[512] / (iters * 8.0);
    printf("B in %s  acc %s  kind %s  fillers %d: %.1f cycles per MFMA (median wave), %.1f us, %.0f TFLOP/s\n", B_A ? "a" : "v", ACC_A ? "a" : "v", KIND == 0 ? "fma" : (KIND == 1 ? "exp+3fma" : "cvt_pk"), F, med, ms * 1e3,
           256.0 * 4 * iters * 8 * 32768.0 / (ms * 1e-3) * 1e-12);
}
#include <algorithm>
int main() {
    float* out; unsigned long long* cyc;
    CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&cyc, 1024 * 8));
#define ROW(A, K) run<A, 0, K>(out, cyc); run<A, 2, K>(out, cyc); run<A, 4, K>(out, cyc); run<A, 5, K>(out, cyc); run<A, 6, K>(out, cyc); run<A, 8, K>(out, cyc);
    if (getenv("PROBE_ALL")) { ROW(0, 0) ROW(1, 0) ROW(0, 1) ROW(1, 1) ROW(0, 2) ROW(1, 2) }
#define ROWB(K) run<0, 0, K, 1>(out, cyc); run<0, 2, K, 1>(out, cyc); run<0, 4, K, 1>(out, cyc); run<0, 5, K, 1>(out, cyc); run<0, 6, K, 1>(out, cyc);
    ROW(0, 0) ROWB(0) ROWB(1) ROWB(2)
    return 0;
}
