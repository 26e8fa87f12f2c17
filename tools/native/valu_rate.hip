// Issue-rate probe for the VALU instructions of the attention softmax segment on gfx950: one wave per SIMD (256 threads per
// workgroup, one workgroup per CU), N independent instructions per loop trip, cycles from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o build/valu_rate tools/native/valu_rate.hip ; ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)
template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, long long* cyc, int trips) {
    float a = threadIdx.x * 1e-3f, b = a + 1.f, c = a + 2.f, d = a + 3.f, e = a + 4.f, f = a + 5.f, g = a + 6.f, h = a + 7.f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p = {a, b}, q = {c, d}, r = {e, f}, s = {g, h};
    unsigned u0 = 0, u1 = 0, u2 = 0, u3 = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < trips; ++i) {
        if (MODE == 0) { REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (MODE == 1) { REP8(asm volatile("v_cvt_pk_bf16_f32 %0, %4, %5\n v_cvt_pk_bf16_f32 %1, %5, %6\n v_cvt_pk_bf16_f32 %2, %6, %7\n v_cvt_pk_bf16_f32 %3, %7, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a), "v"(b), "v"(c), "v"(d));) }
        if (MODE == 2) { REP8(asm volatile("v_max3_f32 %0, %0, %4, %5\n v_max3_f32 %1, %1, %5, %6\n v_max3_f32 %2, %2, %6, %7\n v_max3_f32 %3, %3, %7, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h));) }
        if (MODE == 3) { REP8(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(p), "+v"(q), "+v"(r), "+v"(s) : "v"(p));) }
        if (MODE == 4) { REP8(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));) }
        if (MODE == 5) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(p), "+v"(q), "+v"(r), "+v"(s) : "v"(p), "v"(q));) }
        if (MODE == 6) { REP8(asm volatile("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (MODE == 7) { REP8(asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %4, %4, %6, %7\n v_exp_f32 %1, %1\n v_fma_f32 %5, %5, %6, %7\n v_exp_f32 %2, %2\n v_fma_f32 %4, %4, %6, %7\n v_exp_f32 %3, %3\n v_fma_f32 %5, %5, %6, %7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "v"(g), "v"(h));) }
        if (MODE == 8) { REP8(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u0));) }
        if (MODE == 9) { REP8(asm volatile("v_ldexp_f32 %0, %0, %4\n v_ldexp_f32 %1, %1, %4\n v_ldexp_f32 %2, %2, %4\n v_ldexp_f32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(u0));) }
        if (MODE == 10) { REP8(asm volatile("v_fract_f32 %0, %0\n v_fract_f32 %1, %1\n v_fract_f32 %2, %2\n v_fract_f32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (MODE == 11) { REP8(asm volatile("v_cvt_i32_f32 %0, %4\n v_cvt_i32_f32 %1, %5\n v_cvt_i32_f32 %2, %6\n v_cvt_i32_f32 %3, %7" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a), "v"(b), "v"(c), "v"(d));) }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + e + f + g + h + p.x + q.y + r.x + s.y + (float)(u0 + u1 + u2 + u3);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> static int run(const char* name, int nblk, int nthr_waves) {
    float* out; long long* cyc; CK(hipMalloc(&out, 1024 * 1024 * 4)); CK(hipMalloc(&cyc, 4096 * 8));
    const int trips = 2000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe<MODE>, dim3(nblk), dim3(64 * nthr_waves), 0, 0, out, cyc, trips); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<MODE>, dim3(nblk), dim3(64 * nthr_waves), 0, 0, out, cyc, trips);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long h[4]; CK(hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost));
    const double ninstr = (double)trips * 32;
    printf("%-28s waves/WG %d: %.2f ns per instruction per wave; s_memtime ticks per instr %.3f\n", name, nthr_waves, ms * 1e6 / ninstr, (double)h[0] / ninstr);
    hipFree(out); hipFree(cyc); return 0;
}
int main() {
    for (int w : {4, 8}) {
        run<0>("v_exp_f32", 256, w); run<6>("v_exp_f16", 256, w); run<1>("v_cvt_pk_bf16_f32", 256, w); run<2>("v_max3_f32", 256, w);
        run<3>("v_pk_mul_f32", 256, w); run<4>("v_fma_f32", 256, w); run<5>("v_pk_fma_f32", 256, w); run<7>("v_exp_f32+v_fma_f32 pairs", 256, w);
        run<8>("v_add_u32", 256, w); run<9>("v_ldexp_f32", 256, w); run<10>("v_fract_f32", 256, w); run<11>("v_cvt_i32_f32", 256, w);
    }
    return 0;
}
