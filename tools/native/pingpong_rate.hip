// Two waves per SIMD in ping-pong (one issues an MFMA chain while its partner issues the softmax VALU mix, roles swap at every barrier):
// cycles per phase with the chain as 36 x v_mfma_f32_16x16x32_bf16 or as 18 x v_mfma_f32_32x32x16_bf16 (the same FLOPs).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o build/pingpong_rate tools/native/pingpong_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int SHAPE, int NEXP, int NPLAIN>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int trips) {
    const int grp = threadIdx.x >> 8;                 // waves 0-3 / 4-7
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    f32x4 c4[8];
    f32x16 c16[2];
    for (int i = 0; i < 8; ++i) c4[i] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) c16[i][j] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
    auto mfma_block = [&]() {
        __builtin_amdgcn_s_setprio(1);
        if (SHAPE == 16) {
#pragma unroll
            for (int i = 0; i < 36; ++i) c4[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4[i & 7], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 18; ++i) c16[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c16[i & 1], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    auto valu_block = [&]() {
#pragma unroll
        for (int i = 0; i < NEXP; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i & 7]));
#pragma unroll
        for (int i = 0; i < NPLAIN; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i & 7]) : "v"(v[(i + 1) & 7]), "v"(v[(i + 3) & 7]));
    };
    if (grp == 1) __builtin_amdgcn_s_barrier();
    const long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < trips; ++t) {
        valu_block();
        __builtin_amdgcn_s_barrier();
        mfma_block();
        __builtin_amdgcn_s_barrier();
    }
    const long long t1 = __builtin_readcyclecounter();
    if (grp == 0) __builtin_amdgcn_s_barrier();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i] + c4[i][0];
    s += c16[0][0] + c16[1][5];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int SHAPE, int NEXP, int NPLAIN> static int run() {
    float* out; long long* cyc; CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&cyc, 256 * 8));
    const int trips = 2000;
    hipLaunchKernelGGL((k<SHAPE, NEXP, NPLAIN>), dim3(256), dim3(512), 0, 0, out, cyc, trips); CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<SHAPE, NEXP, NPLAIN>), dim3(256), dim3(512), 0, 0, out, cyc, trips);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    // each trip = two phases (both groups run one MFMA block and one VALU block); MFMA FLOPs per phase per SIMD: 36 * 16384
    const double flops = 256.0 * 4 * 2 * trips * 36 * 16384.0;
    printf("MFMA %2dx%2d  %2d v_exp + %2d v_fma per VALU block: %7.1f cycles per phase (s_memtime), %6.0f TFLOP/s\n", SHAPE, SHAPE, NEXP, NPLAIN, (double)h / (2.0 * trips),
           flops / (ms * 1e-3) * 1e-12);
    return 0;
}
int main() {
    run<16, 0, 0>(); run<32, 0, 0>();
    run<16, 32, 32>(); run<32, 32, 32>();
    run<16, 32, 64>(); run<32, 32, 64>();
    run<16, 32, 96>(); run<32, 32, 96>();
    return 0;
}
