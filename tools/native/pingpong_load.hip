// The igemm_pp phase in miniature: two waves per SIMD; one issues its MFMA chain (20 x 16x16x32 or 10 x 32x32x16 bf16: the same FLOPs)
// while the partner runs a LOAD section (NDMA LDS-DMA pieces from an L2-resident source, NREAD ds_read_b128, a counted vmcnt wait);
// roles swap at every barrier.  Cycles per phase.    ./pingpong_load
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lptr_t;
template <int SHAPE, int NDMA, int NREAD, int NMFMA16>
__global__ __launch_bounds__(512) void k(const char* src, float* out, long long* cyc, int trips) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int grp = threadIdx.x >> 8, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src) + (size_t)blockIdx.x * 262144, 0, 262144, 0x00020000);
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    f32x4 c4[8];
    f32x16 c16[2];
    for (int i = 0; i < 8; ++i) c4[i] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) c16[i][j] = 0.f;
    u32x4 fr[8];
    for (int i = 0; i < 8; ++i) fr[i] = u32x4{0, 0, 0, 0};
    const int voff = (lane >> 3) * 640 + (lane & 7) * 16;
    const int rd = (lane & 15) * 128 + (((lane >> 4) ^ (lane & 7)) << 4);
    int it = 0;
    auto mfma_block = [&]() {
        __builtin_amdgcn_s_setprio(1);
        if (SHAPE == 16) {
#pragma unroll
            for (int i = 0; i < NMFMA16; ++i) c4[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fr[i & 7]), b, c4[i & 7], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < NMFMA16 / 2; ++i) c16[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fr[i & 7]), b, c16[i & 1], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    auto load_block = [&]() {
#pragma unroll
        for (int i = 0; i < NREAD; ++i) fr[i & 7] = *reinterpret_cast<const u32x4*>(smem + ((it + i) & 15) * 2048 + (wave & 3) * 16384 + rd);
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            const int so = (((it * NDMA + i) * 8 + wave) * 5120) & 0x3ffff & ~127;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(smem + 65536 + (wave * 4 + (i & 3)) * 1024), 16, voff, so < 262144 - 5120 ? so : 0, 0, 0);
        }
        if (NDMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        ++it;
    };
    if (grp == 1) __builtin_amdgcn_s_barrier();
    const long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < trips; ++t) {
        __builtin_amdgcn_sched_barrier(0);
        load_block();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        mfma_block();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    }
    const long long t1 = __builtin_readcyclecounter();
    if (grp == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0;
    for (int i = 0; i < 8; ++i) s += c4[i][0] + __uint_as_float(fr[i][1]);
    s += c16[0][0] + c16[1][5];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int SHAPE, int NDMA, int NREAD, int NMFMA16> static int run(const char* src) {
    float* out; long long* cyc; CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&cyc, 256 * 8));
    const int trips = 2000, G = 32;                  // 32 workgroups: the 8 MB source stays in L2
    auto kern = k<SHAPE, NDMA, NREAD, NMFMA16>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipLaunchKernelGGL(kern, dim3(G), dim3(512), 131072, 0, src, out, cyc, trips); CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(kern, dim3(G), dim3(512), 131072, 0, src, out, cyc, trips); CK(hipDeviceSynchronize());
    long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("MFMA %2dx%2d (%2d per phase), load section %d LDS-DMA + %d ds_read_b128: %7.1f cycles per phase (MFMA pipe alone: %d)\n", SHAPE, SHAPE, SHAPE == 16 ? NMFMA16 : NMFMA16 / 2,
           NDMA, NREAD, (double)h / (2.0 * trips), NMFMA16 * 16);
    hipFree(out); hipFree(cyc);
    return 0;
}
int main() {
    char* src; CK(hipMalloc(&src, 32 * 262144)); CK(hipMemset(src, 1, 32 * 262144));
    run<16, 0, 0, 20>(src); run<32, 0, 0, 20>(src);
    run<16, 0, 7, 20>(src); run<32, 0, 7, 20>(src);
    run<16, 2, 0, 20>(src); run<32, 2, 0, 20>(src);
    run<16, 2, 7, 20>(src); run<32, 2, 7, 20>(src);
    run<16, 3, 7, 20>(src); run<32, 3, 7, 20>(src);
    run<16, 4, 13, 40>(src); run<32, 4, 13, 40>(src);
    return 0;
}
