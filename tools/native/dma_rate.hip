// Per-CU rate of LDS-DMA loads (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction = 8 rows x 128 B) from an L2-resident
// source, 8 waves per CU, and the same with C-layout stores mixed in.   ./dma_rate [workgroups]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef __attribute__((address_space(3))) void* lptr_t;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void k(const char* src, char* dst, int bytes_per_wg, int iters, int row_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src) + (size_t)blockIdx.x * bytes_per_wg, 0, bytes_per_wg, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst + (size_t)blockIdx.x * (16 * 1024 * 1024), 0, 16 * 1024 * 1024, 0x00020000);
    // lane -> row (lane >> 3) of 8 rows, 16-byte chunk (lane & 7); rows row_stride apart (128 = contiguous 1 KiB; 640 = activation rows)
    const int voff = (lane >> 3) * row_stride + (lane & 7) * 16;
    const int l15 = lane & 15, g = lane >> 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int so = ((it * 8 + u) * 8 + wave) * 8 * row_stride % (bytes_per_wg - 8 * row_stride);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(smem + (wave * 8 + u) * 1024), 16, voff, so & ~127, 0, 0);
            if (MODE == 1 && (u & 1)) {   // one C-layout store (16 rows x 32 B, rows 5120 B apart) per two loads
                u32x2 v = {1u, (unsigned)it};
                __builtin_amdgcn_raw_buffer_store_b64(v, rd, (l15 * 2560 + 4 * g) * 2, ((it * 4 + (u >> 1)) * 8 + wave) * 16 * 5120 % (15 * 1024 * 1024), 0);
            }
            if (MODE == 2 && (u & 3) == 3) {   // one lane-permuted 16-byte store (16 rows x 64 B, runs of 4 lanes) per four loads: the same bytes per load
                u32x4 v = {1u, 2u, 3u, (unsigned)it};
                __builtin_amdgcn_raw_buffer_store_b128(v, rd, (lane >> 2) * 5120 + (lane & 3) * 16, ((it * 2 + (u >> 2)) * 8 + wave) * 16 * 5120 % (15 * 1024 * 1024), 0);
            }
            if (MODE == 3 && (u & 3) == 3) {   // contiguous 1 KiB store per four loads
                u32x4 v = {1u, 2u, 3u, (unsigned)it};
                __builtin_amdgcn_raw_buffer_store_b128(v, rd, lane * 16, ((it * 2 + (u >> 2)) * 8 + wave) * 1024 % (15 * 1024 * 1024), 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
int main(int argc, char** argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 256;
    const int bytes_per_wg = 256 * 1024;     // per-WG source window: stays in L2
    char *src, *dst; CK(hipMalloc(&src, (size_t)G * bytes_per_wg)); CK(hipMalloc(&dst, (size_t)G * 16 * 1024 * 1024)); CK(hipMemset(src, 1, (size_t)G * bytes_per_wg));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    for (int mode = 0; mode < 4; ++mode)
        for (int rs : {128, 640, 2560}) {
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(G), dim3(512), 65536, 0, src, dst, bytes_per_wg, iters, rs);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(G), dim3(512), 65536, 0, src, dst, bytes_per_wg, iters, rs);
                else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(G), dim3(512), 65536, 0, src, dst, bytes_per_wg, iters, rs);
                else hipLaunchKernelGGL(k<3>, dim3(G), dim3(512), 65536, 0, src, dst, bytes_per_wg, iters, rs);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double bytes = (double)iters * 8 * 8 * 1024;     // loaded per workgroup
                if (rep) printf("grid %d %s row stride %4d: %.1f us, %.1f B/clk/CU loaded @2.4GHz (%.0f GB/s aggregate)%s\n", G, mode == 0 ? "loads only  " : mode == 1 ? "loads+C-layout stores" : mode == 2 ? "loads+permuted 16 B stores" : "loads+contiguous stores", rs, ms * 1e3,
                                bytes / (ms * 1e-3) / 2.4e9, bytes * G / (ms * 1e-3) * 1e-9, mode ? "; stores = 1/4 of the load bytes" : "");
            }
        }
    return 0;
}
