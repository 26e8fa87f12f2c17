// Write-bandwidth probe: how fast can 256 CUs write a [M][N] bf16 matrix with (a) 16-byte-per-lane row-contiguous stores,
// (b) the MFMA C-layout pattern of the igemm epilogues (8 bytes per lane, one instruction = 16 rows x 32 bytes),
// (c) the same widened to 16 bytes per lane (16 rows x 64 bytes).   ./store_bw M N [reps] [workgroups]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// each workgroup (512 threads, 8 waves as 2 x 4) writes 256 x 320 tiles, persistent over tiles
template <int MODE>
__global__ __launch_bounds__(512) void wr(unsigned short* out, int M, int N) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr_ = wave >> 2, wc = wave & 3, l15 = lane & 15, g = lane >> 4;
    const int ntn = N / 320, ntiles = (M / 256) * ntn;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)((long)M * N * 2), 0x00020000);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int m0 = (t / ntn) * 256 + wr_ * 128, n0 = (t % ntn) * 320 + wc * 80;
        if (MODE == 0) {            // row-contiguous: the wave's 128 x 80 block as 160-byte rows, 10 lanes x 16 B per row
            for (int r = lane / 10; r < 128; r += 6) {
                if (lane < 60) { u32x4 v = {1u, 2u, 3u, (unsigned)t}; __builtin_amdgcn_raw_buffer_store_b128(v, rs, ((m0 + r) * N + n0) * 2 + (lane % 10) * 16, 0, 0); }
            }
        } else if (MODE == 1) {     // C layout, 8 B per lane
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 5; ++j) { u32x2 v = {1u, (unsigned)t}; __builtin_amdgcn_raw_buffer_store_b64(v, rs, (l15 * N + 4 * g) * 2, ((m0 + i * 16) * N + n0 + j * 16) * 2, 0); }
        } else if (MODE == 2) {     // widened: 16 B per lane, 16 rows x 64 B per instruction (+ one 8-byte column block)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int j = 0; j < 4; j += 2) { u32x4 v = {1u, 2u, 3u, (unsigned)t}; __builtin_amdgcn_raw_buffer_store_b128(v, rs, (l15 * N + 8 * g) * 2, ((m0 + i * 16) * N + n0 + j * 16) * 2, 0); }
                u32x2 v = {1u, (unsigned)t}; __builtin_amdgcn_raw_buffer_store_b64(v, rs, (l15 * N + 4 * g) * 2, ((m0 + i * 16) * N + n0 + 64) * 2, 0);
            }
        } else if (MODE >= 4) {     // lane-permuted C layouts over the wave's 128 x 80 block: RUN adjacent lanes write adjacent bytes of a row
            // MODE 4: 8 B/lane, 16 rows x 32 B per instruction (runs of 4 lanes)      MODE 5: 8 B/lane, 8 rows x 64 B (runs of 8)
            // MODE 6: 16 B/lane, 16 rows x 64 B (runs of 4)                           MODE 7: 16 B/lane, 8 rows x 128 B (runs of 8)
            // MODE 8: 8 B/lane, 4 rows x 128 B (runs of 16)
            constexpr int W = (MODE == 6 || MODE == 7) ? 16 : 8;                  // bytes per lane
            constexpr int RUN = MODE == 4 || MODE == 6 ? 4 : (MODE == 8 ? 16 : 8);
            constexpr int ROWS = 64 / RUN, RB = RUN * W;                          // rows and bytes per row of one instruction
            const int r = lane / RUN, c = lane % RUN;
            for (int rb = 0; rb < 128; rb += ROWS)
                for (int cb = 0; cb + RB <= 160; cb += RB) {
                    if (W == 16) { u32x4 v = {1u, 2u, 3u, (unsigned)t}; __builtin_amdgcn_raw_buffer_store_b128(v, rs, (r * N) * 2 + c * W, ((m0 + rb) * N + n0) * 2 + cb, 0); }
                    else { u32x2 v = {1u, (unsigned)t}; __builtin_amdgcn_raw_buffer_store_b64(v, rs, (r * N) * 2 + c * W, ((m0 + rb) * N + n0) * 2 + cb, 0); }
                }
        } else {                    // whole 640-byte rows of the TILE by one wave: 40 lanes x 16 B, 32 rows per wave
            for (int r = wave * 32; r < wave * 32 + 32; ++r)
                if (lane < 40) { u32x4 v = {1u, 2u, 3u, (unsigned)t}; __builtin_amdgcn_raw_buffer_store_b128(v, rs, (((t / ntn) * 256 + r) * N + (t % ntn) * 320) * 2 + lane * 16, 0, 0); }
        }
    }
}
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 196608, N = argc > 2 ? atoi(argv[2]) : 2560, reps = argc > 3 ? atoi(argv[3]) : 10, G = argc > 4 ? atoi(argv[4]) : 256;
    unsigned short* out; CK(hipMalloc(&out, (size_t)M * N * 2));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[9] = {"wave block rows, 16 B/lane", "C layout 8 B/lane (16 rows x 32 B)", "C layout widened 16 B/lane (16 rows x 64 B)", "tile rows 640 B, 16 B/lane", "permuted 8 B/lane 16 rows x 32 B (runs of 4)", "permuted 8 B/lane 8 rows x 64 B (runs of 8)", "permuted 16 B/lane 16 rows x 64 B (runs of 4)", "permuted 16 B/lane 8 rows x 128 B (runs of 8)", "permuted 8 B/lane 4 rows x 128 B (runs of 16)"};
    for (int mode = 0; mode < 9; ++mode) {
        for (int it = 0; it < 2; ++it) {
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; ++r) {
                if (mode == 0) hipLaunchKernelGGL(wr<0>, dim3(G), dim3(512), 0, 0, out, M, N);
                if (mode == 1) hipLaunchKernelGGL(wr<1>, dim3(G), dim3(512), 0, 0, out, M, N);
                if (mode == 2) hipLaunchKernelGGL(wr<2>, dim3(G), dim3(512), 0, 0, out, M, N);
                if (mode == 3) hipLaunchKernelGGL(wr<3>, dim3(G), dim3(512), 0, 0, out, M, N);
                if (mode == 4) hipLaunchKernelGGL(wr<4>, dim3(G), dim3(512), 0, 0, out, M, N);
                if (mode == 5) hipLaunchKernelGGL(wr<5>, dim3(G), dim3(512), 0, 0, out, M, N);
                if (mode == 6) hipLaunchKernelGGL(wr<6>, dim3(G), dim3(512), 0, 0, out, M, N);
                if (mode == 7) hipLaunchKernelGGL(wr<7>, dim3(G), dim3(512), 0, 0, out, M, N);
                if (mode == 8) hipLaunchKernelGGL(wr<8>, dim3(G), dim3(512), 0, 0, out, M, N);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (it) printf("M %d N %d grid %d  %-46s %8.1f us  %6.0f GB/s  %.1f B/clk/CU @2.4GHz\n", M, N, G, names[mode], ms * 1e3 / reps, (double)M * N * 2 / (ms * 1e-3 / reps) * 1e-9, (double)M * N * 2 / (ms * 1e-3 / reps) / G / 2.4e9);
        }
    }
    CK(hipMemset(out, 0, (size_t)M * N * 2));
    CK(hipEventRecord(e0)); for (int r = 0; r < reps; ++r) CK(hipMemsetAsync(out, 0, (size_t)M * N * 2, 0)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("hipMemsetAsync %8.1f us  %6.0f GB/s\n", ms * 1e3 / reps, (double)M * N * 2 / (ms * 1e-3 / reps) * 1e-9);
    return 0;
}
