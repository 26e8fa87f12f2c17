// Probe (round 6): read bandwidth of the GroupNorm statistics pass (gn_partial_kernel, norms.h) against its launch geometry.
//   hipcc --offload-arch=gfx950 -O3 -o build/native/gn_partial_probe tools/native/gn_partial_probe.hip ; ./gn_partial_probe
// x: fp32 [B][HW][C] (random values; PROBE_ZEROS=1: all zero -- the first run of this probe, which read 20-40 % faster); partial[b][chunk][C][2].  Variants: pixels per chunk (workgroups per row), loads in flight per thread, and a flat
// "one workgroup = whole 16-byte columns x pixel stripes" reader.  Prints us and GB/s of x read.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int U>
__global__ __launch_bounds__(256) void partial_v(const float* __restrict__ x, float* __restrict__ partial, int HW, int C, int ppc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ls = reinterpret_cast<float*>(smem);
    const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int cch = C / 4, tid = threadIdx.x;
    const int p0 = chunk * ppc, p1 = min(HW, p0 + ppc);
    float* dst = partial + ((long)b * nchunk + chunk) * 2 * C;
    for (int cbase = 0; cbase < cch; cbase += 256) {
        const int cols = min(cch - cbase, 256), ppi = max(1, 256 / cols);
        const int cc = cbase + tid % cols, pp = tid / cols;
        const bool active = pp < ppi;
        float s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
        if (active) {
            const float* src = x + ((long)b * HW) * C + cc * 4;
            int px = p0 + pp;
            for (; px + (U - 1) * ppi < p1; px += U * ppi) {
                f32x4 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (long)(px + u * ppi) * C));
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { s[e] += v[u][e]; ss[e] += v[u][e] * v[u][e]; }
            }
            for (; px < p1; px += ppi) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(src + (long)px * C);
#pragma unroll
                for (int e = 0; e < 4; ++e) { s[e] += v[e]; ss[e] += v[e] * v[e]; }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { ls[(pp * cols + (cc - cbase)) * 8 + 2 * e] = s[e]; ls[(pp * cols + (cc - cbase)) * 8 + 2 * e + 1] = ss[e]; }
        }
        __syncthreads();
        for (int i = tid; i < cols * 8; i += 256) {
            float acc = 0.f;
            for (int r = 0; r < ppi; ++r) acc += ls[r * cols * 8 + i];
            dst[cbase * 8 + i] = acc;
        }
        __syncthreads();
    }
}

// plain variant without the nontemporal hint (the shipped kernel's loads)
template <int U>
__global__ __launch_bounds__(256) void partial_p(const float* __restrict__ x, float* __restrict__ partial, int HW, int C, int ppc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ls = reinterpret_cast<float*>(smem);
    const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int cch = C / 4, tid = threadIdx.x;
    const int p0 = chunk * ppc, p1 = min(HW, p0 + ppc);
    float* dst = partial + ((long)b * nchunk + chunk) * 2 * C;
    for (int cbase = 0; cbase < cch; cbase += 256) {
        const int cols = min(cch - cbase, 256), ppi = max(1, 256 / cols);
        const int cc = cbase + tid % cols, pp = tid / cols;
        const bool active = pp < ppi;
        float s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
        if (active) {
            const float* src = x + ((long)b * HW) * C + cc * 4;
            int px = p0 + pp;
            for (; px + (U - 1) * ppi < p1; px += U * ppi) {
                f32x4 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const f32x4*>(src + (long)(px + u * ppi) * C);
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { s[e] += v[u][e]; ss[e] += v[u][e] * v[u][e]; }
            }
            for (; px < p1; px += ppi) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(src + (long)px * C);
#pragma unroll
                for (int e = 0; e < 4; ++e) { s[e] += v[e]; ss[e] += v[e] * v[e]; }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { ls[(pp * cols + (cc - cbase)) * 8 + 2 * e] = s[e]; ls[(pp * cols + (cc - cbase)) * 8 + 2 * e + 1] = ss[e]; }
        }
        __syncthreads();
        for (int i = tid; i < cols * 8; i += 256) {
            float acc = 0.f;
            for (int r = 0; r < ppi; ++r) acc += ls[r * cols * 8 + i];
            dst[cbase * 8 + i] = acc;
        }
        __syncthreads();
    }
}

// flat reader: the chunk's bytes are one contiguous run (ppc pixels x C floats); thread t reads 16-byte element t, t + 256, ... (fully coalesced
// 4 KB per workgroup instruction); a thread's channel phase advances by (256 * 4) % C per step, so it accumulates into a small per-thread table
// indexed by step % period -- only where C / 4 divides 256 * k for a small k.  Here: sums over everything only (a bandwidth ceiling, not a usable kernel).
template <int U>
__global__ __launch_bounds__(256) void ceiling(const float* __restrict__ x, float* __restrict__ partial, long n16_per_wg) {
    const f32x4* src = reinterpret_cast<const f32x4*>(x) + (long)blockIdx.x * n16_per_wg;
    f32x4 s = {0, 0, 0, 0}, ss = {0, 0, 0, 0};
    long i = threadIdx.x;
    for (; i + (U - 1) * 256 < n16_per_wg; i += U * 256) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) { s += v[u]; ss += v[u] * v[u]; }
    }
    for (; i < n16_per_wg; i += 256) { const f32x4 v = src[i]; s += v; ss += v * v; }
    const float r = s[0] + s[1] + s[2] + s[3] + ss[0] + ss[1] + ss[2] + ss[3];
    if (r == 12345.678f) partial[blockIdx.x * 256 + threadIdx.x] = r;
}

__global__ void fill_random(float* x, long n, unsigned seed) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        x[i] = ((int)(h & 0xffffff) - 0x800000) * (1.0f / 0x400000);      // uniform in [-2, 2): every mantissa bit toggles
    }
}

template <typename F>
static float time_us(F f, int reps = 20) {
    for (int i = 0; i < 3; ++i) f();
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / reps;
}

int main() {
    const int shapes[][3] = {{72, 4096, 320}, {72, 4096, 640}, {72, 1024, 640}, {72, 1024, 1280}, {72, 256, 1280}, {48, 4096, 320}};
    for (auto& sh : shapes) {
        const int B = sh[0], HW = sh[1], C = sh[2];
        const long n = (long)B * HW * C;
        float *x, *partial;
        CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&partial, 64 << 20));
        if (getenv("PROBE_ZEROS")) { CK(hipMemset(x, 0, n * 4)); } else { hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, x, n, 12345u); CK(hipDeviceSynchronize()); }
        printf("B=%d HW=%d C=%d (%.0f MB)\n", B, HW, C, n * 4e-6);
        const int cols = C / 4 < 256 ? C / 4 : 256;
        const int lds = (256 / cols > 0 ? 256 / cols : 1) * cols * 8 * 4;
        for (int ppc : {32, 64, 128, 256, 512, 1024}) {
            if (ppc > HW) continue;
            const int nchunk = HW / ppc;
            float t4 = time_us([&] { hipLaunchKernelGGL(partial_p<4>, dim3(nchunk, B), dim3(256), lds, 0, x, partial, HW, C, ppc); });
            float t8 = time_us([&] { hipLaunchKernelGGL(partial_p<8>, dim3(nchunk, B), dim3(256), lds, 0, x, partial, HW, C, ppc); });
            float t16 = time_us([&] { hipLaunchKernelGGL(partial_p<16>, dim3(nchunk, B), dim3(256), lds, 0, x, partial, HW, C, ppc); });
            float n8 = time_us([&] { hipLaunchKernelGGL(partial_v<8>, dim3(nchunk, B), dim3(256), lds, 0, x, partial, HW, C, ppc); });
            float n16 = time_us([&] { hipLaunchKernelGGL(partial_v<16>, dim3(nchunk, B), dim3(256), lds, 0, x, partial, HW, C, ppc); });
            printf("  pixels/chunk %4d (%5d workgroups): U=4 %.1f us %.0f GB/s | U=8 %.1f us %.0f GB/s | U=16 %.1f us %.0f GB/s | nontemporal U=8 %.1f us %.0f GB/s  U=16 %.1f us %.0f GB/s\n", ppc, nchunk * B,
                   t4, n * 4e-3 / t4, t8, n * 4e-3 / t8, t16, n * 4e-3 / t16, n8, n * 4e-3 / n8, n16, n * 4e-3 / n16);
        }
        for (int wgs : {256, 512, 1024, 2048, 4096, 8192}) {
            const long per = n / 4 / wgs;
            float c8 = time_us([&] { hipLaunchKernelGGL(ceiling<8>, dim3(wgs), dim3(256), 0, 0, x, partial, per); });
            float c16 = time_us([&] { hipLaunchKernelGGL(ceiling<16>, dim3(wgs), dim3(256), 0, 0, x, partial, per); });
            printf("  flat contiguous reader, %4d workgroups: U=8 %.1f us %.0f GB/s | U=16 %.1f us %.0f GB/s\n", wgs, c8, n * 4e-3 / c8, c16, n * 4e-3 / c16);
        }
        CK(hipFree(x)); CK(hipFree(partial));
    }
    return 0;
}
