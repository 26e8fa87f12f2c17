// conv3x3_n4_kernel: 3x3 / stride 1 / pad 1 convolution with FOUR output channels -- the UNet's conv_out (320 -> 4 at 64x64; reference call
// site: the diffusers UNet's conv_out behind conv_norm_out + SiLU, reached from /root/reference/src/utils/attention.py:205-214).
// On the MFMA tiles this layer wastes 31/32 of a 128-column tile: the generic split-bf16 kernel took 490-510 us per launch (1.1 % of the step)
// for 1.1 GFLOP of fp32 work.  Here it is a direct fp32 convolution on the vector ALU: a pixel per four lanes (each lane 4 of every 16 channels,
// so four lanes read 64 contiguous bytes of the pixel), weights [4][9][Cin] fp32 in LDS (broadcast reads), 16 FMAs per 16-byte activation load,
// the four lanes of a pixel summed by two cross-lane adds.  Exact fp32 products and fp32 accumulation in EVERY mode (fp32 or bf16 activations in,
// fp32 out): more accurate than the split-bf16 GEMM it replaces.  HBM / L2 bound: every activation line is read by nine output pixels.
#pragma once
#include "common.h"

template <typename TIN>
__global__ __launch_bounds__(256) void conv3x3_n4_kernel(const TIN* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                         float* __restrict__ out, int B, int H, int W, int Cin) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* wl = reinterpret_cast<float*>(smem);                 // [4][9][Cin]
    const int tid = threadIdx.x;
    for (int i = tid * 4; i < 36 * Cin; i += 1024) *reinterpret_cast<f32x4*>(wl + i) = *reinterpret_cast<const f32x4*>(w + i);
    __syncthreads();
    const long P = (long)blockIdx.x * 64 + (tid >> 2);
    const int q = tid & 3;
    const long npix = (long)B * H * W;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};                      // the four output channels of this lane's channel subset
    if (P < npix) {
        const int hw = H * W;
        const int b = (int)(P / hw), rem = (int)(P - (long)b * hw);
        const int y = rem / W, x0 = rem - y * W;
        const int nit = Cin >> 4;
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = y + tap / 3 - 1, xx = x0 + tap % 3 - 1;
            if ((unsigned)yy >= (unsigned)H || (unsigned)xx >= (unsigned)W) continue;
            const TIN* px = x + (((long)b * H + yy) * W + xx) * Cin + 4 * q;
            const float* wt = wl + tap * Cin + 4 * q;
#pragma unroll 4
            for (int it = 0; it < nit; ++it) {
                float v[4];
                load4(px + 16 * it, v);
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(wt + n * 9 * Cin + 16 * it);
                    acc[n] = fmaf(v[0], wv[0], fmaf(v[1], wv[1], fmaf(v[2], wv[2], fmaf(v[3], wv[3], acc[n]))));
                }
            }
        }
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        acc[n] += __shfl_xor(acc[n], 1);
        acc[n] += __shfl_xor(acc[n], 2);
    }
    if (P < npix && q == 0) {
        const f32x4 bv = bias ? *reinterpret_cast<const f32x4*>(bias) : f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(out + P * 4) = acc + bv;
    }
}
