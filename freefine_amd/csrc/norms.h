// GroupNorm / LayerNorm / row-softmax for [B, HW, C] (channel-contiguous) activations, gfx950.
// HBM-bound kernels: 16-byte vector accesses, fp32 statistics (final group reduction in fp64).
//
// GroupNorm(32 groups) is split into (1) per-channel partial sums over pixel chunks, (2) a tiny finalize
// that turns them into per-(batch, channel) scale/shift, (3) an elementwise apply with optional SiLU.
// Replaces torch.nn.GroupNorm + SiLU inside diffusers ResnetBlock2D / Transformer2DModel / conv_norm_out
// (reached from /root/reference/src/utils/attention.py:105-214).
#pragma once
#include "common.h"

// PAIR output (split-bf16 mode, T = float): the normalised row is written in the bf16 pair form an FFN_BF16X3 GEMM reads as its A operand
// (hi = bf16(v), lo = bf16(v - hi); blocked layout: common.h pair_pos) instead of fp32 -- same bytes, and the separate ffn_split_pair pass disappears.
__device__ __forceinline__ void store_pair4(bf16* yp, long row, int C, int c, const float* f) {
    u32x2 hi, lo;
    hi[0] = pack_bf16x2(f[0], f[1]);
    hi[1] = pack_bf16x2(f[2], f[3]);
    lo[0] = pack_bf16x2(f[0] - __uint_as_float(hi[0] << 16), f[1] - __uint_as_float(hi[0] & 0xffff0000u));
    lo[1] = pack_bf16x2(f[2] - __uint_as_float(hi[1] << 16), f[3] - __uint_as_float(hi[1] & 0xffff0000u));
    bf16* q = yp + row * 2 * C + pair_pos(c, C);
    *reinterpret_cast<u32x2*>(q) = hi;
    *reinterpret_cast<u32x2*>(q + pair_lo(C)) = lo;
}

// ---- (1) partial sums: grid (nchunk, B), 256 threads ------------------------------------------------------
// partial[b][chunk][c][2] = (sum, sumsq) over the chunk's pixels
template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(const T* __restrict__ x, float* __restrict__ partial, int HW,
                                                         int C, int pix_per_chunk) {
    constexpr int EPC = DT<T>::EPC;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ls = reinterpret_cast<float*>(smem);  // [ppi][C][2] per-thread-row partials (round 2: no LDS atomics -- 16 contended atomic
                                                 // adds per thread made this HBM-bound kernel run at 2.7 TB/s)
    const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int cch = C / EPC;                 // 16-byte chunks per pixel
    const int tid = threadIdx.x;
    const int p0 = chunk * pix_per_chunk;
    const int p1 = min(HW, p0 + pix_per_chunk);
    float* dst = partial + ((long)b * nchunk + chunk) * 2 * C;
    for (int cbase = 0; cbase < cch; cbase += 256) {
        const int cols = min(cch - cbase, 256);  // chunk columns handled concurrently in this sweep
        const int ppi = max(1, 256 / cols);       // pixel rows of threads
        const int cc = cbase + tid % cols, pp = tid / cols;
        const bool active = pp < ppi;
        float s[EPC], ss[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) s[e] = ss[e] = 0.f;
        if (active) {
            // eight pixels in flight per thread: a single dependent load per iteration made this kernel latency bound
            const T* src = x + ((long)b * HW) * C + cc * EPC;
            int px = p0 + pp;
            for (; px + 7 * ppi < p1; px += 8 * ppi) {
                u32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const u32x4*>(src + (long)(px + u * ppi) * C);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    float f[EPC];
                    DT<T>::unpack(v[u], f);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        s[e] += f[e];
                        ss[e] += f[e] * f[e];
                    }
                }
            }
            for (; px < p1; px += ppi) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(src + (long)px * C);
                float f[EPC];
                DT<T>::unpack(v, f);
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    s[e] += f[e];
                    ss[e] += f[e] * f[e];
                }
            }
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                ls[(pp * cols + (cc - cbase)) * 2 * EPC + 2 * e] = s[e];
                ls[(pp * cols + (cc - cbase)) * 2 * EPC + 2 * e + 1] = ss[e];
            }
        }
        __syncthreads();
        // fixed-order sum over the pixel rows of threads (deterministic)
        for (int i = tid; i < cols * 2 * EPC; i += 256) {
            float acc = 0.f;
            for (int r = 0; r < ppi; ++r) acc += ls[r * cols * 2 * EPC + i];
            dst[cbase * 2 * EPC + i] = acc;      // [c][2] with c = (cbase + column) * EPC + e: same interleaving as ls
        }
        __syncthreads();
    }
}

// ---- (2) finalize: grid (G, B), 64 threads: scale[b][c] = rstd*gamma[c], shift[b][c] = beta[c]-mean*rstd*gamma[c]
__global__ __launch_bounds__(64) void gn_finalize_kernel(const float* __restrict__ partial, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ scale,
                                                         float* __restrict__ shift, int HW, int C, int G, int nchunk,
                                                         float eps) {
    const int g = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const int cg = C / G;
    double s = 0.0, ss = 0.0;
    for (int i = lane; i < nchunk * cg; i += 64) {
        const int ch = i / cg, c = g * cg + (i - ch * cg);
        const float* src = partial + (((long)b * nchunk + ch) * C + c) * 2;
        s += (double)src[0];
        ss += (double)src[1];
    }
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_xor(s, off);
        ss += __shfl_xor(ss, off);
    }
    const double n = (double)HW * cg;
    const double mean = s / n;
    double var = ss / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float fmean = (float)mean;
    for (int i = lane; i < cg; i += 64) {
        const int c = g * cg + i;
        const float sc = rstd * gamma[c];
        scale[(long)b * C + c] = sc;
        shift[(long)b * C + c] = beta[c] - fmean * sc;
    }
}

// ---- (3) apply: y = act(x*scale[b][c] + shift[b][c]) ------------------------------------------------------
template <typename T, bool SILU, bool PAIR = false>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       long nchunks_total, int HW, int C) {
    constexpr int EPC = DT<T>::EPC;
    const int cch = C / EPC;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nchunks_total; i += (long)gridDim.x * 256) {
        const long pix = i / cch;
        const int cc = (int)(i - pix * cch);
        const int b = (int)(pix / HW);
        const u32x4 v = *reinterpret_cast<const u32x4*>(x + i * EPC);
        float f[EPC];
        DT<T>::unpack(v, f);
        const float* sc = scale + (long)b * C + cc * EPC;
        const float* sh = shift + (long)b * C + cc * EPC;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            float t = f[e] * sc[e] + sh[e];
            if (SILU) t = silu_for<T>(t);          // bf16 activations: the fast form (common.h); fp32 (parity / split-bf16 modes): exact
            f[e] = t;
        }
        if constexpr (PAIR) store_pair4(reinterpret_cast<bf16*>(y), pix, C, cc * EPC, f);
        else *reinterpret_cast<u32x4*>(y + i * EPC) = DT<T>::pack(f);
    }
}

// ---- (3p) apply with PAIR output, 8 channels per thread (round 6): two 16-byte loads, two 16-byte stores (hi / lo).  The 4-channel form above
// stores 8 bytes per lane (measured on gfx950: 8-byte accesses run at 0.54-0.70 of the 16-byte rate) and, for T = float, evaluates SiLU with libm's
// expf + an IEEE division (~30 instructions per value: the kernel was VALU-bound, 25 % slower per byte than the statistics pass).  The pair form
// carries 16-17 significant bits, so SiLU is x * rcp(1 + exp2(-x log2 e)) here (v_exp_f32 / v_rcp_f32: ~1 ulp each, 2e-7 relative).
// C % 8 == 0 (a thread's 8 channels never straddle a 32-column block of the blocked layout).
// RAW (round 6, ffn_groupnorm_pair_raw): the same pass also writes the pair form of x ITSELF to yraw -- the operand of the ResBlock's 1x1 shortcut GEMM, which
// otherwise costs a ffn_split_pair pass (a second read of x) beside this one.
template <bool SILU, bool RAW = false>
__global__ __launch_bounds__(256) void gn_apply_pair8_kernel(const float* __restrict__ x, bf16* __restrict__ y, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, long n8_total, int HW, int C, bf16* __restrict__ yraw = nullptr) {
    const int cch = C / 8;
    const int lo_off = pair_lo(C);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8_total; i += (long)gridDim.x * 256) {
        const long pix = i / cch;
        const int c = (int)(i - pix * cch) * 8;
        const int b = (int)(pix / HW);
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(x + pix * C + c), v1 = *reinterpret_cast<const f32x4*>(x + pix * C + c + 4);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(scale + (long)b * C + c), s1 = *reinterpret_cast<const f32x4*>(scale + (long)b * C + c + 4);
        const f32x4 h0 = *reinterpret_cast<const f32x4*>(shift + (long)b * C + c), h1 = *reinterpret_cast<const f32x4*>(shift + (long)b * C + c + 4);
        float f[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f[e] = v0[e] * s0[e] + h0[e];
            f[4 + e] = v1[e] * s1[e] + h1[e];
        }
        if (SILU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = silu_fast(f[e]);
        }
        u32x4 hi, lo;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            hi[w] = pack_bf16x2(f[2 * w], f[2 * w + 1]);
            lo[w] = pack_bf16x2(f[2 * w] - __uint_as_float(hi[w] << 16), f[2 * w + 1] - __uint_as_float(hi[w] & 0xffff0000u));
        }
        bf16* q = y + pix * 2 * C + pair_pos(c, C);
        *reinterpret_cast<u32x4*>(q) = hi;
        *reinterpret_cast<u32x4*>(q + lo_off) = lo;
        if (RAW) {
            const float r[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                hi[w] = pack_bf16x2(r[2 * w], r[2 * w + 1]);
                lo[w] = pack_bf16x2(r[2 * w] - __uint_as_float(hi[w] << 16), r[2 * w + 1] - __uint_as_float(hi[w] & 0xffff0000u));
            }
            bf16* qr = yraw + pix * 2 * C + pair_pos(c, C);
            *reinterpret_cast<u32x4*>(qr) = hi;
            *reinterpret_cast<u32x4*>(qr + lo_off) = lo;
        }
    }
}

// ---- (3b) apply with fp8 output (FFN_FP8 convolutions): y8[pixel][Cp] = e4m3(act(x * scale + shift) * qs), channels C .. Cp-1 = 0.
// Cp (a multiple of 128) is the padded channel count the fp8 ping-pong conv needs; qs is the power-of-two activation scale.
template <bool SILU>
__global__ __launch_bounds__(256) void gn_apply_f8_kernel(const bf16* __restrict__ x, uint8_t* __restrict__ y, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, long nchunks_total, int HW, int C, int Cp, float qs) {
    // one thread = 16 channels of one pixel: two 16-byte loads, ONE 16-byte store (8-byte stores ran this pass at 2.5 TB/s of real traffic)
    const int cch = Cp / 16;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nchunks_total; i += (long)gridDim.x * 256) {
        const long pix = i / cch;
        const int c0 = (int)(i - pix * cch) * 16;
        u32x4 o = u32x4{0u, 0u, 0u, 0u};
        if (c0 < C) {                                        // C % 16 == 0 is checked by the launcher
            const int b = (int)(pix / HW);
            const u32x4 v0 = *reinterpret_cast<const u32x4*>(x + pix * C + c0);
            const u32x4 v1 = *reinterpret_cast<const u32x4*>(x + pix * C + c0 + 8);
            float f[16];
            DT<bf16>::unpack(v0, f);
            DT<bf16>::unpack(v1, f + 8);
            const float* sc = scale + (long)b * C + c0;
            const float* sh = shift + (long)b * C + c0;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float t = f[e] * sc[e] + sh[e];
                if (SILU) t = silu_fast(t);
                f[e] = t * qs;
            }
#pragma unroll
            for (int w = 0; w < 4; ++w) o[w] = pack_fp8x4(f[4 * w], f[4 * w + 1], f[4 * w + 2], f[4 * w + 3]);
        }
        *reinterpret_cast<u32x4*>(y + pix * Cp + c0) = o;
    }
}

// ---- fused GroupNorm for slices that one workgroup can own: grid (G, B), 1024 threads ----------------------------------
// pass 1: sum / sumsq over the (batch, group) slice [HW][cg] (element pairs, 4 or 8 bytes per load; the slice is L2 resident),
// block reduction (fp32 per thread, fp64 across threads), pass 2: y = act((x - mean) * rstd * gamma + beta).
template <typename T, bool SILU, bool PAIR = false>
__global__ __launch_bounds__(1024) void gn_fused_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int HW, int C, int G, float eps) {
    __shared__ double red[2][16];
    __shared__ float stat[2];
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cg = C / G, hp = cg / 2;                      // channel pairs per pixel in this group
    const long base = (long)b * HW * C + (long)g * cg;
    const int npair = HW * hp;
    float s = 0.f, ss = 0.f;
    for (int i = tid; i < npair; i += 1024) {
        const int px = i / hp, c2 = i - px * hp;
        const T* src = x + base + (long)px * C + 2 * c2;
        const float a = DT<T>::ld(src), c = DT<T>::ld(src + 1);
        s += a + c;
        ss += a * a + c * c;
    }
    double ds = s, dss = ss;
    for (int off = 32; off > 0; off >>= 1) {
        ds += __shfl_xor(ds, off);
        dss += __shfl_xor(dss, off);
    }
    if (lane == 0) {
        red[0][wave] = ds;
        red[1][wave] = dss;
    }
    __syncthreads();
    if (tid == 0) {
        double t = 0, tt = 0;
        for (int w = 0; w < 16; ++w) {
            t += red[0][w];
            tt += red[1][w];
        }
        const double n = (double)HW * cg, mean = t / n;
        double var = tt / n - mean * mean;
        if (var < 0) var = 0;
        stat[0] = (float)mean;
        stat[1] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const float mean = stat[0], rstd = stat[1];
    for (int i = tid; i < npair; i += 1024) {
        const int px = i / hp, c2 = i - px * hp;
        const int c = g * cg + 2 * c2;
        const long off = base + (long)px * C + 2 * c2;
        const float sc0 = rstd * gamma[c], sc1 = rstd * gamma[c + 1];
        float a = DT<T>::ld(x + off) * sc0 + (beta[c] - mean * sc0);
        float d = DT<T>::ld(x + off + 1) * sc1 + (beta[c + 1] - mean * sc1);
        if (SILU) {
            a = silu_for<T>(a);
            d = silu_for<T>(d);
        }
        if constexpr (PAIR) {
            bf16* yp = reinterpret_cast<bf16*>(y);
            const long prow = ((long)b * HW + px) * 2 * C;
            const uint32_t hi = pack_bf16x2(a, d);
            *reinterpret_cast<uint32_t*>(yp + prow + pair_pos(c, C)) = hi;
            *reinterpret_cast<uint32_t*>(yp + prow + pair_pos(c, C) + pair_lo(C)) = pack_bf16x2(a - __uint_as_float(hi << 16), d - __uint_as_float(hi & 0xffff0000u));
        } else {
            DT<T>::st(y + off, a);
            DT<T>::st(y + off + 1, d);
        }
    }
}

// ---- LayerNorm over C per row; one wave per row, two-pass in registers --------------------------------------
template <typename T, int MAXCH, bool PAIR = false>  // MAXCH = max 16-byte chunks per lane
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int M, int C, float eps) {
    constexpr int EPC = DT<T>::EPC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= M) return;
    const int cch = C / EPC;
    float f[MAXCH][EPC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int cc = lane + 64 * i;
        if (cc < cch) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(x + (long)row * C + cc * EPC);
            DT<T>::unpack(v, f[i]);
#pragma unroll
            for (int e = 0; e < EPC; ++e) s += f[i][e];
        }
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float mean = s / (float)C;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int cc = lane + 64 * i;
        if (cc < cch) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float d = f[i][e] - mean;
                ss += d * d;
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    const float rstd = 1.0f / sqrtf(ss / (float)C + eps);
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int cc = lane + 64 * i;
        if (cc < cch) {
            float o[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const int c = cc * EPC + e;
                o[e] = (f[i][e] - mean) * rstd * gamma[c] + beta[c];
            }
            if constexpr (PAIR) store_pair4(reinterpret_cast<bf16*>(y), row, C, cc * EPC, o);
            else *reinterpret_cast<u32x4*>(y + (long)row * C + cc * EPC) = DT<T>::pack(o);
        }
    }
}

// ---- row softmax (VAE mid-block single-head attention path: scores materialised once, 2 calls per image) ----
// in: fp32 or T scores [M][N] (already scaled), out: T probabilities.  One 256-thread block per row.
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const T* __restrict__ x, T* __restrict__ y, int N, float scale) {
    __shared__ float red[8];
    const long row = blockIdx.x;
    const T* xr = x + row * N;
    T* yr = y + row * N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float m = -1e30f;
    for (int i = tid; i < N; i += 256) m = fmaxf(m, DT<T>::ld(xr + i) * scale);
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s = 0.f;
    for (int i = tid; i < N; i += 256) s += expf(DT<T>::ld(xr + i) * scale - m);
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) red[4 + wave] = s;
    __syncthreads();
    s = red[4] + red[5] + red[6] + red[7];
    const float inv = 1.0f / s;
    for (int i = tid; i < N; i += 256) DT<T>::st(yr + i, expf(DT<T>::ld(xr + i) * scale - m) * inv);
}
