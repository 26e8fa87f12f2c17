// HBM-bound elementwise kernels of the FreeFine hot path (gfx950): masked CFG, DDIM inversion step, masked
// DDIM/DDPM control step, layout packing, channel concat, timestep embedding, transposes, casts.
#pragma once
#include "common.h"
#include "../../include/freefine_hip.h"

// eps = eu + (cfg * (ec - eu)) * mask[hw]        (/root/reference/src/demo/model.py:605-611, 418-424, 778-785)
// eps_u/eps_c/out: fp32 [rows, C, HW]; mask: float [HW] (already promoted exactly as torch would) or null (plain CFG, :608)
__global__ __launch_bounds__(256) void cfg_masked_kernel(const float* __restrict__ eu, const float* __restrict__ ec,
                                                         const float* __restrict__ mask, float cfg, float* __restrict__ out,
                                                         long n, int HW) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float d = cfg * (ec[i] - eu[i]);
        out[i] = eu[i] + (mask ? d * mask[i % HW] : d);
    }
}

// DDIM inversion step (model.py:109-132): pred_x0 = (x - c_bt*eps)/c_at ; x_next = c_an*pred_x0 + c_bn*eps
__global__ __launch_bounds__(256) void ddim_inv_step_kernel(const float* __restrict__ eps, const float* __restrict__ x,
                                                            float c_bt, float c_at, float c_an, float c_bn,
                                                            float* __restrict__ x_next, float* __restrict__ pred_x0, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float e = eps[i];
        const float p0 = (x[i] - c_bt * e) / c_at;
        x_next[i] = c_an * p0 + c_bn * e;
        if (pred_x0) pred_x0[i] = p0;
    }
}

// masked DDIM/DDPM control step (model.py:134-198).  Row b uses the mask iff row_masked[b]; other rows use mask==1,
// (1-mask)==0 (model.py:172-174).  m / om are float [HW] holding mask and (1 - mask) evaluated in the mask's OWN dtype
// on the host (uint8 wrap-around preserved).  Per-row coefficients: c_dirm[b] = sqrt(1-a_prev-std_b^2).
typedef ffn_ctrl_step_desc CtrlStepParams;
__global__ __launch_bounds__(256) void ddim_ctrl_step_kernel(const CtrlStepParams p) {
    const long n = (long)p.rows * p.CHW;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int b = (int)(i / p.CHW);
        const int hw = (int)(i % p.HW);
        const float e = p.eps[i];
        const float mk = p.row_masked[b] ? p.m[hw] : 1.f;
        const float omk = p.row_masked[b] ? p.om[hw] : 0.f;
        const float p0 = (p.x[i] - p.c_bt * e) / p.c_at;
        const float dirm = p.c_dirm[b] * e * mk;
        const float dir = p.c_dir * e * omk + dirm;
        float xp = p.c_ap * p0 + dir;
        if (p.noise) xp = xp + p.stdv[b] * p.noise[i] * mk;
        p.x_prev[i] = xp;
        if (p.pred_x0) p.pred_x0[i] = p0;
    }
}

// latents fp32 NCHW [Bsrc, Cl, HW] -> T NHWC [B, HW, CP] (channels >= Cl zero), row b reads src_row[b]
typedef ffn_pack_desc PackParams;
template <typename T>
__global__ __launch_bounds__(256) void pack_nchw_kernel(const PackParams p) {
    const long n = (long)p.B * p.HW * p.CP;
    T* dst = reinterpret_cast<T*>(p.dst);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % p.CP);
        const long t = i / p.CP;
        const int hw = (int)(t % p.HW), b = (int)(t / p.HW);
        float v = 0.f;
        if (c < p.Cl) v = p.src[((long)p.src_row[b] * p.Cl + c) * p.HW + hw];
        DT<T>::st(dst + i, v);
    }
}

// fp32 NHWC [B, HW, C] (row stride ld) -> fp32 NCHW [B, C, HW]
__global__ __launch_bounds__(256) void nhwc_to_nchw_f32_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                               int B, int HW, int C, int ld) {
    const long n = (long)B * C * HW;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int hw = (int)(i % HW);
        const long t = i / HW;
        const int c = (int)(t % C), b = (int)(t / C);
        dst[i] = src[((long)b * HW + hw) * ld + c];
    }
}

// out[r, 0:C1] = a[r, :], out[r, C1:C1+C2] = b[r, :]   (skip-connection concat of the UNet up path), 16-byte chunks.
// a == nullptr: the left block is already in place (its producer wrote it there with ldo = C1 + C2): only b is copied.
template <typename T>
__global__ __launch_bounds__(256) void concat_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out,
                                                     long rows, int C1, int C2) {
    constexpr int EPC = DT<T>::EPC;
    const int c1 = C1 / EPC, c2 = C2 / EPC;
    const int c0 = a ? 0 : c1, ct = c1 + c2 - c0;     // chunk columns written per row
    const long n = rows * ct;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long r = i / ct;
        const int c = c0 + (int)(i - r * ct);
        const u32x4 v = (c < c1) ? *reinterpret_cast<const u32x4*>(a + r * C1 + c * EPC)
                                 : *reinterpret_cast<const u32x4*>(b + r * C2 + (c - c1) * EPC);
        *reinterpret_cast<u32x4*>(out + r * (C1 + C2) + c * EPC) = v;
    }
}

// sinusoidal timestep embedding (diffusers Timesteps: flip_sin_to_cos -> [cos | sin]); t read from device memory so
// a captured hipGraph can be replayed with a new timestep.  freq[j] = exp(-ln(10000) * j / (half - shift)) from host.
template <typename T>
__global__ void timestep_embed_kernel(const float* __restrict__ t_dev, const float* __restrict__ freq, T* __restrict__ out,
                                      int B, int half, int flip) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 2 * half) return;
    const int j = i % (2 * half);
    const float t = *t_dev;
    const int jj = j < half ? j : j - half;
    const float a = t * freq[jj];
    const bool want_cos = flip ? (j < half) : (j >= half);
    DT<T>::st(out + i, want_cos ? cosf(a) : sinf(a));
}

// [B, R, C] -> [B, C, R] through a 32x32 LDS tile
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ src, T* __restrict__ dst, int R, int C,
                                                        int ld_src, int ld_dst) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? DT<T>::ld(src + ((long)b * R + r) * ld_src + c) : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (r < R && c < C) DT<T>::st(dst + ((long)b * C + c) * ld_dst + r, tile[tx][i]);
    }
}

template <typename TS, typename TD>
__global__ __launch_bounds__(256) void cast_kernel(const TS* __restrict__ src, TD* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) DT<TD>::st(dst + i, DT<TS>::ld(src + i));
}

// fp32 [rows][ld_src] (first C columns) -> the split-bf16 PAIR form the FFN_BF16X3 GEMMs read: bf16 [rows][2C], hi = bf16(x) (RNE),
// lo = bf16(x - hi), laid out by common.h pair_pos (C % 32 == 0: 128-byte blocks [hi(32) | lo(32)]; else the planes [hi(C) | lo(C)]).
// hi + lo carries 16-17 significant bits of x; 4 elements (16 B in, 2 x 8 B out) per thread.
__global__ __launch_bounds__(256) void split_pair_kernel(const float* __restrict__ src, bf16* __restrict__ dst, long rows, int C, int ld_src) {
    const int cq = C >> 2;
    const long n = rows * cq;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long r = i / cq;
        const int c = (int)(i - r * cq) << 2;
        const f32x4 x = *reinterpret_cast<const f32x4*>(src + r * ld_src + c);
        u32x2 hi, lo;
        hi[0] = pack_bf16x2(x[0], x[1]);
        hi[1] = pack_bf16x2(x[2], x[3]);
        lo[0] = pack_bf16x2(x[0] - __uint_as_float(hi[0] << 16), x[1] - __uint_as_float(hi[0] & 0xffff0000u));
        lo[1] = pack_bf16x2(x[2] - __uint_as_float(hi[1] << 16), x[3] - __uint_as_float(hi[1] & 0xffff0000u));
        bf16* q = dst + r * 2 * C + pair_pos(c, C);
        *reinterpret_cast<u32x2*>(q) = hi;
        *reinterpret_cast<u32x2*>(q + pair_lo(C)) = lo;
    }
}

// the same, 8 elements per thread (C % 8 == 0): 16-byte stores instead of 8-byte ones (round 6)
__global__ __launch_bounds__(256) void split_pair8_kernel(const float* __restrict__ src, bf16* __restrict__ dst, long rows, int C, int ld_src) {
    const int cq = C >> 3;
    const long n = rows * cq;
    const int lo_off = pair_lo(C);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long r = i / cq;
        const int c = (int)(i - r * cq) << 3;
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(src + r * ld_src + c), x1 = *reinterpret_cast<const f32x4*>(src + r * ld_src + c + 4);
        const float f[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        u32x4 hi, lo;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            hi[w] = pack_bf16x2(f[2 * w], f[2 * w + 1]);
            lo[w] = pack_bf16x2(f[2 * w] - __uint_as_float(hi[w] << 16), f[2 * w + 1] - __uint_as_float(hi[w] & 0xffff0000u));
        }
        bf16* q = dst + r * 2 * C + pair_pos(c, C);
        *reinterpret_cast<u32x4*>(q) = hi;
        *reinterpret_cast<u32x4*>(q + lo_off) = lo;
    }
}

// image pre/post for the VAE bracket (model.py:1282-1288, 270-280)
// uint8 HWC [B,HW,3] -> T NHWC [B,HW,CP]: v/127.5 - 1
template <typename T>
__global__ __launch_bounds__(256) void image_to_nhwc_kernel(const uint8_t* __restrict__ img, T* __restrict__ dst, long npix, int CP) {
    const long n = npix * CP;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % CP);
        const long px = i / CP;
        float v = 0.f;
        if (c < 3) v = (float)img[px * 3 + c] / 127.5f - 1.0f;
        DT<T>::st(dst + i, v);
    }
}
// decoder output T NHWC [B,HW,ld] -> fp32 NCHW [B,3,HW] in [0,1]: clamp(x/2+0.5)
template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_image_kernel(const T* __restrict__ src, float* __restrict__ dst, int B, int HW, int ld) {
    const long n = (long)B * 3 * HW;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int hw = (int)(i % HW);
        const long t = i / HW;
        const int c = (int)(t % 3), b = (int)(t / 3);
        float v = DT<T>::ld(src + ((long)b * HW + hw) * ld + c) / 2.f + 0.5f;
        dst[i] = fminf(fmaxf(v, 0.f), 1.f);
    }
}

// ---- depth front end ----------------------------------------------------------------------------------------
// y = max(a, 0) / y = a + b, four elements per thread-iteration (depth_anything/blocks.py:68, 137-139)
template <typename T, int OP>
__global__ __launch_bounds__(256) void eltwise_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float va[4], vb[4];
        load4(a + 4 * i, va);
        if (OP == FFN_ELT_ADD) {
            load4(b + 4 * i, vb);
#pragma unroll
            for (int r = 0; r < 4; ++r) va[r] += vb[r];
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) va[r] = fmaxf(va[r], 0.f);
        }
        store4(y + 4 * i, va);
    }
}

// bilinear resampling, align_corners = True, NHWC (torch.nn.functional.interpolate as called by depth_anything/blocks.py:147-149 and
// dpt.py:132, 165): source index = dst * (in - 1) / (out - 1) in fp32, the four taps blended as
// h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11) -- the order of ATen's upsample_bilinear2d.  Four channels per thread.
template <typename T, bool RELU>
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const T* __restrict__ x, T* __restrict__ y, long n4, int Hin, int Win, int Hout,
                                                              int Wout, int C) {
    const int c4 = C / 4;
    const float sh = Hout > 1 ? (float)(Hin - 1) / (float)(Hout - 1) : 0.f;
    const float sw = Wout > 1 ? (float)(Win - 1) / (float)(Wout - 1) : 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const int c = (int)(i % c4) * 4;
        long t = i / c4;
        const int ox = (int)(t % Wout);
        t /= Wout;
        const int oy = (int)(t % Hout);
        const long b = t / Hout;
        const float fy = sh * oy, fx = sw * ox;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < Hin - 1 ? 1 : 0), x1 = x0 + (x0 < Win - 1 ? 1 : 0);
        const float h1 = fy - y0, h0 = 1.f - h1, w1 = fx - x0, w0 = 1.f - w1;
        const T* base = x + b * Hin * Win * (long)C + c;
        float v00[4], v01[4], v10[4], v11[4], o[4];
        load4(base + ((long)y0 * Win + x0) * C, v00);
        load4(base + ((long)y0 * Win + x1) * C, v01);
        load4(base + ((long)y1 * Win + x0) * C, v10);
        load4(base + ((long)y1 * Win + x1) * C, v11);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            o[r] = h0 * (w0 * v00[r] + w1 * v01[r]) + h1 * (w0 * v10[r] + w1 * v11[r]);
            if (RELU) o[r] = fmaxf(o[r], 0.f);
        }
        store4(y + 4 * i, o);
    }
}
