// Common device helpers for the FreeFine gfx950 (CDNA4 / MI355X) kernels.
//
// Data model shared by every kernel in this library:
//   * activations live in HBM as [B, H*W, C] ("NHWC", channel-contiguous) so that the conv
//     blocks and the transformer blocks of the SD UNet read and write the SAME buffers with no
//     transposes in between (the reference, src/utils/attention.py:246, 297, transposes NCHW<->NLC
//     around every attention call);
//   * the element type T is either float (parity mode: exact fp32 MFMA, fp32 storage) or bf16
//     (fast mode: bf16 MFMA operands, fp32 accumulate, bf16 storage);
//   * all matrix products go through 16x16 MFMA tiles.  A "chunk" is 16 bytes of the contraction
//     (K) dimension = 4 floats or 8 bf16; one MFMA k-substep consumes a 64-byte K slab per row, i.e.
//     4 chunks, chunk g supplied by lane group g = lane>>4.  With that convention the LDS byte
//     geometry of a tile is identical for both element types and only DT<T>::mma differs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

struct bf16 {
    uint16_t v;
};

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// round-to-nearest-even; a plain cast keeps NaN a NaN (v_cvt_pk_bf16_f32 on gfx950).
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(uint16_t, h);
}
// two floats -> one dword of two bf16 (lo in bits 0..15): ONE v_cvt_pk_bf16_f32
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const bf16x2_t h = __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t);
    return __builtin_bit_cast(uint32_t, h);
}

template <typename T>
struct DT;

template <>
struct DT<float> {
    static constexpr int EPC = 4;  // elements per 16-byte chunk
    static constexpr int ID = 0;
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
    // one 64-byte K slab (4 chunks x 4 floats): four exact-fp32 16x16x4 MFMAs; MFMA i consumes element i
    // of every lane group's chunk, so the hardware k index g maps to K element 4g+i for A and B alike.
    __device__ static __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[0]), __uint_as_float(b[0]), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[1]), __uint_as_float(b[1]), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[2]), __uint_as_float(b[2]), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[3]), __uint_as_float(b[3]), c, 0, 0, 0);
    }
    // unpack a chunk to floats / pack floats to a chunk (used by fused prologues)
    __device__ static __forceinline__ void unpack(const u32x4& c, float* f) {
        for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(c[i]);
    }
    __device__ static __forceinline__ u32x4 pack(const float* f) {
        u32x4 c;
        for (int i = 0; i < 4; ++i) c[i] = __float_as_uint(f[i]);
        return c;
    }
};

template <>
struct DT<bf16> {
    static constexpr int EPC = 8;
    static constexpr int ID = 1;
    __device__ static __forceinline__ float ld(const bf16* p) { return bf16_to_f32(p->v); }
    __device__ static __forceinline__ void st(bf16* p, float v) { p->v = f32_to_bf16(v); }
    __device__ static __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c,
                                                    0, 0, 0);
    }
    __device__ static __forceinline__ void unpack(const u32x4& c, float* f) {
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(c[i] << 16);
            f[2 * i + 1] = __uint_as_float(c[i] & 0xffff0000u);
        }
    }
    __device__ static __forceinline__ u32x4 pack(const float* f) {
        u32x4 c;
        for (int i = 0; i < 4; ++i) c[i] = pack_bf16x2(f[2 * i], f[2 * i + 1]);
        return c;
    }
};

// fp8 (OCP e4m3) operands in the SAME byte geometry as bf16: a 16-byte chunk holds 16 values, so a 128-byte K row holds 128 of them and
// one chunk pair (A, B) is TWO 16x16x32 MFMAs (bytes 0-7, bytes 8-15) where bf16 needs one -- twice the contraction per staged byte at the
// bf16 MFMA rate.  Any assignment of chunk bytes to k indices is fine as long as A and B agree, which they do by construction.
__device__ __forceinline__ void mma_fp8(const u32x4& a, const u32x4& b, f32x4& c) {
    const long a0 = (long)(((uint64_t)a[1] << 32) | a[0]), a1 = (long)(((uint64_t)a[3] << 32) | a[2]);
    const long b0 = (long)(((uint64_t)b[1] << 32) | b[0]), b1 = (long)(((uint64_t)b[3] << 32) | b[2]);
    c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a0, b0, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a1, b1, c, 0, 0, 0);
}
// four floats -> four e4m3 bytes (saturating at +-448: v_cvt_pk_fp8_f32 itself does not clamp)
__device__ __forceinline__ uint32_t pack_fp8x4(float a, float b, float c, float d) {
    const float lim = 448.0f;
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(a, -lim), lim), fminf(fmaxf(b, -lim), lim), w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(c, -lim), lim), fminf(fmaxf(d, -lim), lim), w, true);
    return (uint32_t)w;
}

// store 4 consecutive elements of type T from 4 floats (16 B for float, 8 B for bf16)
__device__ __forceinline__ void store4(float* p, const float* v) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
}
__device__ __forceinline__ void store4(bf16* p, const float* v) {
    u32x2 w;
    w[0] = pack_bf16x2(v[0], v[1]);
    w[1] = pack_bf16x2(v[2], v[3]);
    *reinterpret_cast<u32x2*>(p) = w;
}
__device__ __forceinline__ void load4(const float* p, float* v) {
    f32x4 w = *reinterpret_cast<const f32x4*>(p);
    v[0] = w[0]; v[1] = w[1]; v[2] = w[2]; v[3] = w[3];
}
__device__ __forceinline__ void load4(const bf16* p, float* v) {
    u32x2 w = *reinterpret_cast<const u32x2*>(p);
    v[0] = __uint_as_float(w[0] << 16);
    v[1] = __uint_as_float(w[0] & 0xffff0000u);
    v[2] = __uint_as_float(w[1] << 16);
    v[3] = __uint_as_float(w[1] & 0xffff0000u);
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_exact(float x) { return x / (1.0f + expf(-x)); }
// SiLU where the result is rounded to bf16 (2^-9) or fp8 anyway: one v_exp_f32, one v_rcp_f32 (~1 ulp each) and two multiplies instead of
// libm's expf and an IEEE division (~25 instructions per value: gn_apply_kernel<bf16, SiLU> was VALU-bound on them -- 56.8 us for the
// 126 MB in / 126 MB out that layernorm_kernel moves in 33 us).  Saturates cleanly: x -> -inf gives x * 0, x -> +inf gives x * 1.
__device__ __forceinline__ float silu_fast(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * x));
}
template <typename T>
__device__ __forceinline__ float silu_for(float x) {
    if constexpr (sizeof(T) == 2) return silu_fast(x);
    else return silu_exact(x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7): one v_rcp, one v_exp and six FMAs instead of libm's branchy ~30-instruction
// erff.  Used where the result is rounded to bf16 (2^-9) anyway: the GEGLU epilogue of the K = 320 feed-forward GEMM spent more
// issue slots on erff than its five K stages spent on MFMAs.
__device__ __forceinline__ float gelu_erf_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-z * z * 1.44269504088896340736f);
    const float erf_abs = fmaf(-poly * t, e, 1.0f);          // erf(|x|/sqrt2)
    const float cdf = 0.5f + copysignf(0.5f * erf_abs, x);
    return x * cdf;
}
// Two GELUs at once for the GEGLU epilogue of the bf16 GEMMs (the result is rounded to bf16, 2^-9): erf by Abramowitz-Stegun
// 7.1.25 (three coefficients, |error| <= 2.5e-5) on float2 values, so that everything but the two v_rcp / v_exp runs as packed f32
// instructions (v_pk_fma_f32 / v_pk_mul_f32: two lanes' worth of work per issue slot -- the epilogue has no MFMA beside it to disturb).
// gelu_erf_fast cost ~17 instructions per value; the K = 320 feed-forward GEMM spent as long in its GEGLU epilogue as in its K loop.
__device__ __forceinline__ f32x2_t gelu_erf_fast2(f32x2_t x) {
    const f32x2_t z = f32x2_t{fabsf(x[0]), fabsf(x[1])} * 0.70710678118654752440f;
    f32x2_t t = z * 0.47047f + 1.0f;
    t = f32x2_t{__builtin_amdgcn_rcpf(t[0]), __builtin_amdgcn_rcpf(t[1])};
    f32x2_t poly = t * 0.7478556f + (-0.0958798f);
    poly = poly * t + 0.3480242f;
    poly = poly * t;
    const f32x2_t zz = z * z * (-1.44269504088896340736f);
    const f32x2_t e = f32x2_t{__builtin_amdgcn_exp2f(zz[0]), __builtin_amdgcn_exp2f(zz[1])};
    const f32x2_t half_erf = poly * e * (-0.5f) + 0.5f;                     // 0.5 * erf(|x| / sqrt 2)
    const f32x2_t cdf = f32x2_t{copysignf(half_erf[0], x[0]), copysignf(half_erf[1], x[1])} + 0.5f;
    return x * cdf;
}
// gelu_erf_fast (Abramowitz-Stegun 7.1.26, |error| <= 1.5e-7: accurate enough for fp32 results) on two values at once -- the GEGLU epilogue of the
// split-bf16 GEMMs (round 4): everything but the two v_rcp / v_exp as packed f32 instructions, same arithmetic per value as the scalar form.
__device__ __forceinline__ f32x2_t gelu_erf26_2(f32x2_t x) {
    const f32x2_t z = f32x2_t{fabsf(x[0]), fabsf(x[1])} * 0.70710678118654752440f;
    f32x2_t t = z * 0.3275911f + 1.0f;
    t = f32x2_t{__builtin_amdgcn_rcpf(t[0]), __builtin_amdgcn_rcpf(t[1])};
    f32x2_t poly = t * 1.061405429f + (-1.453152027f);
    poly = poly * t + 1.421413741f;
    poly = poly * t + (-0.284496736f);
    poly = poly * t + 0.254829592f;
    const f32x2_t zz = z * z * (-1.44269504088896340736f);
    const f32x2_t e = f32x2_t{__builtin_amdgcn_exp2f(zz[0]), __builtin_amdgcn_exp2f(zz[1])};
    const f32x2_t half_erf = (poly * t) * e * (-0.5f) + 0.5f;               // 0.5 * erf(|x| / sqrt 2)
    const f32x2_t cdf = f32x2_t{copysignf(half_erf[0], x[0]), copysignf(half_erf[1], x[1])} + 0.5f;
    return x * cdf;
}
template <typename T>
__device__ __forceinline__ float gelu_for(float x) {
    if constexpr (sizeof(T) == 2) return gelu_erf_fast(x);
    else return gelu_erf(x);
}

// The split-bf16 PAIR form of a row of C fp32 values (FFN_BF16X3 A operands; include/freefine_hip.h): bf16 [2C].  C % 32 == 0 -> BLOCKED: every 32
// columns are one 128-byte block [hi(32) | lo(32)], so that a 32-deep K stage of the GEMM is one whole 128-byte line holding both planes
// (column c: hi at 64 (c / 32) + c % 32, lo 32 elements behind it); otherwise one block of C columns = the planes [hi(C) | lo(C)].
__device__ __forceinline__ int pair_pos(int c, int C) { return (C & 31) ? c : ((c >> 5) << 6) + (c & 31); }
__device__ __forceinline__ int pair_lo(int C) { return (C & 31) ? C : 32; }

// Bijective XCD-aware remap of a 1-D grid: blocks b and b+8 share an XCD (round-robin dispatch), so give
// every XCD a contiguous run of logical tiles (neighbouring tiles share operand panels -> L2 hits).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}
