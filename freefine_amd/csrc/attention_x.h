// xattn_kernel: cross attention against a SHORT key sequence (the 77 text tokens of the UNet's attn2: Sk <= 96), bf16, head dim 64,
// one unmasked pass.  Replaces attn_kernel<bf16, 64, 2, 64, 2, false> for that case (reference: plain attention,
// /root/reference/src/utils/attention.py:394-404 with encoder_hidden_states as K / V).
//
// Why its own kernel.  With 77 keys the arithmetic is nothing (20 GFLOP at the 64x64 level) and the launch is bound by reading Q and
// writing the output once (252 MB there); the flash-style kernel spends it on per-workgroup prologues (K / V^T tiles through LDS,
// two 64-key tiles for 77 keys, 128 queries per workgroup): 151 us = 1.7 TB/s.  Here a WAVE keeps the whole K and V^T of its
// (batch row, head) in registers (88 VGPRs at 77 keys) and streams query blocks of 32 past them: no LDS, no barrier, Q of the next
// block in flight while the current one is multiplied.
//
// Formulation (same as attention.h): S^T = K . Q^T by 16x16x32 MFMA, a lane owns query l15 and keys 16 f + 4 g + r of key fragment
// f; the exponentiated fragments, packed to bf16, are the B operand of O^T = V^T . P^T as they stand when the MFMA k index of lane
// group g is read as keys {32 s + 4 g + e, 32 s + 16 + 4 g + e} -- the V^T fragments are loaded in that order (two 8-byte loads).
// The output leaves through the lane permutation of the igemm epilogue (runs of four lanes write 64 contiguous bytes).
#pragma once
#include "attention.h"

template <int NKF>                                    // 16-key fragments held (Sk <= 16 NKF, Sk >= 1)
__global__ __launch_bounds__(256) void xattn_kernel(const AttnParams p, int waves_per_pair, int blocks_per_wave) {
    typedef bf16 T;
    constexpr int NKS = (NKF + 1) / 2;                // 32-key steps of the PV product
    constexpr int QF = 2;                             // 16-query fragments per block
    const int lane = threadIdx.x & 63;
    const int gw = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int pair = gw / waves_per_pair, chunk = gw - pair * waves_per_pair;
    if (pair >= p.Bo * p.heads) return;
    const int b = pair / p.heads, head = pair - b * p.heads;
    const int l15 = lane & 15, g = lane >> 4;
    const AttnEntry& e = p.e[b];
    const float w = e.w_const + (e.w_slope != 0.f ? e.w_slope * *p.w_dev : 0.f);
    const int C = p.heads * 64;
    constexpr int OOB = (int)0x80000000;

    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.q), 0, 0x7ffff000, 0x00020000);
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.k), 0, 0x7ffff000, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.vt), 0, 0x7ffff000, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, 0x7ffff000, 0x00020000);

    // ---- K and V^T of (kv_row, head): registers for the life of the wave ----
    u32x4 kf[NKF][2], vf[4][NKS];
#pragma unroll
    for (int f = 0; f < NKF; ++f) {
        const int key = 16 * f + l15;
        const int vo = key < p.Sk ? (key * p.ldk + 8 * g) * 2 : OOB;                      // keys past Sk: zeros (masked below)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            kf[f][ks] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rk, vo, ((e.kv_row * p.Sk) * p.ldk + head * 64 + 32 * ks) * 2, 0));
    }
#pragma unroll
    for (int df = 0; df < 4; ++df) {
        const int vo = (l15 * p.ldvt + 4 * g) * 2;
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            // keys inside the row (< ldvt; the ABI keeps the padding up to ldvt finite) are read, keys past it are zeros: both only ever
            // meet P = 0 there
            const int so = ((e.kv_row * C + head * 64 + 16 * df) * p.ldvt + 32 * s) * 2;
            const u32x2 lo = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rv, 32 * s + 4 * g < p.ldvt ? vo : OOB, so, 0));
            const u32x2 hi = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rv, 32 * s + 16 + 4 * g < p.ldvt ? vo : OOB, so + 32, 0));
            vf[df][s] = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
    }

    const float c = p.scale * 1.44269504088896340736f;
    const int qvo = (l15 * p.ldq + 8 * g) * 2;
    const int blk0 = chunk * blocks_per_wave;
    const int nblk_total = (p.S + 31) / 32;
    int nblk = nblk_total - blk0;
    if (nblk > blocks_per_wave) nblk = blocks_per_wave;
    if (nblk <= 0) return;

    auto load_q = [&](int blk, u32x4 (&qf)[QF][2]) {
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            const int vo = blk * 32 + 16 * f + l15 < p.S ? qvo : OOB;                      // rows past S: zeros (never stored)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                qf[f][ks] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rq, vo, ((e.q_row * p.S + blk * 32 + 16 * f) * p.ldq + head * 64 + 32 * ks) * 2, 0));
        }
    };
    // output lane permutation (see igemm_p8.h, store_rows): lane 4 r + cc stores the 16-byte piece cc of row r
    const int pr = lane >> 2, pg = lane & 3;
    const int paddr16 = (pr + 16 * (((pg & 1) << 1) | (pg >> 1))) * 4;
    const int ovo = pr * p.ldo * 2 + pg * 16;

    u32x4 qa[QF][2], qb[QF][2];
    load_q(blk0, qa);
    for (int ib = 0; ib < nblk; ++ib) {
        const int blk = blk0 + ib;
        if (ib + 1 < nblk) load_q(blk + 1, qb);       // next block's Q flies under this block's arithmetic
        // ---- S^T = K . Q^T ----
        f32x4 st[NKF][QF];
#pragma unroll
        for (int f = 0; f < NKF; ++f)
#pragma unroll
            for (int qf = 0; qf < QF; ++qf) {
                st[f][qf] = f32x4{0.f, 0.f, 0.f, 0.f};
                DT<T>::mma(kf[f][0], qa[qf][0], st[f][qf]);
                DT<T>::mma(kf[f][1], qa[qf][1], st[f][qf]);
            }
        // keys past Sk
#pragma unroll
        for (int f = 0; f < NKF; ++f)
            if (16 * (f + 1) > p.Sk) {                // (scalar condition: whole fragments inside Sk cost nothing)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (16 * f + 4 * g + r >= p.Sk) {
#pragma unroll
                        for (int qf = 0; qf < QF; ++qf) st[f][qf][r] = -__builtin_inff();
                    }
            }
        u32x4 pf[QF][NKS];
        float inv_l[QF];
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) {
            float m = st[0][qf][0];
#pragma unroll
            for (int f = 0; f < NKF; ++f) {
                if (f) m = att_max(m, st[f][qf][0]);
                m = att_max3(m, st[f][qf][1], st[f][qf][2]);
                m = att_max(m, st[f][qf][3]);
            }
            m = att_max_groups(m);
            const float mc = -m * c;
            float l = 0.f;
#pragma unroll
            for (int f = 0; f < NKF; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(st[f][qf][r], c, mc));
                    st[f][qf][r] = pv;
                    l += pv;
                }
            {   // sum over the four key groups of the query
                auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(l), __float_as_uint(l), false, false);
                l = __uint_as_float(a[0]) + __uint_as_float(a[1]);
                auto d2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(l), __float_as_uint(l), false, false);
                l = __uint_as_float(d2[0]) + __uint_as_float(d2[1]);
            }
            inv_l[qf] = w / l;
#pragma unroll
            for (int s = 0; s < NKS; ++s) {
                pf[qf][s][0] = pack_bf16x2(st[2 * s][qf][0], st[2 * s][qf][1]);
                pf[qf][s][1] = pack_bf16x2(st[2 * s][qf][2], st[2 * s][qf][3]);
                if (2 * s + 1 < NKF) {
                    pf[qf][s][2] = pack_bf16x2(st[2 * s + 1][qf][0], st[2 * s + 1][qf][1]);
                    pf[qf][s][3] = pack_bf16x2(st[2 * s + 1][qf][2], st[2 * s + 1][qf][3]);
                } else {
                    pf[qf][s][2] = 0u;
                    pf[qf][s][3] = 0u;
                }
            }
        }
        // ---- O^T = V^T . P^T, normalise, store ----
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) {
            unsigned wo[4][2];
#pragma unroll
            for (int df = 0; df < 4; ++df) {
                f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < NKS; ++s) DT<T>::mma(vf[df][s], pf[qf][s], o);
                o *= inv_l[qf];
                wo[df][0] = pack_bf16x2(o[0], o[1]);
                wo[df][1] = pack_bf16x2(o[2], o[3]);
            }
            const int q0 = blk * 32 + 16 * qf;
            const int vo = q0 + pr < p.S ? ovo : OOB;
#pragma unroll
            for (int dp = 0; dp < 4; dp += 2) {
                const auto s0 = __builtin_amdgcn_permlane16_swap(wo[dp][0], wo[dp + 1][0], false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(wo[dp][1], wo[dp + 1][1], false, false);
                u32x4 v;
                v[0] = (unsigned)__builtin_amdgcn_ds_bpermute(paddr16, (int)s0[0]);
                v[1] = (unsigned)__builtin_amdgcn_ds_bpermute(paddr16, (int)s1[0]);
                v[2] = (unsigned)__builtin_amdgcn_ds_bpermute(paddr16, (int)s0[1]);
                v[3] = (unsigned)__builtin_amdgcn_ds_bpermute(paddr16, (int)s1[1]);
                __builtin_amdgcn_raw_buffer_store_b128(v, ro, vo, ((b * p.S + q0) * p.ldo + head * 64 + 16 * dp) * 2, 0);
            }
        }
        if (ib + 1 < nblk) {
#pragma unroll
            for (int f = 0; f < QF; ++f)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) qa[f][ks] = qb[f][ks];
        }
    }
}

// xattn_mp_kernel: the same against short key sequences for launches with SEVERAL passes and / or per-query weights -- the guided
// pass's local cross-attention (modulate_local_cross_attn, /root/reference/src/utils/attention.py:1360-1393: the edit row is
// f * Attn(q_edit, text_edit) + (1 - f) * Attn(q_src, text_src) with the per-query blend f of the local edit region).
// out[b, q] = sum_p w_p(b) wq_p[q] softmax(Q[qrow_p(b), q] K[kvrow_p(b)]^T) V[kvrow_p(b)].  K and V^T of all passes of a
// (row, head) do not fit one wave's registers, so a workgroup (4 waves) keeps them in LDS as FRAGMENT IMAGES (the 16 bytes a lane
// would hold, lane-major: 1 KiB per fragment, conflict-free by construction), 22 KiB per pass at 77 keys, and its waves stream query
// blocks of 32 past them, pass by pass, summing the weighted pass results in registers.
template <int NKF>
__global__ __launch_bounds__(256) void xattn_mp_kernel(const AttnParams p, int wgs_per_pair, int blocks_per_wave) {
    typedef bf16 T;
    constexpr int NKS = (NKF + 1) / 2;
    constexpr int QF = 2;
    constexpr int NFR = NKF * 2 + 4 * NKS;            // fragments per pass (K: NKF x 2 d-steps, V^T: 4 d-fragments x NKS key steps)
    constexpr int OOB = (int)0x80000000;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pair = blockIdx.x / wgs_per_pair, chunk = (blockIdx.x - pair * wgs_per_pair) * 4 + wave;
    const int b = pair / p.heads, head = pair - b * p.heads;
    const int l15 = lane & 15, g = lane >> 4;
    const int C = p.heads * 64;

    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.q), 0, 0x7ffff000, 0x00020000);
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.k), 0, 0x7ffff000, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.vt), 0, 0x7ffff000, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, 0x7ffff000, 0x00020000);

    // ---- fragment images of every active pass -> LDS (fragment i of pass ps at (ps * NFR + i) KiB; the waves share the fragments) ----
    for (int ps = 0; ps < p.npass; ++ps) {
        const AttnEntry& e = p.e[ps * ATT_MAXB + b];
        if (e.w_const == 0.f && e.w_slope == 0.f) continue;
        for (int i = wave; i < NFR; i += 4) {
            u32x4 v;
            if (i < 2 * NKF) {
                const int f = i >> 1, ks = i & 1;
                const int key = 16 * f + l15;
                v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rk, key < p.Sk ? (key * p.ldk + 8 * g) * 2 : OOB, ((e.kv_row * p.Sk) * p.ldk + head * 64 + 32 * ks) * 2, 0));
            } else {
                const int j = i - 2 * NKF, df = j / NKS, s = j - df * NKS;
                const int vo = (l15 * p.ldvt + 4 * g) * 2;
                const int so = ((e.kv_row * C + head * 64 + 16 * df) * p.ldvt + 32 * s) * 2;
                const u32x2 lo = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rv, 32 * s + 4 * g < p.ldvt ? vo : OOB, so, 0));
                const u32x2 hi = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rv, 32 * s + 16 + 4 * g < p.ldvt ? vo : OOB, so + 32, 0));
                v = u32x4{lo[0], lo[1], hi[0], hi[1]};
            }
            *reinterpret_cast<u32x4*>(smem + (ps * NFR + i) * 1024 + lane * 16) = v;
        }
    }
    __syncthreads();

    const float c = p.scale * 1.44269504088896340736f;
    const int qvo = (l15 * p.ldq + 8 * g) * 2;
    const int blk0 = chunk * blocks_per_wave;
    const int nblk_total = (p.S + 31) / 32;
    int nblk = nblk_total - blk0;
    if (nblk > blocks_per_wave) nblk = blocks_per_wave;
    const int pr = lane >> 2, pg = lane & 3;
    const int paddr16 = (pr + 16 * (((pg & 1) << 1) | (pg >> 1))) * 4;
    const int ovo = pr * p.ldo * 2 + pg * 16;

    for (int ib = 0; ib < nblk; ++ib) {
        const int blk = blk0 + ib;
        f32x4 acc[4][QF];
#pragma unroll
        for (int df = 0; df < 4; ++df)
#pragma unroll
            for (int qf = 0; qf < QF; ++qf) acc[df][qf] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ps = 0; ps < p.npass; ++ps) {
            const AttnEntry& e = p.e[ps * ATT_MAXB + b];
            if (e.w_const == 0.f && e.w_slope == 0.f) continue;
            const float w = e.w_const + (e.w_slope != 0.f ? e.w_slope * *p.w_dev : 0.f);
            const char* img = smem + ps * NFR * 1024 + lane * 16;
            u32x4 qa[QF][2];
            float wql[QF];
#pragma unroll
            for (int f = 0; f < QF; ++f) {
                const int qrow = blk * 32 + 16 * f + l15;
                const int vo = qrow < p.S ? qvo : OOB;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    qa[f][ks] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rq, vo, ((e.q_row * p.S + blk * 32 + 16 * f) * p.ldq + head * 64 + 32 * ks) * 2, 0));
                wql[f] = (e.wq && qrow < p.S) ? e.wq[qrow] : 1.f;
            }
            f32x4 st[NKF][QF];
#pragma unroll
            for (int f = 0; f < NKF; ++f) {
                const u32x4 k0 = *reinterpret_cast<const u32x4*>(img + (2 * f) * 1024), k1 = *reinterpret_cast<const u32x4*>(img + (2 * f + 1) * 1024);
#pragma unroll
                for (int qf = 0; qf < QF; ++qf) {
                    st[f][qf] = f32x4{0.f, 0.f, 0.f, 0.f};
                    DT<T>::mma(k0, qa[qf][0], st[f][qf]);
                    DT<T>::mma(k1, qa[qf][1], st[f][qf]);
                }
            }
#pragma unroll
            for (int f = 0; f < NKF; ++f)
                if (16 * (f + 1) > p.Sk) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (16 * f + 4 * g + r >= p.Sk) {
#pragma unroll
                            for (int qf = 0; qf < QF; ++qf) st[f][qf][r] = -__builtin_inff();
                        }
                }
            u32x4 pf[QF][NKS];
            float sc[QF];
#pragma unroll
            for (int qf = 0; qf < QF; ++qf) {
                float m = st[0][qf][0];
#pragma unroll
                for (int f = 0; f < NKF; ++f) {
                    if (f) m = att_max(m, st[f][qf][0]);
                    m = att_max3(m, st[f][qf][1], st[f][qf][2]);
                    m = att_max(m, st[f][qf][3]);
                }
                m = att_max_groups(m);
                const float mc = -m * c;
                float l = 0.f;
#pragma unroll
                for (int f = 0; f < NKF; ++f)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(st[f][qf][r], c, mc));
                        st[f][qf][r] = pv;
                        l += pv;
                    }
                {
                    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(l), __float_as_uint(l), false, false);
                    l = __uint_as_float(a[0]) + __uint_as_float(a[1]);
                    auto d2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(l), __float_as_uint(l), false, false);
                    l = __uint_as_float(d2[0]) + __uint_as_float(d2[1]);
                }
                sc[qf] = w * wql[qf] / l;
#pragma unroll
                for (int s = 0; s < NKS; ++s) {
                    pf[qf][s][0] = pack_bf16x2(st[2 * s][qf][0], st[2 * s][qf][1]);
                    pf[qf][s][1] = pack_bf16x2(st[2 * s][qf][2], st[2 * s][qf][3]);
                    if (2 * s + 1 < NKF) {
                        pf[qf][s][2] = pack_bf16x2(st[2 * s + 1][qf][0], st[2 * s + 1][qf][1]);
                        pf[qf][s][3] = pack_bf16x2(st[2 * s + 1][qf][2], st[2 * s + 1][qf][3]);
                    } else {
                        pf[qf][s][2] = 0u;
                        pf[qf][s][3] = 0u;
                    }
                }
            }
#pragma unroll
            for (int df = 0; df < 4; ++df) {
                u32x4 vfr[NKS];
#pragma unroll
                for (int s = 0; s < NKS; ++s) vfr[s] = *reinterpret_cast<const u32x4*>(img + (2 * NKF + df * NKS + s) * 1024);
#pragma unroll
                for (int qf = 0; qf < QF; ++qf) {
                    f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < NKS; ++s) DT<T>::mma(vfr[s], pf[qf][s], o);
                    acc[df][qf] += o * sc[qf];
                }
            }
        }
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) {
            const int q0 = blk * 32 + 16 * qf;
            const int vo = q0 + pr < p.S ? ovo : OOB;
#pragma unroll
            for (int dp = 0; dp < 4; dp += 2) {
                const auto s0 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(acc[dp][qf][0], acc[dp][qf][1]), pack_bf16x2(acc[dp + 1][qf][0], acc[dp + 1][qf][1]), false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(acc[dp][qf][2], acc[dp][qf][3]), pack_bf16x2(acc[dp + 1][qf][2], acc[dp + 1][qf][3]), false, false);
                u32x4 v;
                v[0] = (unsigned)__builtin_amdgcn_ds_bpermute(paddr16, (int)s0[0]);
                v[1] = (unsigned)__builtin_amdgcn_ds_bpermute(paddr16, (int)s1[0]);
                v[2] = (unsigned)__builtin_amdgcn_ds_bpermute(paddr16, (int)s0[1]);
                v[3] = (unsigned)__builtin_amdgcn_ds_bpermute(paddr16, (int)s1[1]);
                __builtin_amdgcn_raw_buffer_store_b128(v, ro, vo, ((b * p.S + q0) * p.ldo + head * 64 + 16 * dp) * 2, 0);
            }
        }
    }
}
