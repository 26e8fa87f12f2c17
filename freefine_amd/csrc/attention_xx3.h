// xattn_x3_kernel: the short-key cross attention of attention_x.h (the 77 text tokens of the UNet's attn2: Sk <= 96, head dim 64, no key masks) in
// SPLIT-BF16 arithmetic (FFN_BF16X3): q / k / v^T are fp32 in HBM, every operand of both products is carried as hi = bf16(x), lo = bf16(x - hi),
//     S^T = K_lo.Q_hi^T + K_hi.Q_lo^T + K_hi.Q_hi^T          O^T = V^T_lo.P_hi^T + V^T_hi.P_lo^T + V^T_hi.P_hi^T
// on the bf16 16x16x32 MFMA with fp32 accumulation, fp32 softmax.  Until round 4 these launches ran on the generic attn_x3_kernel<false>: two
// 64-key tiles for 77 keys, a workgroup barrier per tile, 128 ... 256 queries per workgroup prologue -- 46 TFLOP/s AND 1.2 TB/s, bound by neither
// (profiles/r4_bench_event_table_1stream.txt).  Structure = xattn_mp_kernel's: a workgroup (4 waves) splits K and V^T of every active pass ONCE
// into FRAGMENT IMAGES in LDS (the 16 bytes a lane holds, lane-major: 1 KiB per fragment, hi and lo images: 44 KiB per pass at 77 keys,
// conflict-free by construction) and its waves stream query blocks of 32 past them, pass by pass, summing the weighted pass results in
// registers: no barrier after the prologue.  Covers the single plain pass and the guided pass's two-pass local form with per-query blend
// weights (modulate_local_cross_attn, /root/reference/src/utils/attention.py:1360-1393).  Output fp32, or (out_pair) the blocked pair form
// the to_out projection's split-bf16 GEMM reads.
#pragma once
#include "attention_x.h"
#include "attention_x3.h"

template <int NKF, int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void xattn_x3_kernel(const AttnParams p, int wgs_per_pair, int blocks_per_wave) {
    constexpr int NKS = (NKF + 1) / 2;
    constexpr int QF = 2;
    constexpr int NFR = NKF * 2 + 4 * NKS;            // fragments per pass (K: NKF x 2 d-steps, V^T: 4 d-fragments x NKS key steps); x 2 images (hi, lo)
    constexpr int OOB = (int)0x80000000;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pair = blockIdx.x / wgs_per_pair, chunk = (blockIdx.x - pair * wgs_per_pair) * NW + wave;
    const int b = pair / p.heads, head = pair - b * p.heads;
    const int l15 = lane & 15, g = lane >> 4;
    const int C = p.heads * 64;

    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.q), 0, 0x7ffff000, 0x00020000);
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.k), 0, 0x7ffff000, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.vt), 0, 0x7ffff000, 0x00020000);
    auto ld4 = [&](const __amdgpu_buffer_rsrc_t& r, int vo, int so) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0)); };

    // ---- fragment images of every active pass -> LDS: fragment i of pass ps, hi image at (ps * 2 NFR + i) KiB, lo image NFR KiB behind it ----
    for (int ps = 0; ps < p.npass; ++ps) {
        const AttnEntry& e = p.e[ps * ATT_MAXB + b];
        if (e.w_const == 0.f && e.w_slope == 0.f) continue;
        for (int i = wave; i < NFR; i += NW) {
            f32x4 a, c;
            if (i < 2 * NKF) {                        // K fragment (key fragment f, d-step ks): lane = key 16 f + l15, d = 32 ks + 8 g .. + 7
                const int f = i >> 1, ks = i & 1;
                const int key = 16 * f + l15;
                const int vo = key < p.Sk ? (key * p.ldk + 8 * g) * 4 : OOB;
                const int so = ((e.kv_row * p.Sk) * p.ldk + head * 64 + 32 * ks) * 4;
                a = ld4(rk, vo, so);
                c = ld4(rk, vo, so + 16);
            } else {                                  // V^T fragment (d fragment df, key step s): lane = d 16 df + l15, keys {32 s + 4 g + e, 32 s + 16 + 4 g + e}
                const int j = i - 2 * NKF, df = j / NKS, s = j - df * NKS;
                const int vo = (l15 * p.ldvt + 4 * g) * 4;
                const int so = ((e.kv_row * C + head * 64 + 16 * df) * p.ldvt + 32 * s) * 4;
                a = ld4(rv, 32 * s + 4 * g < p.ldvt ? vo : OOB, so);
                c = ld4(rv, 32 * s + 16 + 4 * g < p.ldvt ? vo : OOB, so + 64);
#pragma unroll
                for (int r = 0; r < 4; ++r) {         // columns past Sk are padding of the V^T rows (ldvt >= Sk): P is exactly 0 there, the operand must be finite
                    if (32 * s + 4 * g + r >= p.Sk) a[r] = 0.f;
                    if (32 * s + 16 + 4 * g + r >= p.Sk) c[r] = 0.f;
                }
            }
            u32x4 hi, lo;
            x3_split8(a, c, hi, lo);
            char* dst = smem + (ps * 2 * NFR + i) * 1024 + lane * 16;
            *reinterpret_cast<u32x4*>(dst) = hi;
            *reinterpret_cast<u32x4*>(dst + NFR * 1024) = lo;
        }
    }
    __syncthreads();

    const float c = p.scale * 1.44269504088896340736f;
    const int qvo = (l15 * p.ldq + 8 * g) * 4;
    const int blk0 = chunk * blocks_per_wave;
    const int nblk_total = (p.S + 31) / 32;
    int nblk = nblk_total - blk0;
    if (nblk > blocks_per_wave) nblk = blocks_per_wave;

    // work items = (query block, active pass) in order; the Q rows (and per-query weights) of the NEXT item are requested before the current
    // item's arithmetic starts, so a wave's HBM latency hides behind its own MFMA / softmax work (one or two waves per SIMD run here)
    auto active_from = [&](int ps) {
        for (; ps < p.npass; ++ps) {
            const AttnEntry& e = p.e[ps * ATT_MAXB + b];
            if (e.w_const != 0.f || e.w_slope != 0.f) break;
        }
        return ps;
    };
    f32x4 qraw[QF][2][2];
    float wqn[QF];
    auto request_q = [&](int blk, int ps) {
        const AttnEntry& e = p.e[ps * ATT_MAXB + b];
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            const int qrow = blk * 32 + 16 * f + l15;
            const int vo = qrow < p.S ? qvo : OOB;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int so = ((e.q_row * p.S + blk * 32 + 16 * f) * p.ldq + head * 64 + 32 * ks) * 4;
                qraw[f][ks][0] = ld4(rq, vo, so);
                qraw[f][ks][1] = ld4(rq, vo, so + 16);
            }
            wqn[f] = (e.wq && qrow < p.S) ? e.wq[qrow] : 1.f;
        }
    };
    const int ps0 = active_from(0);
    if (nblk > 0 && ps0 < p.npass) request_q(blk0, ps0);

    for (int ib = 0; ib < nblk; ++ib) {
        const int blk = blk0 + ib;
        f32x4 acc[4][QF];
#pragma unroll
        for (int df = 0; df < 4; ++df)
#pragma unroll
            for (int qf = 0; qf < QF; ++qf) acc[df][qf] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ps = ps0; ps < p.npass;) {
            const AttnEntry& e = p.e[ps * ATT_MAXB + b];
            const int ps_next = active_from(ps + 1);
            const float w = e.w_const + (e.w_slope != 0.f ? e.w_slope * *p.w_dev : 0.f);
            const char* img = smem + ps * 2 * NFR * 1024 + lane * 16;
            u32x4 qh[QF][2], ql[QF][2];
            float wql[QF];
#pragma unroll
            for (int f = 0; f < QF; ++f) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) x3_split8(qraw[f][ks][0], qraw[f][ks][1], qh[f][ks], ql[f][ks]);
                wql[f] = wqn[f];
            }
            if (ps_next < p.npass) request_q(blk, ps_next);
            else if (ib + 1 < nblk) request_q(blk + 1, ps0);
            f32x4 st[NKF][QF];
#pragma unroll
            for (int f = 0; f < NKF; ++f) {
#pragma unroll
                for (int qf = 0; qf < QF; ++qf) st[f][qf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const u32x4 kh = *reinterpret_cast<const u32x4*>(img + (2 * f + ks) * 1024);
                    const u32x4 kl = *reinterpret_cast<const u32x4*>(img + (NFR + 2 * f + ks) * 1024);
#pragma unroll
                    for (int qf = 0; qf < QF; ++qf) {
                        x3_mma(kl, qh[qf][ks], st[f][qf]);      // small terms first
                        x3_mma(kh, ql[qf][ks], st[f][qf]);
                        x3_mma(kh, qh[qf][ks], st[f][qf]);
                    }
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
            u32x4 ph[QF][NKS], pl[QF][NKS];
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
                    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
                    x3_split8(st[2 * s][qf], 2 * s + 1 < NKF ? st[2 * s + 1][qf] : zero, ph[qf][s], pl[qf][s]);
                }
            }
#pragma unroll
            for (int df = 0; df < 4; ++df) {
                u32x4 vh[NKS], vl[NKS];
#pragma unroll
                for (int s = 0; s < NKS; ++s) {
                    vh[s] = *reinterpret_cast<const u32x4*>(img + (2 * NKF + df * NKS + s) * 1024);
                    vl[s] = *reinterpret_cast<const u32x4*>(img + (NFR + 2 * NKF + df * NKS + s) * 1024);
                }
#pragma unroll
                for (int qf = 0; qf < QF; ++qf) {
                    f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < NKS; ++s) {
                        x3_mma(vl[s], ph[qf][s], o);
                        x3_mma(vh[s], pl[qf][s], o);
                        x3_mma(vh[s], ph[qf][s], o);
                    }
                    acc[df][qf] += o * sc[qf];
                }
            }
            ps = ps_next;
        }
        // lane (l15, g) holds O[q = 16 qf + l15][d = 16 df + 4 g + r]: four consecutive columns.  A swap between the 16-lane rows (v_permlane16_swap on
        // the fragment pair (2 j, 2 j + 1)) leaves it with EIGHT consecutive columns, 16 (g & 1) + 8 (g >> 1) .., of the 32-column block j: 16-byte
        // stores, half as many (-8 % on the 64x64-level launch; the GEMM epilogues' wave permutation on top of it measured no further gain here)
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) {
            const int q = blk * 32 + 16 * qf + l15;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 x, y;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[2 * j][qf][r]), __float_as_uint(acc[2 * j + 1][qf][r]), false, false);
                    x[r] = __uint_as_float(sw[0]);
                    y[r] = __uint_as_float(sw[1]);
                }
                if (q >= p.S) continue;
                const int col = head * 64 + 32 * j + 16 * (g & 1) + 8 * (g >> 1);
                if (p.out_pair) {                     // C = 64 heads: always the blocked pair form (common.h pair_pos)
                    u32x4 hi, lo;
                    x3_split8(x, y, hi, lo);
                    bf16* row = reinterpret_cast<bf16*>(p.out) + ((long)b * p.S + q) * p.ldo + pair_pos(col, C);
                    *reinterpret_cast<u32x4*>(row) = hi;
                    *reinterpret_cast<u32x4*>(row + 32) = lo;
                } else {
                    float* row = reinterpret_cast<float*>(p.out) + ((long)b * p.S + q) * p.ldo + col;
                    *reinterpret_cast<f32x4*>(row) = x;
                    *reinterpret_cast<f32x4*>(row + 4) = y;
                }
            }
        }
    }
}
