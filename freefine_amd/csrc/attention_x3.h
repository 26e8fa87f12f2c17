// attn_x3_kernel: the multi-pass masked attention of attention.h in SPLIT-BF16 arithmetic (FFN_BF16X3): q / k / v^T / out are fp32 in
// HBM exactly as in parity mode; every operand of the two products is carried as hi = bf16(x), lo = bf16(x - hi) and
//     S^T = K_hi.Q_hi^T + K_lo.Q_hi^T + K_hi.Q_lo^T          O^T += V^T_hi.P_hi^T + V^T_lo.P_hi^T + V^T_hi.P_lo^T
// run on the bf16 16x16x32 MFMA with fp32 accumulation (dropped terms ~2^-18): fp32-level results at ~1/3 of the bf16 MFMA rate
// instead of the 1/16 of the exact-fp32 MFMA chain.  Softmax statistics, the running maximum, the probabilities before their split
// and all pass / mask logic are fp32 and identical to attn_kernel's generic tile (same pass table, same semantics: see attention.h).
//
// Structure: 8 waves per workgroup (2 per SIMD: one wave's softmax / conversions run beside the other's MFMA chain), 32 queries per
// wave (256 per workgroup), 64-key tiles.  The fp32 K and V^T tiles are fetched to registers one tile ahead, SPLIT there, and written
// to LDS as four bf16 tiles (K_hi, K_lo: [64 keys][D]; V^T_hi, V^T_lo: [D][64 keys]; double buffered); Q is split in registers when
// its fragments are loaded, P when it leaves the softmax.  d <= 64 (padded to 64): SD-2.1's heads; other head sizes take the exact
// fp32 kernel.
#pragma once
#include "attention.h"
#include "igemm.h"

__device__ __forceinline__ void x3_split8(const f32x4& a, const f32x4& b, u32x4& hi, u32x4& lo) {
    hi[0] = pack_bf16x2(a[0], a[1]);
    hi[1] = pack_bf16x2(a[2], a[3]);
    hi[2] = pack_bf16x2(b[0], b[1]);
    hi[3] = pack_bf16x2(b[2], b[3]);
    lo[0] = pack_bf16x2(a[0] - __uint_as_float(hi[0] << 16), a[1] - __uint_as_float(hi[0] & 0xffff0000u));
    lo[1] = pack_bf16x2(a[2] - __uint_as_float(hi[1] << 16), a[3] - __uint_as_float(hi[1] & 0xffff0000u));
    lo[2] = pack_bf16x2(b[0] - __uint_as_float(hi[2] << 16), b[1] - __uint_as_float(hi[2] & 0xffff0000u));
    lo[3] = pack_bf16x2(b[2] - __uint_as_float(hi[3] << 16), b[3] - __uint_as_float(hi[3] & 0xffff0000u));
}
__device__ __forceinline__ void x3_mma(const u32x4& a, const u32x4& b, f32x4& c) { DT<bf16>::mma(a, b, c); }

template <bool MASKS>
__global__ __launch_bounds__(512) void attn_x3_kernel(const AttnParams p) {
    constexpr int DP = 64, QF = 2, KT = 64, NW = 8;
    constexpr int NT = KT / 16;                 // 16-key fragments per tile
    constexpr int KROW = 128, VROW = 128;       // bf16 rows of 64 elements; 16-byte chunks XOR-swizzled (K: row & 7, V^T: (row >> 1) & 7)
    constexpr int DSL = DP / 32;                // 32-element d-slabs = MFMA k-steps of QK^T
    constexpr int FD = DP / 16;                 // d fragments of O^T
    constexpr int NPC = KT / 32;                // 32-key chunks = MFMA k-steps of PV
    constexpr int KBUF = KT * KROW, VBUF = DP * VROW, BUF = 2 * KBUF + 2 * VBUF;      // one stage: K_hi | K_lo | V_hi | V_lo
    constexpr float NEG = -1e30f;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int nqb = (p.S + 32 * NW - 1) / (32 * NW);
    const int Lb = (p.heads * p.Bo >= ATT_XCD_MIN_GROUPS) ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int qblk = Lb % nqb, head = (Lb / nqb) % p.heads, b = Lb / (nqb * p.heads);
    const int q0 = qblk * (32 * NW) + wave * 32;
    const int D = p.D;
    const float* __restrict__ Qg = reinterpret_cast<const float*>(p.q);
    const float* __restrict__ Kg = reinterpret_cast<const float*>(p.k);
    const float* __restrict__ Vg = reinterpret_cast<const float*>(p.vt);
    float* __restrict__ Og = reinterpret_cast<float*>(p.out);
    const float c_exp = p.scale * 1.44269504088896340736f;      // softmax in base 2: p = exp2(s * c - m)

    f32x4* totl = reinterpret_cast<f32x4*>(smem + 2 * BUF) + wave * (FD * QF * 64) + lane;     // multi-pass sums: [wave][FD][QF][lane]
    int nactive = 0, nseen = 0;
    for (int pass = 0; pass < p.npass; ++pass) {
        const AttnEntry& e0 = p.e[pass * ATT_MAXB + b];
        nactive += (e0.w_const != 0.f || e0.w_slope != 0.f) ? 1 : 0;
    }
    const int dup = att_duplicate_pass(p, b, head);      // a self-referencing row's second pass on a head the mask skips: folded into the first
    if (dup >= 0) nactive = 1;
    // output row q of batch row b, columns head * D + d .. + 3: fp32, or (out_pair) the bf16 pair form hi | lo at ldo / 2
    auto store_out = [&](int q, int d, const float* vv) {
        if (p.out_pair) store_pair_row4(reinterpret_cast<bf16*>(p.out) + ((long)b * p.S + q) * p.ldo, head * D + d, p.ldo / 2, vv);
        else store4(Og + ((long)b * p.S + q) * p.ldo + head * D + d, vv);
    };
    if (nactive == 0) {
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            const int q = q0 + f * 16 + l15;
#pragma unroll
            for (int i = 0; i < FD; ++i) {
                const int d = i * 16 + 4 * g;
                float z[4] = {0.f, 0.f, 0.f, 0.f};
                if (q < p.S && d < D) store_out(q, d, z);
            }
        }
        return;
    }
    const int ntiles = (p.Sk + KT - 1) / KT;
    // staging role of this thread: one 8-element chunk of the K tile (key sk, d chunk sc) and one of the V^T tile (d row sk, key chunk sc)
    const int sk = tid >> 3, sc = tid & 7;

    for (int pass = 0; pass < p.npass; ++pass) {
        const AttnEntry& en = p.e[pass * ATT_MAXB + b];
        if (en.w_const == 0.f && en.w_slope == 0.f) continue;
        if (pass == dup) continue;
        float w = en.w_const;
        if (p.w_dev) w += en.w_slope * (*p.w_dev);
        if (dup >= 0) w += att_pass_weight(p, p.e[dup * ATT_MAXB + b]);
        const int hb = en.hr_row > 0 ? en.hr_row - 1 : b;
        const bool pass_masked = MASKS && en.kmask && (!(en.flags & ATT_HEAD_RULE) || (((hb * p.heads + head) & 1) == 0));

        // ---- Q^T fragments, split: lane = query l15 of fragment f, d elements 32 s + 8 g .. + 7
        u32x4 qh[QF][DSL], ql[QF][DSL];
        float wq[QF];
        int mode[QF];
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            const int q = q0 + f * 16 + l15;
            const bool qok = q < p.S;
#pragma unroll
            for (int s = 0; s < DSL; ++s) {
                const int d = 32 * s + 8 * g;
                f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f}, c = a;
                const float* src = Qg + ((long)en.q_row * p.S + q) * p.ldq + head * D + d;
                if (qok && d < D) a = *reinterpret_cast<const f32x4*>(src);
                if (qok && d + 4 < D) c = *reinterpret_cast<const f32x4*>(src + 4);
                x3_split8(a, c, qh[f][s], ql[f][s]);
            }
            wq[f] = (en.wq && qok) ? en.wq[q] : 1.f;
            int md = 0;
            if (pass_masked) {
                const int sel = (en.qsel && qok) ? (en.qsel[q] != 0) : 1;
                md = sel ? 1 : 2;
                if (sel && (en.flags & ATT_UNIFORM_SEL1)) md = 3;
                if (!sel && (en.flags & ATT_UNIFORM_SEL0)) md = 3;
            }
            mode[f] = md;
        }
        f32x4 o[FD][QF];
        float mrun[QF], lrun[QF];
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            mrun[f] = NEG;
            lrun[f] = 0.f;
#pragma unroll
            for (int i = 0; i < FD; ++i) o[i][f] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

        // ---- tile staging: fp32 global -> registers (one tile ahead) -> split -> four bf16 LDS tiles (double buffered)
        const float* kbase = Kg + (long)en.kv_row * p.Sk * p.ldk + head * D;
        const float* vbase = Vg + ((long)en.kv_row * p.heads * D + head * D) * p.ldvt;
        f32x4 rk0, rk1, rv0, rv1;
        auto issue = [&](int k0) {
            rk0 = rk1 = rv0 = rv1 = f32x4{0.f, 0.f, 0.f, 0.f};
            {   // K: key k0 + sk, d elements 8 sc .. 8 sc + 7
                const int d = 8 * sc;
                const float* src = kbase + (long)(k0 + sk) * p.ldk + d;
                if (k0 + sk < p.Sk && d < D) rk0 = *reinterpret_cast<const f32x4*>(src);
                if (k0 + sk < p.Sk && d + 4 < D) rk1 = *reinterpret_cast<const f32x4*>(src + 4);
            }
            {   // V^T: d row sk, keys k0 + 8 sc .. + 7 (ldvt is a multiple of 4: a 4-key piece that starts inside Sk may end in the
                // finite padding up to ldvt only when Sk % 4 != 0 -> element-wise then)
                const int kk = k0 + 8 * sc;
                const float* src = vbase + (long)sk * p.ldvt + kk;
                if (sk < D) {
                    if (kk + 4 <= p.Sk) rv0 = *reinterpret_cast<const f32x4*>(src);
                    else
                        for (int e = 0; e < 4; ++e)
                            if (kk + e < p.Sk) rv0[e] = src[e];
                    if (kk + 8 <= p.Sk) rv1 = *reinterpret_cast<const f32x4*>(src + 4);
                    else
                        for (int e = 0; e < 4; ++e)
                            if (kk + 4 + e < p.Sk) rv1[e] = src[4 + e];
                }
            }
        };
        auto stage = [&](int buf) {
            char* B = smem + buf * BUF;
            u32x4 hi, lo;
            x3_split8(rk0, rk1, hi, lo);
            const int ko = sk * KROW + ((sc ^ (sk & 7)) << 4);
            *reinterpret_cast<u32x4*>(B + ko) = hi;
            *reinterpret_cast<u32x4*>(B + KBUF + ko) = lo;
            x3_split8(rv0, rv1, hi, lo);
            const int vo = sk * VROW + ((sc ^ ((sk >> 1) & 7)) << 4);
            *reinterpret_cast<u32x4*>(B + 2 * KBUF + vo) = hi;
            *reinterpret_cast<u32x4*>(B + 2 * KBUF + VBUF + vo) = lo;
        };

        auto tile = [&](int k0, int buf, auto masked_tag) {
            constexpr bool MASKED = decltype(masked_tag)::value;
            const char* Kh = smem + buf * BUF;
            const char* Kl = Kh + KBUF;
            const char* Vh = Kh + 2 * KBUF;
            const char* Vl = Vh + VBUF;
            f32x4 st[NT][QF];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int f = 0; f < QF; ++f) st[t][f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < DSL; ++s) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int krow = t * 16 + l15;
                    const int off = krow * KROW + (((4 * s + g) ^ (krow & 7)) << 4);
                    const u32x4 kh = *reinterpret_cast<const u32x4*>(Kh + off);
                    const u32x4 kl = *reinterpret_cast<const u32x4*>(Kl + off);
#pragma unroll
                    for (int f = 0; f < QF; ++f) {
                        x3_mma(kl, qh[f][s], st[t][f]);      // small terms first
                        x3_mma(kh, ql[f][s], st[t][f]);
                        x3_mma(kh, qh[f][s], st[t][f]);
                    }
                }
            }
            uint32_t inr = 0, mk = 0;
            if (MASKED) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int kb = k0 + t * 16 + 4 * g;
                    uint32_t w4 = 0;
                    if (kb + 4 <= p.Sk) {
                        inr |= 0xfu << (4 * t);
                        if (pass_masked) w4 = *reinterpret_cast<const uint32_t*>(en.kmask + kb);
                    } else {
                        for (int r = 0; r < 4; ++r)
                            if (kb + r < p.Sk) {
                                inr |= 1u << (4 * t + r);
                                if (pass_masked) w4 |= (uint32_t)en.kmask[kb + r] << (8 * r);
                            }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) mk |= (((w4 >> (8 * r)) & 0xff) != 0 ? 1u : 0u) << (4 * t + r);
                }
            }
            u32x4 ph[NPC][QF], pl[NPC][QF];      // split P^T chunks (B operand of O^T = V^T.P^T): keys 32 c + 4 g + r and 32 c + 16 + 4 g + r
#pragma unroll
            for (int f = 0; f < QF; ++f) {
                float tmax = NEG;
                uint32_t am = 0;
                if (MASKED) {
                    const int md = mode[f];
                    am = inr & (md == 1 ? mk : (md == 2 ? ~mk : 0xffffffffu));
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float sv = (md == 3) ? 0.f : st[t][f][r];
                            sv = ((am >> (4 * t + r)) & 1u) ? sv : NEG;
                            st[t][f][r] = sv;
                            tmax = fmaxf(tmax, sv);
                        }
                } else {
#pragma unroll
                    for (int t = 0; t < NT; ++t) tmax = fmaxf(tmax, fmaxf(fmaxf(st[t][f][0], st[t][f][1]), fmaxf(st[t][f][2], st[t][f][3])));
                }
                tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
                const float mnew = fmaxf(mrun[f], tmax * c_exp);
                const float alpha = __builtin_amdgcn_exp2f(mrun[f] - mnew);
                const bool grew = mnew != mrun[f];
                mrun[f] = mnew;
                float psum = 0.f;
                float pv[NT][4];
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float pe = __builtin_amdgcn_exp2f(fmaf(st[t][f][r], c_exp, -mnew));
                        if (MASKED) pe = ((am >> (4 * t + r)) & 1u) ? pe : 0.f;
                        pv[t][r] = pe;
                        psum += pe;
                    }
                lrun[f] = lrun[f] * alpha + psum;
                if (__any(grew)) {
#pragma unroll
                    for (int i = 0; i < FD; ++i) o[i][f] *= alpha;
                }
#pragma unroll
                for (int c = 0; c < NPC; ++c)
                    x3_split8(f32x4{pv[2 * c][0], pv[2 * c][1], pv[2 * c][2], pv[2 * c][3]},
                              f32x4{pv[2 * c + 1][0], pv[2 * c + 1][1], pv[2 * c + 1][2], pv[2 * c + 1][3]}, ph[c][f], pl[c][f]);
            }
#pragma unroll
            for (int c = 0; c < NPC; ++c) {
#pragma unroll
                for (int i = 0; i < FD; ++i) {
                    const int vr = i * 16 + l15;
                    const int vsw = (vr >> 1) & 7;
                    const int o0 = vr * VROW + (((4 * c + (g >> 1)) ^ vsw) << 4) + 8 * (g & 1);          // keys 32 c + 4 g .. + 3
                    const int o1 = vr * VROW + (((4 * c + 2 + (g >> 1)) ^ vsw) << 4) + 8 * (g & 1);      // keys 32 c + 16 + 4 g .. + 3
                    const u32x2 h0 = *(const volatile lds_u32x2_t*)(Vh + o0), h1 = *(const volatile lds_u32x2_t*)(Vh + o1);
                    const u32x2 l0 = *(const volatile lds_u32x2_t*)(Vl + o0), l1 = *(const volatile lds_u32x2_t*)(Vl + o1);
                    const u32x4 vh = u32x4{h0[0], h0[1], h1[0], h1[1]}, vl = u32x4{l0[0], l0[1], l1[0], l1[1]};
#pragma unroll
                    for (int f = 0; f < QF; ++f) {
                        x3_mma(vl, ph[c][f], o[i][f]);
                        x3_mma(vh, pl[c][f], o[i][f]);
                        x3_mma(vh, ph[c][f], o[i][f]);
                    }
                }
            }
        };

        issue(0);
        stage(0);
        __syncthreads();
        for (int t = 0; t < ntiles; ++t) {
            const int k0 = t * KT, buf = t & 1;
            if (t + 1 < ntiles) issue(k0 + KT);
            if (pass_masked || k0 + KT > p.Sk) tile(k0, buf, std::true_type{});
            else tile(k0, buf, std::false_type{});
            if (t + 1 < ntiles) stage(buf ^ 1);      // the other stage was last read in tile t-1, before the barrier that ended it
            __syncthreads();
        }

        ++nseen;
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            float l = lrun[f];
            l += __shfl_xor(l, 16);
            l += __shfl_xor(l, 32);
            const float sc_ = (l > 0.f) ? (w * wq[f] / l) : 0.f;
            const int q = q0 + f * 16 + l15;
#pragma unroll
            for (int i = 0; i < FD; ++i) {
                f32x4 v = o[i][f] * sc_;
                if (nseen > 1) v += totl[(i * QF + f) * 64];
                if (nseen < nactive) {
                    totl[(i * QF + f) * 64] = v;
                } else {
                    const int d = i * 16 + 4 * g;
                    float vv[4] = {v[0], v[1], v[2], v[3]};
                    if (q < p.S && d < D) store_out(q, d, vv);
                }
            }
        }
    }
}
