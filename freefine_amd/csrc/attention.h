// Multi-pass masked attention for the FreeFine attention modulation, flash style, gfx950.
//
//   out[b, q, head, :] = sum_p  w_p(b) * wq_p[q] * softmax_k( scale * <Q[qrow_p(b), q], K[kvrow_p(b), k]> + mask_p(q, k) ) V[kvrow_p(b), k]
//
// One launch evaluates, for every output batch row b, up to MAXP "passes"; each (pass, row) entry names which
// batch row supplies Q, which supplies K/V, a scalar weight (constant + slope * device scalar, so that the
// per-step context_guidance can live in device memory under hipGraph replay), an optional per-query weight
// vector, an optional per-key byte mask and an optional per-query byte selector:
//     allowed(q, k) = (kmask[k] != 0) == (qsel[q] != 0)          (qsel == null -> 1)
// With the reference's tiled-head quirk (head_rule): the mask applies only where j = b*heads + head is even.
// If the allowed set of a query is empty the reference's additive finfo.min mask degenerates to a uniform
// softmax over ALL keys (src/utils/attention.py:856-858, baddbmm with beta=1) -- reproduced via the
// ATT_UNIFORM_SEL* flags computed by the host from the mask population.
//
// This one kernel covers (reference file:line, /root/reference/src/utils/attention.py):
//   plain attention (394-404), Temporal_contextal_attention (1043-1091), _bg (1284-1324), _compose (1092-1140),
//   modulate_local_cross_attn (1360-1393), _bg (1326-1357), _compose (1394-1432), and SSA/SDSA (1142-1192) once
//   the host has concatenated own+reference K/V along the key axis.
// It never materialises the [4*heads, S, S] additive masks the reference builds (attention.py:871-881).
//
// Formulation: S^T = K . Q^T with 16x16 MFMA tiles so that a lane owns ONE query (column l&15) and 4 keys per
// 16-key fragment: softmax statistics are lane-local apart from two cross-lane-group shuffles, the exponentiated
// P^T fragments are already the B operand of O^T = V^T . P^T (no LDS round trip for P), and the O^T accumulator
// keeps the same query on the lane, so the online-softmax rescale is lane-local as well.  V arrives TRANSPOSED
// ([Bk, heads*D, ldvt], written that way for free by the to_v projection's epilogue, IG_OUT_TRANSPOSED).
#pragma once
#include "common.h"
#include "../../include/freefine_hip.h"

#define ATT_MAXP FFN_ATT_MAXP
#define ATT_MAXB FFN_ATT_MAXB
enum { ATT_HEAD_RULE = FFN_ATT_HEAD_RULE, ATT_UNIFORM_SEL1 = FFN_ATT_UNIFORM_SEL1, ATT_UNIFORM_SEL0 = FFN_ATT_UNIFORM_SEL0 };
typedef ffn_attn_entry AttnEntry;
typedef ffn_attn_desc AttnParams;

template <typename T, int DP, int QF>
__global__ __launch_bounds__(256) void attn_kernel(const AttnParams p) {
    constexpr int EPC = DT<T>::EPC;
    constexpr int SZ = sizeof(T);
    constexpr int KT = 64;                      // keys per tile
    constexpr int KROW = DP * SZ + 16;          // K tile row stride (bytes), +16 B pad against bank conflicts
    constexpr int VROW = KT * SZ + 16;          // V^T tile row stride
    constexpr int DCH = DP / EPC;               // 16-byte chunks per K row
    constexpr int DSL = DP * SZ / 64;           // 64-byte d-slabs (MFMA k-substeps of QK^T)
    constexpr int FD = DP / 16;                 // d fragments of O^T
    constexpr int KPC = EPC / 4;                // key fragments per PV chunk (f32: 1, bf16: 2)
    constexpr int VCH = KT / EPC;               // 16-byte chunks per V^T row
    constexpr float NEG = -1e30f;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;                 // [KT][KROW]
    char* Vs = smem + KT * KROW;     // [DP][VROW]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, head = blockIdx.y;
    const int q0 = blockIdx.x * (64 * QF) + wave * (16 * QF);
    const int D = p.D;
    const T* __restrict__ Qg = reinterpret_cast<const T*>(p.q);
    const T* __restrict__ Kg = reinterpret_cast<const T*>(p.k);
    const T* __restrict__ Vg = reinterpret_cast<const T*>(p.vt);
    const float c_exp = p.scale * 1.44269504088896340736f;

    f32x4 tot[FD][QF];
#pragma unroll
    for (int i = 0; i < FD; ++i)
#pragma unroll
        for (int j = 0; j < QF; ++j) tot[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int pass = 0; pass < p.npass; ++pass) {
        const AttnEntry& en = p.e[pass * ATT_MAXB + b];
        float w = en.w_const;
        if (p.w_dev) w += en.w_slope * (*p.w_dev);
        const bool skip = (en.w_const == 0.f && en.w_slope == 0.f);  // block-uniform
        if (skip) continue;

        // ---- Q^T fragments (B operand of S^T = K.Q^T): lane = query l15 of fragment f, chunk 4s+g ------------
        u32x4 qf[QF][DSL];
        float wq[QF];
        int mode[QF];  // 0 = all keys, 1 = keys with mask!=0, 2 = keys with mask==0, 3 = uniform over all keys
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            const int q = q0 + f * 16 + l15;
            const bool qok = q < p.S;
#pragma unroll
            for (int s = 0; s < DSL; ++s) {
                const int d = (4 * s + g) * EPC;
                qf[f][s] = u32x4{0, 0, 0, 0};
                if (qok && d < D)
                    qf[f][s] = *reinterpret_cast<const u32x4*>(Qg + ((long)en.q_row * p.S + q) * p.ldq + head * D + d);
            }
            wq[f] = (en.wq && qok) ? en.wq[q] : 1.f;
            int md = 0;
            const bool masked = en.kmask && (!(en.flags & ATT_HEAD_RULE) || (((b * p.heads + head) & 1) == 0));
            if (masked) {
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

        for (int k0 = 0; k0 < p.Sk; k0 += KT) {
            __syncthreads();  // previous tile fully consumed
            // ---- stage K tile [KT][DP] and V^T tile [DP][KT] ------------------------------------------------
            for (int cid = tid; cid < KT * DCH; cid += 256) {
                const int key = cid / DCH, c = cid - key * DCH;
                u32x4 v = u32x4{0, 0, 0, 0};
                if (k0 + key < p.Sk && c * EPC < D)
                    v = *reinterpret_cast<const u32x4*>(Kg + ((long)en.kv_row * p.Sk + k0 + key) * p.ldk + head * D + c * EPC);
                *reinterpret_cast<u32x4*>(Ks + key * KROW + c * 16) = v;
            }
            for (int cid = tid; cid < DP * VCH; cid += 256) {
                const int d = cid / VCH, c = cid - d * VCH;
                u32x4 v = u32x4{0, 0, 0, 0};
                const int kk = k0 + c * EPC;
                if (d < D) {
                    const T* src = Vg + ((long)en.kv_row * p.heads * D + head * D + d) * p.ldvt + kk;
                    if (kk + EPC <= p.Sk) {
                        v = *reinterpret_cast<const u32x4*>(src);
                    } else if (kk < p.Sk) {
                        float tmp[EPC];
#pragma unroll
                        for (int e = 0; e < EPC; ++e) tmp[e] = (kk + e < p.Sk) ? DT<T>::ld(src + e) : 0.f;
                        v = DT<T>::pack(tmp);
                    }
                }
                *reinterpret_cast<u32x4*>(Vs + d * VROW + c * 16) = v;
            }
            __syncthreads();

            // ---- S^T = K . Q^T : st[t][f] holds keys 16t+4g+r (r = reg) for query l15 of fragment f ---------
            f32x4 st[4][QF];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int f = 0; f < QF; ++f) st[t][f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < DSL; ++s) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const u32x4 ka = *reinterpret_cast<const u32x4*>(Ks + (t * 16 + l15) * KROW + (4 * s + g) * 16);
#pragma unroll
                    for (int f = 0; f < QF; ++f) DT<T>::mma(ka, qf[f][s], st[t][f]);
                }
            }

            // ---- key masks for this lane's keys ------------------------------------------------------------
            uint32_t km[4];  // byte r of km[t] = kmask[k0 + 16t + 4g + r]
            bool kin[4][4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int kb = k0 + t * 16 + 4 * g;
                km[t] = 0;
                if (en.kmask) {
                    if (kb + 4 <= p.Sk)
                        km[t] = *reinterpret_cast<const uint32_t*>(en.kmask + kb);
                    else
                        for (int r = 0; r < 4; ++r)
                            if (kb + r < p.Sk) km[t] |= (uint32_t)en.kmask[kb + r] << (8 * r);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) kin[t][r] = (kb + r) < p.Sk;
            }

            // ---- online softmax per query fragment ---------------------------------------------------------
            u32x4 pb[4 / KPC][QF];  // packed P^T chunks (B operand of O^T = V^T.P^T)
#pragma unroll
            for (int f = 0; f < QF; ++f) {
                const int md = mode[f];
                bool al[4][4];
                float tmax = NEG;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool mk = ((km[t] >> (8 * r)) & 0xff) != 0;
                        bool a = kin[t][r];
                        if (md == 1) a = a && mk;
                        if (md == 2) a = a && !mk;
                        al[t][r] = a;
                        float sv = (md == 3) ? 0.f : st[t][f][r];
                        sv = a ? sv : NEG;
                        st[t][f][r] = sv;
                        tmax = fmaxf(tmax, sv);
                    }
                tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
                const float mnew = fmaxf(mrun[f], tmax);
                const float alpha = exp2f((mrun[f] - mnew) * c_exp);
                mrun[f] = mnew;
                float psum = 0.f;
                float pv[4][4];
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pe = al[t][r] ? exp2f((st[t][f][r] - mnew) * c_exp) : 0.f;
                        pv[t][r] = pe;
                        psum += pe;
                    }
                lrun[f] = lrun[f] * alpha + psum;
#pragma unroll
                for (int i = 0; i < FD; ++i) o[i][f] *= alpha;
#pragma unroll
                for (int c = 0; c < 4 / KPC; ++c) {
                    float tmp[EPC];
#pragma unroll
                    for (int e = 0; e < EPC; ++e) tmp[e] = pv[c * KPC + e / 4][e & 3];
                    pb[c][f] = DT<T>::pack(tmp);
                }
            }

            // ---- O^T += V^T . P^T ----------------------------------------------------------------------------
#pragma unroll
            for (int c = 0; c < 4 / KPC; ++c) {
#pragma unroll
                for (int i = 0; i < FD; ++i) {
                    u32x4 va;
                    const char* vrow = Vs + (i * 16 + l15) * VROW;
                    if (KPC == 1) {
                        va = *reinterpret_cast<const u32x4*>(vrow + (16 * c + 4 * g) * SZ);
                    } else {
                        const u32x2 lo = *reinterpret_cast<const u32x2*>(vrow + (32 * c + 4 * g) * SZ);
                        const u32x2 hi = *reinterpret_cast<const u32x2*>(vrow + (32 * c + 16 + 4 * g) * SZ);
                        va = u32x4{lo[0], lo[1], hi[0], hi[1]};
                    }
#pragma unroll
                    for (int f = 0; f < QF; ++f) DT<T>::mma(va, pb[c][f], o[i][f]);
                }
            }
        }

        // ---- finish this pass: tot += w * wq[q] * O / l ------------------------------------------------------
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            float l = lrun[f];
            l += __shfl_xor(l, 16);
            l += __shfl_xor(l, 32);
            const float sc = (l > 0.f) ? (w * wq[f] / l) : 0.f;
#pragma unroll
            for (int i = 0; i < FD; ++i) tot[i][f] += o[i][f] * sc;
        }
    }

    // ---- store: lane holds O^T[d = 16i + 4g + r][q = l15] -> 4 consecutive d of one query ---------------------
    T* __restrict__ Og = reinterpret_cast<T*>(p.out);
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        const int q = q0 + f * 16 + l15;
        if (q >= p.S) continue;
#pragma unroll
        for (int i = 0; i < FD; ++i) {
            const int d = i * 16 + 4 * g;
            if (d < D) {
                float v[4] = {tot[i][f][0], tot[i][f][1], tot[i][f][2], tot[i][f][3]};
                store4(Og + ((long)b * p.S + q) * p.ldo + head * D + d, v);
            }
        }
    }
}
