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
#include <type_traits>

#include "common.h"
#include "../../include/freefine_hip.h"

#ifndef ATT_XCD_MIN_GROUPS
#define ATT_XCD_MIN_GROUPS 64
#endif
#define ATT_MAXP FFN_ATT_MAXP
#define ATT_MAXB FFN_ATT_MAXB
enum { ATT_HEAD_RULE = FFN_ATT_HEAD_RULE, ATT_UNIFORM_SEL1 = FFN_ATT_UNIFORM_SEL1, ATT_UNIFORM_SEL0 = FFN_ATT_UNIFORM_SEL0 };
typedef __attribute__((address_space(3))) u32x2 lds_u32x2_t;
typedef __attribute__((address_space(1))) const void* att_gptr_t;
typedef __attribute__((address_space(3))) void* att_lptr_t;
__device__ __attribute__((aligned(16))) const uint32_t g_att_zero[4] = {0, 0, 0, 0};   // explicit LDS pointer type for the volatile V^T fragment reads
typedef ffn_attn_entry AttnEntry;
typedef ffn_attn_desc AttnParams;

// plain v_max3 / v_max (fmaxf on MFMA outputs makes the compiler insert a canonicalising v_max per operand)
__device__ __forceinline__ float att_max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float att_max(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// max over the four lanes l, l^16, l^32, l^48 (the four key groups of one query) without an LDS round trip
__device__ __forceinline__ float att_max_groups(float x) {
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = att_max(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto c = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return att_max(__uint_as_float(c[0]), __uint_as_float(c[1]));
}

// Duplicate passes of a self-referencing row.  The TCA tables give EVERY row two passes -- (q = b, kv = ref(b), key mask under the
// tiled-head rule, weight cg) and (q = b, kv = b, no mask, weight 1 - cg) -- and for a reference row ref(b) = b
// (attention.py:1033-1035, 1060-1083).  Under the tiled-head rule the mask applies to the heads with even b * heads + head only
// (attention.py:859 vs 761), so on the other heads that row's two passes are the same plain attention twice, blended with weights
// that sum to one.  A (row, head) workgroup detects this and runs the first pass once with the summed weight: 1/12 of the two-pass
// work of an edit batch (reference rows = 1/3, odd heads = 1/2, one pass of two).  Returns the pass to skip (folded into pass 0) or -1.
__device__ __forceinline__ int att_duplicate_pass(const AttnParams& p, int b, int head) {
    if (p.npass != 2) return -1;
    const AttnEntry& a0 = p.e[b];
    const AttnEntry& a1 = p.e[ATT_MAXB + b];
    if ((a0.w_const == 0.f && a0.w_slope == 0.f) || (a1.w_const == 0.f && a1.w_slope == 0.f)) return -1;
    if (a0.q_row != a1.q_row || a0.kv_row != a1.kv_row || a0.wq != a1.wq) return -1;
    auto masked = [&](const AttnEntry& e) {
        const int hb = e.hr_row > 0 ? e.hr_row - 1 : b;
        return e.kmask && (!(e.flags & ATT_HEAD_RULE) || (((hb * p.heads + head) & 1) == 0));
    };
    return (masked(a0) || masked(a1)) ? -1 : 1;
}
__device__ __forceinline__ float att_pass_weight(const AttnParams& p, const AttnEntry& e) {
    float w = e.w_const;
    if (p.w_dev) w += e.w_slope * (*p.w_dev);
    return w;
}

// KT = keys per tile, OCC = min waves per SIMD, MASKS = some entry of the launch carries a key mask (bf16: compiles the
// mask-on-MFMA tile in; launches without masks -- cross attention, plain self attention -- get the leaner kernel)
template <typename T, int DP, int QF, int KT = 64, int OCC = 1, bool MASKS = true>
__global__ __launch_bounds__(256, OCC) void attn_kernel(const AttnParams p) {
    constexpr int EPC = DT<T>::EPC;
    constexpr int SZ = sizeof(T);
    constexpr int NT = KT / 16;                 // 16-key fragments per tile
    constexpr bool KSWZ = (DP * SZ == 128);     // 128-byte K rows: XOR-swizzled chunks (conflict-free ds_read_b128, like igemm)
    constexpr int KROW = KSWZ ? 128 : DP * SZ + 16;   // otherwise +16 B row padding (2-way conflicts on some lane groups)
    constexpr bool GLDS = KSWZ && (KT * SZ == 128);    // bf16, d = 64: K and V^T tiles go global -> LDS directly (no register staging)
    constexpr int VROW = GLDS ? 128 : KT * SZ + 16;   // V^T tile row stride (GLDS: unpadded 128-byte rows, chunks XOR-swizzled by (row>>1)&7)
    constexpr int DCH = DP / EPC;               // 16-byte chunks per K row
    constexpr int DSL = DP * SZ / 64;           // 64-byte d-slabs (MFMA k-substeps of QK^T)
    constexpr int FD = DP / 16;                 // d fragments of O^T
    constexpr int KPC = EPC / 4;                // key fragments per PV chunk (f32: 1, bf16: 2)
    constexpr int VCH = KT / EPC;               // 16-byte chunks per V^T row
    constexpr int NKC = KT * DCH / 256;         // K chunks staged per thread per tile
    constexpr int NVC = DP * VCH / 256;         // V^T chunks staged per thread per tile
    constexpr int KBUF = KT * KROW, VBUF = DP * VROW;
    constexpr float NEG = -1e30f;
    static_assert((KT * DCH) % 256 == 0 && (DP * VCH) % 256 == 0, "tile must split evenly over 256 threads");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;                 // [2][KT][KROW]
    char* Vs = smem + 2 * KBUF;      // [2][DP][VROW]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int nqb = (p.S + 64 * QF - 1) / (64 * QF);
    // XCD-contiguous logical order (query block fastest, then head, then row) once every XCD gets many (row, head) groups; with
    // few groups (single-image batches) whole groups of unequal cost (masked / unmasked heads) per XCD unbalance the chip instead
    const int Lb = (p.heads * p.Bo >= ATT_XCD_MIN_GROUPS) ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int qblk = Lb % nqb, head = (Lb / nqb) % p.heads, b = Lb / (nqb * p.heads);
    const int q0 = qblk * (64 * QF) + wave * (16 * QF);
    const int D = p.D;
    const T* __restrict__ Qg = reinterpret_cast<const T*>(p.q);
    const T* __restrict__ Kg = reinterpret_cast<const T*>(p.k);
    const T* __restrict__ Vg = reinterpret_cast<const T*>(p.vt);
    // bf16 fast path (FAST): Q is pre-multiplied by scale*log2(e) when its fragments are loaded, so the MFMA result already is the
    // base-2 logit; full unmasked key tiles then start the S^T accumulator at -m (running reference), which removes the per-element
    // scale-and-subtract, take the row sums from one extra MFMA against a ones operand, and only re-reference m when a tile exceeds
    // it by more than FAST_THR (deferred rescale: any reference works as long as O, l and P share it).
    constexpr bool FAST = std::is_same<T, bf16>::value;
    constexpr bool FAUG = FAST && MASKS;      // the mask k-step variant of the fast tile exists
    constexpr float FAST_THR = 6.0f;
    const float c_pre = p.scale * 1.44269504088896340736f;
    const float c_exp = FAST ? 1.0f : c_pre;                 // softmax in base 2: p = exp2(s*c - m)

    // multi-pass sums live in LDS between passes (touched once per pass), not in registers: [wave][FD][QF][lane] f32x4
    f32x4* totl = reinterpret_cast<f32x4*>(smem + 2 * (KBUF + VBUF)) + wave * (FD * QF * 64) + lane;
    uint8_t* Ms = reinterpret_cast<uint8_t*>(smem + 2 * (KBUF + VBUF) + 4 * FD * QF * 64 * 16);   // FAST: [2][KT] key-mask bytes of the staged tiles
    int nactive = 0, nseen = 0;
    for (int pass = 0; pass < p.npass; ++pass) {
        const AttnEntry& e0 = p.e[pass * ATT_MAXB + b];
        nactive += (e0.w_const != 0.f || e0.w_slope != 0.f) ? 1 : 0;
    }
    const int dup = att_duplicate_pass(p, b, head);      // a self-referencing row's second pass on a head the mask skips: folded into the first
    if (dup >= 0) nactive = 1;
    T* __restrict__ Og = reinterpret_cast<T*>(p.out);
    if (nactive == 0) {   // nothing contributes to this output row: zeros
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            const int q = q0 + f * 16 + l15;
#pragma unroll
            for (int i = 0; i < FD; ++i) {
                const int d = i * 16 + 4 * g;
                float z[4] = {0.f, 0.f, 0.f, 0.f};
                if (q < p.S && d < D) store4(Og + ((long)b * p.S + q) * p.ldo + head * D + d, z);
            }
        }
        return;
    }

    const int ntiles = (p.Sk + KT - 1) / KT;

    for (int pass = 0; pass < p.npass; ++pass) {
        const AttnEntry& en = p.e[pass * ATT_MAXB + b];
        if (en.w_const == 0.f && en.w_slope == 0.f) continue;   // block-uniform skip
        if (pass == dup) continue;
        float w = en.w_const;
        if (p.w_dev) w += en.w_slope * (*p.w_dev);
        if (dup >= 0) w += att_pass_weight(p, p.e[dup * ATT_MAXB + b]);
        const int hb = en.hr_row > 0 ? en.hr_row - 1 : b;   // batch row the reference's j = b*heads + head refers to
        const bool pass_masked = en.kmask && (!(en.flags & ATT_HEAD_RULE) || (((hb * p.heads + head) & 1) == 0));  // block-uniform

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
                if (FAST) {
                    float tmp[EPC];
                    DT<T>::unpack(qf[f][s], tmp);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) tmp[e] *= c_pre;
                    qf[f][s] = DT<T>::pack(tmp);
                }
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
        // FAST: the query side of the mask k-step (see tile_fast) and whether this wave holds a degenerate uniform-softmax query
        u32x4 qaug[QF];
        bool any_uniform = false;
        if constexpr (FAUG) {
            bool u = false;
#pragma unroll
            for (int f = 0; f < QF; ++f) {
                qaug[f] = u32x4{0, 0, 0, 0};
                if (g == 0) {
                    qaug[f][0] = (mode[f] == 1 ? 0x3f80u : 0u) | (mode[f] == 2 ? 0x3f800000u : 0u);   // [wants mask!=0 | wants mask==0]
                    qaug[f][1] = 0x3f80u;                                                              // out-of-range keys: always
                }
                u |= mode[f] == 3;
            }
            any_uniform = __any(u);
        }

        f32x4 o[FD][QF];
        f32x4 lacc[QF];             // FAST: row sums of the fast tiles (ones-operand MFMA), replicated over the lane's 4 registers
        float mrun[QF], lrun[QF];   // running max (already multiplied by c_exp) and per-lane partial row sum
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            mrun[f] = NEG;
            lrun[f] = 0.f;
            lacc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < FD; ++i) o[i][f] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

        // ---- tile staging: global -> registers (issued one tile ahead) -> LDS (double buffered) ----------------------
        u32x4 rk[NKC], rv[NVC];
        uint32_t rmk = 0;           // FAST: mask byte of key k0 + tid of the tile in flight (threads < KT)
        const T* kbase = Kg + (long)en.kv_row * p.Sk * p.ldk + head * D;
        const T* vbase = Vg + ((long)en.kv_row * p.heads * D + head * D) * p.ldvt;
        // GLDS: wave-instruction q (2 per wave for K, 2 for V^T) covers tile rows 8q..8q+7; lane>>3 picks the row, lane&7 the chunk
        // position, the lane fetches the source chunk position ^ swizzle(row); out-of-range rows / chunks come from the zero page
        auto issue_glds = [&](int k0, int buf) {
            if constexpr (GLDS) {
                const char* zp = reinterpret_cast<const char*>(g_att_zero);
                const int lr = lane >> 3, lp = lane & 7;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = 8 * (wave + 4 * i) + lr;               // key row of the K tile
                    const int c = lp ^ (row & 7);
                    const bool ok = k0 + row < p.Sk && c * EPC < D;
                    const char* src = ok ? reinterpret_cast<const char*>(kbase + (long)(k0 + row) * p.ldk + c * EPC) : zp;
                    __builtin_amdgcn_global_load_lds((att_gptr_t)src, (att_lptr_t)(Ks + buf * KBUF + (8 * (wave + 4 * i)) * 128), 16, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = 8 * (wave + 4 * i) + lr;               // d row of the V^T tile
                    const int c = lp ^ ((row >> 1) & 7);
                    const int kk = k0 + c * EPC;
                    const bool ok = row < D && kk < p.Sk;                  // a chunk that starts inside Sk ends inside ldvt (multiple of 8, finite padding)
                    const char* src = ok ? reinterpret_cast<const char*>(vbase + (long)row * p.ldvt + kk) : zp;
                    __builtin_amdgcn_global_load_lds((att_gptr_t)src, (att_lptr_t)(Vs + buf * VBUF + (8 * (wave + 4 * i)) * 128), 16, 0, 0);
                }
            }
        };
        auto issue = [&](int k0) {
            if (FAUG && pass_masked && tid < KT) rmk = (k0 + tid < p.Sk) ? en.kmask[k0 + tid] : 0;
            if (GLDS) return;
#pragma unroll
            for (int i = 0; i < NKC; ++i) {
                const int cid = tid + 256 * i;
                const int key = cid / DCH, c = cid - key * DCH;
                rk[i] = u32x4{0, 0, 0, 0};
                if (k0 + key < p.Sk && c * EPC < D) rk[i] = *reinterpret_cast<const u32x4*>(kbase + (long)(k0 + key) * p.ldk + c * EPC);
            }
#pragma unroll
            for (int i = 0; i < NVC; ++i) {
                const int cid = tid + 256 * i;
                const int d = cid / VCH, c = cid - d * VCH;
                const int kk = k0 + c * EPC;
                rv[i] = u32x4{0, 0, 0, 0};
                if (d < D) {
                    const T* src = vbase + (long)d * p.ldvt + kk;
                    if (kk + EPC <= p.Sk) {
                        rv[i] = *reinterpret_cast<const u32x4*>(src);
                    } else if (kk < p.Sk) {
                        float tmp[EPC];
#pragma unroll
                        for (int e = 0; e < EPC; ++e) tmp[e] = (kk + e < p.Sk) ? DT<T>::ld(src + e) : 0.f;
                        rv[i] = DT<T>::pack(tmp);
                    }
                }
            }
        };
        auto stage = [&](int buf) {
            if (FAUG && pass_masked && tid < KT) Ms[buf * KT + tid] = (uint8_t)rmk;
            if (GLDS) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the tile requested by issue_glds has landed (this wave's share)
                return;
            }
#pragma unroll
            for (int i = 0; i < NKC; ++i) {
                const int cid = tid + 256 * i;
                const int key = cid / DCH, c = cid - key * DCH;
                *reinterpret_cast<u32x4*>(Ks + buf * KBUF + key * KROW + ((KSWZ ? (c ^ (key & 7)) : c) << 4)) = rk[i];
            }
#pragma unroll
            for (int i = 0; i < NVC; ++i) {
                const int cid = tid + 256 * i;
                const int d = cid / VCH, c = cid - d * VCH;
                *reinterpret_cast<u32x4*>(Vs + buf * VBUF + d * VROW + c * 16) = rv[i];
            }
        };

        // ---- one key tile: S^T = K.Q^T, online softmax, O^T += V^T.P^T ---------------------------------------------
        auto tile = [&](int k0, int buf, auto masked_tag) {
            constexpr bool MASKED = decltype(masked_tag)::value;
            const char* Kb = Ks + buf * KBUF;
            const char* Vb = Vs + buf * VBUF;
            f32x4 st[NT][QF];   // st[t][f][r] = score of key 16t+4g+r for query l15 of fragment f
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int f = 0; f < QF; ++f) st[t][f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < DSL; ++s) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int krow = t * 16 + l15;
                    const u32x4 ka = *reinterpret_cast<const u32x4*>(Kb + krow * KROW + ((KSWZ ? ((4 * s + g) ^ (krow & 7)) : (4 * s + g)) << 4));
#pragma unroll
                    for (int f = 0; f < QF; ++f) DT<T>::mma(ka, qf[f][s], st[t][f]);
                }
            }
            // masked path: bit (4t+r) of `inr` = key in range, of `mk` = kmask byte != 0 (keys 16t+4g+r of this lane)
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
            u32x4 pb[NT / KPC][QF];  // packed P^T chunks (B operand of O^T = V^T.P^T)
#pragma unroll
            for (int f = 0; f < QF; ++f) {
                float tmax = NEG;
                uint32_t am = 0;   // allowed keys of this query
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
                    for (int t = 0; t < NT; ++t)
                        tmax = fmaxf(tmax, fmaxf(fmaxf(st[t][f][0], st[t][f][1]), fmaxf(st[t][f][2], st[t][f][3])));
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
                        float pe = __builtin_amdgcn_exp2f(fmaf(st[t][f][r], c_exp, -mnew));   // raw v_exp_f32
                        if (MASKED) pe = ((am >> (4 * t + r)) & 1u) ? pe : 0.f;
                        pv[t][r] = pe;
                        psum += pe;
                    }
                lrun[f] = lrun[f] * alpha + psum;
                if (__any(grew)) {   // lazy rescale: exact, skipped once the running max has settled
#pragma unroll
                    for (int i = 0; i < FD; ++i) o[i][f] *= alpha;
                    if (FAST) lacc[f] *= alpha;
                }
#pragma unroll
                for (int c = 0; c < NT / KPC; ++c) {
                    float tmp[EPC];
#pragma unroll
                    for (int e = 0; e < EPC; ++e) tmp[e] = pv[c * KPC + e / 4][e & 3];
                    pb[c][f] = DT<T>::pack(tmp);
                }
            }
#pragma unroll
            for (int c = 0; c < NT / KPC; ++c) {
#pragma unroll
                for (int i = 0; i < FD; ++i) {
                    u32x4 va;
                    const int vr = i * 16 + l15;
                    const char* vrow = Vb + vr * VROW;
                    if (KPC == 1) {
                        va = *reinterpret_cast<const u32x4*>(vrow + (16 * c + 4 * g) * SZ);
                    } else {
                        const int vsw = GLDS ? ((vr >> 1) & 7) : 0;
                        const u32x2 lo = *(const volatile lds_u32x2_t*)(vrow + (((4 * c + (g >> 1)) ^ vsw) << 4) + 8 * (g & 1));
                        const u32x2 hi = *(const volatile lds_u32x2_t*)(vrow + (((4 * c + 2 + (g >> 1)) ^ vsw) << 4) + 8 * (g & 1));
                        va = u32x4{lo[0], lo[1], hi[0], hi[1]};
                    }
#pragma unroll
                    for (int f = 0; f < QF; ++f) DT<T>::mma(va, pb[c][f], o[i][f]);
                }
            }
        };


        // ---- FAST (bf16): one key tile.  AUG = the tile needs masking (key mask of a masked pass and / or keys past Sk): the mask
        // rides on ONE extra MFMA k-step instead of per-element selects -- K gets three extra "dimensions" per key
        // [mask==0 ? -BIG : 0, mask!=0 ? -BIG : 0, key>=Sk ? -BIG : 0] and Q the matching [wants mask!=0, wants mask==0, 1], so a
        // disallowed score arrives as ~-1e30 and its exp2 is exactly 0 (masked tiles cost 2x an unmasked one with selects).
        auto tile_fast = [&](int k0, int buf, bool first, auto aug_tag) {      // first: tile 0 of the pass (wave-uniform)
            constexpr bool AUG = decltype(aug_tag)::value;
            const char* Kb = Ks + buf * KBUF;
            const char* Vb = Vs + buf * VBUF;
            f32x4 st[NT][QF];
            bool unseen[QF];        // no allowed key of this query met so far (its reference m is still undefined)
#pragma unroll
            for (int f = 0; f < QF; ++f) {
                unseen[f] = AUG ? (mrun[f] == NEG) : first;      // unmasked full tiles: every query meets keys in tile 0
                const float nm = unseen[f] ? 0.f : -mrun[f];
#pragma unroll
                for (int t = 0; t < NT; ++t) st[t][f] = f32x4{nm, nm, nm, nm};
            }
            if constexpr (AUG && FAUG) {
                constexpr uint32_t NB = 0xf14au;                    // bf16(-1e30)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int kl = t * 16 + l15;
                    const bool m1 = pass_masked && Ms[buf * KT + kl] != 0;
                    const bool oor = k0 + kl >= p.Sk;
                    u32x4 kaug = u32x4{0, 0, 0, 0};
                    if (g == 0) {
                        kaug[0] = pass_masked ? (m1 ? (NB << 16) : NB) : 0u;   // [mask==0 -> -BIG | mask!=0 -> -BIG]
                        kaug[1] = oor ? NB : 0u;
                    }
#pragma unroll
                    for (int f = 0; f < QF; ++f) DT<T>::mma(kaug, qaug[f], st[t][f]);
                }
            }
            // all K fragments of the tile are requested before the first MFMA (the compiler otherwise emits read -> wait -> 2 MFMA
            // chains and every LDS latency is exposed); the V^T fragments are requested before the softmax and land under it
            u32x4 ka[DSL][NT];
#pragma unroll
            for (int s = 0; s < DSL; ++s)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int krow = t * 16 + l15;
                    ka[s][t] = *reinterpret_cast<const u32x4*>(Kb + krow * KROW + ((KSWZ ? ((4 * s + g) ^ (krow & 7)) : (4 * s + g)) << 4));
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < DSL; ++s)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int f = 0; f < QF; ++f) DT<T>::mma(ka[s][t], qf[f][s], st[t][f]);
            u32x4 va[NT / KPC][FD];
#pragma unroll
            for (int c = 0; c < NT / KPC; ++c)
#pragma unroll
                for (int i = 0; i < FD; ++i) {
                    const int vr = i * 16 + l15;
                    const char* vrow = Vb + vr * VROW;
                    // bytes 64c + 8g (keys 32c+4g..+3) and +32: chunk 4c + (g>>1) (+2), half g&1; GLDS rows are chunk-swizzled
                    const int vsw = GLDS ? ((vr >> 1) & 7) : 0;
                    const u32x2 lo = *(const volatile lds_u32x2_t*)(vrow + (((4 * c + (g >> 1)) ^ vsw) << 4) + 8 * (g & 1));   // volatile: keep two ds_read_b64 (a merged ds_read2_b64 banks mod 32: 2-way conflicts, half rate)
                    const u32x2 hi = *(const volatile lds_u32x2_t*)(vrow + (((4 * c + 2 + (g >> 1)) ^ vsw) << 4) + 8 * (g & 1));
                    va[c][i] = u32x4{lo[0], lo[1], hi[0], hi[1]};
                }
            __builtin_amdgcn_sched_barrier(0);
            u32x4 pb[NT / KPC][QF];
#pragma unroll
            for (int f = 0; f < QF; ++f) {
                float tm = att_max3(st[0][f][0], st[0][f][1], st[0][f][2]);
                tm = att_max(tm, st[0][f][3]);
#pragma unroll
                for (int t = 1; t < NT; ++t) {
                    tm = att_max3(tm, st[t][f][0], st[t][f][1]);
                    tm = att_max3(tm, st[t][f][2], st[t][f][3]);
                }
                tm = att_max_groups(tm);
                // first allowed key(s) of this query: reference m = tile max; later: only when the tile exceeds m by 2^FAST_THR.
                // A tile whose keys are all disallowed for this query has tm ~ -1e30: nothing happens, its P is exactly 0.
                const bool need = AUG ? (unseen[f] ? (tm > -1e29f) : (tm > FAST_THR)) : (first || tm > FAST_THR);
                if (__any(need)) {            // re-reference: rare after the first tiles
                    const float delta = need ? tm : 0.f;
                    const float alpha = unseen[f] ? 1.f : __builtin_amdgcn_exp2f(-delta);
                    mrun[f] = need ? (unseen[f] ? delta : mrun[f] + delta) : mrun[f];
                    lrun[f] *= alpha;
                    lacc[f] *= alpha;
#pragma unroll
                    for (int i = 0; i < FD; ++i) o[i][f] *= alpha;
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) st[t][f][r] -= delta;
                }
#pragma unroll
                for (int c = 0; c < NT / KPC; ++c) {
                    float tmp[EPC];
#pragma unroll
                    for (int e = 0; e < EPC; ++e) tmp[e] = __builtin_amdgcn_exp2f(st[c * KPC + e / 4][f][e & 3]);
                    pb[c][f] = DT<T>::pack(tmp);
                }
            }
            const u32x4 ones = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
#pragma unroll
            for (int c = 0; c < NT / KPC; ++c) {
#pragma unroll
                for (int f = 0; f < QF; ++f) DT<T>::mma(ones, pb[c][f], lacc[f]);
#pragma unroll
                for (int i = 0; i < FD; ++i)
#pragma unroll
                    for (int f = 0; f < QF; ++f) DT<T>::mma(va[c][i], pb[c][f], o[i][f]);
            }
        };

        issue(0);
        issue_glds(0, 0);
        stage(0);
        __syncthreads();
        for (int t = 0; t < ntiles; ++t) {
            const int k0 = t * KT, buf = t & 1;
            if (t + 1 < ntiles) {
                issue(k0 + KT);
                issue_glds(k0 + KT, buf ^ 1);      // the other buffer was last read in tile t-1, before the barrier that ended it
            }
            if constexpr (FAUG) {
                if (any_uniform)                               // degenerate uniform-softmax queries in this wave: generic tile
                    tile(k0, buf, std::true_type{});
                else if (pass_masked || k0 + KT > p.Sk)
                    tile_fast(k0, buf, t == 0, std::true_type{});
                else
                    tile_fast(k0, buf, t == 0, std::false_type{});
            } else if constexpr (FAST) {
                if (pass_masked || k0 + KT > p.Sk)             // (no masks in this launch: only the ragged last tile comes here)
                    tile(k0, buf, std::true_type{});
                else
                    tile_fast(k0, buf, t == 0, std::false_type{});
            } else {
                if (pass_masked || k0 + KT > p.Sk)
                    tile(k0, buf, std::true_type{});
                else
                    tile(k0, buf, std::false_type{});
            }
            if (t + 1 < ntiles) stage(buf ^ 1);
            __syncthreads();
        }

        // ---- finish this pass: acc = (previous passes) + w * wq[q] * O / l; the last active pass stores to HBM ------------
        ++nseen;
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            float l = lrun[f];
            l += __shfl_xor(l, 16);
            l += __shfl_xor(l, 32);
            if (FAST) l += lacc[f][0];
            const float sc = (l > 0.f) ? (w * wq[f] / l) : 0.f;
            const int q = q0 + f * 16 + l15;
#pragma unroll
            for (int i = 0; i < FD; ++i) {
                f32x4 v = o[i][f] * sc;
                if (nseen > 1) v += totl[(i * QF + f) * 64];
                if (nseen < nactive) {
                    totl[(i * QF + f) * 64] = v;
                } else {
                    const int d = i * 16 + 4 * g;   // lane holds O^T[d = 16i + 4g + r][q = l15] -> 4 consecutive d of one query
                    float vv[4] = {v[0], v[1], v[2], v[3]};
                    if (q < p.S && d < D) store4(Og + ((long)b * p.S + q) * p.ldo + head * D + d, vv);
                }
            }
        }
    }
}
