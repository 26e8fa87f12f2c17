// attn_x3p_kernel: the split-bf16 (FFN_BF16X3) attention of attention_x3.h in the SCHEDULE of attention_pp.h, for the hot case -- head dim
// 64, Sk a multiple of 64, S >= 128, no degenerate uniform-softmax entries.  Same arithmetic as attn_x3_kernel (fp32 q / k / v^T in HBM,
// every operand of both products carried as hi = bf16(x), lo = bf16(x - hi), three bf16 MFMAs per product term, fp32 accumulation, fp32
// softmax) and the same FAST softmax as attn_pp_kernel (Q pre-scaled by scale * log2 e BEFORE its split, S accumulators started at -m,
// row sums from ones-operand MFMAs over P_hi and P_lo, deferred re-referencing, the key mask folded into one extra exact bf16 k-step).
//
// attn_x3_kernel runs its 8 waves in lockstep (one barrier per key tile; [QK^T MFMAs, softmax, P split, PV MFMAs] in every wave at the
// same time): the two waves of a SIMD want the matrix pipe together and the vector ALU together -- 30 % MFMA busy.  Here, as in
// attn_pp_kernel, waves 4-7 run one barrier behind waves 0-3 and the key loop is cut into
//     V segment t:  softmax of S(t) -> P(t) -> (P_hi, P_lo);  fragment reads of V^T(t)'s first key chunk requested
//     M segment t:  O += V^T(t).P(t) (48 MFMAs + 4 for the row sums) with the reads of the second chunk / of K(t+1) requested before / between its halves;
//                   S(t+1) = K(t+1).Q^T (48, + 8 with a key mask);  the fp32 pieces of K(t+2) / V^T(t+2) are requested at its start and,
//                   behind the MFMAs, split and written to LDS
// so that a SIMD always has one wave in each.  With three MFMAs per term the M segment (100-108 MFMAs = 1600-1730 matrix-pipe cycles)
// is the long one: the vector work of the partner's V segment (32 v_exp, the P split, ~800 issue cycles) hides beside it.  Every
// fragment read is requested at least 24 MFMAs before its first use (a first version that left the reads where the compiler put them
// -- next to their MFMAs -- ran 45 % MFMA busy: sixteen exposed LDS latencies per M segment).
// K / V^T tiles are bf16 images [hi | lo] in rings of two (K) and three (V^T) slots; a tile is complete one phase before its first
// read (the segment barrier is the only synchronisation) and overwritten after its last one -- see the hazard table in the loop.  The
// K tile's rows are the same permutation of its keys as in attn_pp_kernel (a lane's eight P values of a 32-key chunk are consecutive
// keys), so a V^T fragment is one ds_read_b128.
//
// Replaces (same call sites as attention_x3.h): the TCA / plain self-attention launches of the SD UNet in split-bf16 mode at
// S = 256 ... 4096 (/root/reference/src/utils/attention.py:394-404, 1043-1091, 1284-1324).
#pragma once
#include "attention_pp.h"
#include "attention_x3.h"

#ifndef X3P_ABL
#define X3P_ABL 0     // timing-only ablation builds of tools/native/x3p_bench.hip: 1 = no in-loop staging (tiles 0 / 1 stay in the rings: results garbage)
#endif
// PAIRKV (round 5): K and V^T arrive PRE-SPLIT -- ffn_attn_presplit writes them once per attention call as bf16 images in exactly the tiles this
// kernel stages: K as [row][key][head][hi(64 d) | lo(64 d)], V^T as [row][head * 64 + d][key tile][hi(64 keys) | lo(64 keys)], every 128-byte
// line wholly hi or lo -- and the tiles go global -> LDS by LDS-DMA (four 1 KiB pieces per wave and tile: K_hi, K_lo, V^T_hi, V^T_lo; the K-row
// permutation and the XOR swizzle are applied to the per-lane SOURCE offset; the key-mask bytes of a tile ride along as a fifth, 4-byte, piece),
// requested at the START of M segment t for tile t + 2 and waited for (vmcnt(0)) at its end.  The in-loop staging of the fp32 form -- four
// global_load_dwordx4 to registers, two 24-instruction splits, four ds_write_b128 per wave and tile, repeated by all 16 query-block workgroups of a
// (row, head) -- cost 16-25 % of the launch (timing ablation, profiles/r5_x3p_staging_ablation.txt); the one-off split pass costs 2-5 %.
template <bool MASKS, bool PAIRKV = false>
__global__ __launch_bounds__(512) void attn_x3p_kernel(const AttnParams p) {
    constexpr int D = 64, KT = 64, QF = 2, NT = 4, FD = 4, DSL = 2, NC = 2;     // NC = 32-key chunks per tile (PV k-steps)
    constexpr int TILE = KT * 128;                              // one bf16 image: 64 rows x 128 B
    constexpr int SLOT = 2 * TILE;                              // hi | lo
    constexpr int OFF_V = 2 * SLOT, OFF_TOT = OFF_V + 3 * SLOT; // K slots 0, 1 | V^T slots 0, 1, 2 | multi-pass sums (144 KB in all)
    constexpr int OFF_M = OFF_TOT + 8 * (FD * QF * 64) * 16;    // PAIRKV: key-mask bytes of the two K tiles in flight, 256 B each
    constexpr int OOB = (int)0x80000000;
    constexpr float FAST_THR = 6.0f;
    constexpr float NEG = -1e30f;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int grp = wave >> 2;                                  // 0: leading waves, 1: lagging waves
    const int nqb = (p.S + 255) / 256;
    const int Lb = (p.heads * p.Bo >= ATT_XCD_MIN_GROUPS) ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int qblk = Lb % nqb, head = (Lb / nqb) % p.heads, b = Lb / (nqb * p.heads);
    const int q0 = qblk * 256 + wave * 32;
    const float* __restrict__ Qg = reinterpret_cast<const float*>(p.q);
    const float* __restrict__ Kg = reinterpret_cast<const float*>(p.k);
    const float* __restrict__ Vg = reinterpret_cast<const float*>(p.vt);
    float* __restrict__ Og = reinterpret_cast<float*>(p.out);
    const float c_pre = p.scale * 1.44269504088896340736f;

    f32x4* totl = reinterpret_cast<f32x4*>(smem + OFF_TOT) + wave * (FD * QF * 64) + lane;     // multi-pass sums, wave private
    int nactive = 0, nseen = 0;
    for (int pass = 0; pass < p.npass; ++pass) {
        const AttnEntry& e0 = p.e[pass * ATT_MAXB + b];
        nactive += (e0.w_const != 0.f || e0.w_slope != 0.f) ? 1 : 0;
    }
    const int dup = att_duplicate_pass(p, b, head);      // a self-referencing row's second pass on a head the mask skips: folded into the first
    if (dup >= 0) nactive = 1;
    auto store_out = [&](int q, int d, const float* vv) {
        if (p.out_pair) store_pair_row4(reinterpret_cast<bf16*>(p.out) + ((long)b * p.S + q) * p.ldo, head * D + d, p.ldo / 2, vv);
        else store4(Og + ((long)b * p.S + q) * p.ldo + head * D + d, vv);
    };
    if (nactive == 0) {   // nothing contributes to this output row: zeros (workgroup-uniform: no barrier has been executed yet)
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            const int q = q0 + f * 16 + l15;
#pragma unroll
            for (int i = 0; i < FD; ++i) {
                float z[4] = {0.f, 0.f, 0.f, 0.f};
                if (q < p.S) store_out(q, i * 16 + 4 * g, z);
            }
        }
        return;
    }
    const int ntiles = p.Sk / KT;

    // staging role of this thread: 8 d elements (chunk sc) of key sk of a K tile, 8 keys (chunk sc) of d row sk of a V^T tile.
    // LDS row of key sk: the permutation of attn_pp_kernel -- row 16 t + 4 g + r holds key 32 (t >> 1) + 8 g + 4 (t & 1) + r.
    const int sk = tid >> 3, sc = tid & 7;
    const int krow = 16 * (2 * (sk >> 5) + ((sk >> 2) & 1)) + 4 * ((sk >> 3) & 3) + (sk & 3);
    const int k_st = krow * 128 + ((sc ^ (krow & 7)) << 4);
    const int v_st = sk * 128 + ((sc ^ (sk & 7)) << 4);
    // fragment read addresses (image relative): row l15 of fragment t / i (+ 2048 each), chunk 4 s + g
    int rd[DSL];
#pragma unroll
    for (int s = 0; s < DSL; ++s) rd[s] = l15 * 128 + (((4 * s + g) ^ (l15 & 7)) << 4);

    if (grp == 1) attpp_barrier();                              // the lagging group starts one barrier late

    for (int pass = 0; pass < p.npass; ++pass) {
        const AttnEntry& en = p.e[pass * ATT_MAXB + b];
        if (en.w_const == 0.f && en.w_slope == 0.f) continue;   // workgroup-uniform skip
        if (pass == dup) continue;
        float w = en.w_const;
        if (p.w_dev) w += en.w_slope * (*p.w_dev);
        if (dup >= 0) w += att_pass_weight(p, p.e[dup * ATT_MAXB + b]);
        const int hb = en.hr_row > 0 ? en.hr_row - 1 : b;
        const bool pass_masked = MASKS && en.kmask && (!(en.flags & ATT_HEAD_RULE) || (((hb * p.heads + head) & 1) == 0));

        // ---- Q^T fragments: pre-scaled in fp32, then split ------------------------------------------------------------------
        u32x4 qh[QF][DSL], ql[QF][DSL], qaug[QF];
        float wq[QF];
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            const int q = q0 + f * 16 + l15;
            const bool qok = q < p.S;
#pragma unroll
            for (int s = 0; s < DSL; ++s) {
                f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f}, c = a;
                const float* src = Qg + ((long)en.q_row * p.S + q) * p.ldq + head * D + 32 * s + 8 * g;
                if (qok) {
                    a = *reinterpret_cast<const f32x4*>(src);
                    c = *reinterpret_cast<const f32x4*>(src + 4);
                }
                x3_split8(a * c_pre, c * c_pre, qh[f][s], ql[f][s]);
            }
            wq[f] = (en.wq && qok) ? en.wq[q] : 1.f;
            qaug[f] = u32x4{0, 0, 0, 0};
            if (MASKS && pass_masked && g == 0) {
                const int sel = (en.qsel && qok) ? (en.qsel[q] != 0) : 1;
                qaug[f][0] = sel ? 0x3f80u : 0x3f800000u;       // [wants mask != 0 | wants mask == 0]
            }
        }

        f32x4 o[FD][QF], lacc[QF], st[NT][QF];
        float mrun[QF];
        bool unseen[QF];                                        // the reference m of S(t+1)'s accumulator start was still undefined
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            mrun[f] = NEG;
            unseen[f] = true;
            lacc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < FD; ++i) o[i][f] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

        // ---- staging: fp32 global -> registers -> split -> bf16 images in LDS ----------------------------------------------------
        const float* kbase = Kg + (long)en.kv_row * p.Sk * p.ldk + head * D + (long)sk * p.ldk + 8 * sc;
        const float* vbase = Vg + ((long)en.kv_row * p.heads * D + head * D + sk) * p.ldvt + 8 * sc;
        f32x4 rk0, rk1, rv0, rv1;
        auto fetch_k = [&](int t) {                             // K tile t (t >= ntiles: nothing is requested, the registers keep stale finite data)
            if (t < ntiles) {
                const float* src = kbase + (long)t * KT * p.ldk;
                rk0 = *reinterpret_cast<const f32x4*>(src);
                rk1 = *reinterpret_cast<const f32x4*>(src + 4);
            }
        };
        auto fetch_v = [&](int t) {
            if (t < ntiles) {
                const float* src = vbase + t * KT;
                rv0 = *reinterpret_cast<const f32x4*>(src);
                rv1 = *reinterpret_cast<const f32x4*>(src + 4);
            }
        };
        auto stage_k = [&](int t) {
            u32x4 hi, lo;
            x3_split8(rk0, rk1, hi, lo);
            char* B = smem + (t & 1) * SLOT + k_st;
            *reinterpret_cast<u32x4*>(B) = hi;
            *reinterpret_cast<u32x4*>(B + TILE) = lo;
        };
        auto stage_v = [&](int slot) {
            u32x4 hi, lo;
            x3_split8(rv0, rv1, hi, lo);
            char* B = smem + OFF_V + slot * SLOT + v_st;
            *reinterpret_cast<u32x4*>(B) = hi;
            *reinterpret_cast<u32x4*>(B + TILE) = lo;
        };
        // mask bytes of the four keys this lane's K fragments hold in tile t (LDS row 16 tt + l15), packed into one word
        uint32_t mbytes = 0;
        auto fetch_mask = [&](int t) {
            if (MASKS && pass_masked && t < ntiles) {
                if constexpr (PAIRKV) {                         // the tile's 64 mask bytes came by LDS-DMA beside its K images (slot t & 1)
                    const uint8_t* mb = reinterpret_cast<const uint8_t*>(smem + OFF_M + (t & 1) * 256) + 8 * (l15 >> 2) + (l15 & 3);
                    mbytes = (uint32_t)mb[0] | ((uint32_t)mb[4] << 8) | ((uint32_t)mb[32] << 16) | ((uint32_t)mb[36] << 24);
                } else {
                    const uint8_t* mb = en.kmask + t * KT + 8 * (l15 >> 2) + (l15 & 3);
                    mbytes = (uint32_t)mb[0] | ((uint32_t)mb[4] << 8) | ((uint32_t)mb[32] << 16) | ((uint32_t)mb[36] << 24);     // tt = 0 .. 3: key 32 (tt >> 1) + 4 (tt & 1) + ...
                }
            }
        };
        // PAIRKV: this wave's four pieces of a tile (LDS rows 8 wave .. 8 wave + 7 of each image).  LDS row r, 16-byte position c' holds chunk
        // c' ^ (r & 7) of key (K) / of d row (V^T) r; K rows are the key permutation row 16 tq + 4 gq + rr <-> key 32 (tq >> 1) + 8 gq + 4 (tq & 1) + rr
        const int dr = 8 * wave + (lane >> 3), dc = ((lane & 7) ^ (dr & 7)) * 16;
        const int dkey = 32 * (dr >> 5) + 8 * ((dr >> 2) & 3) + 4 * ((dr >> 4) & 1) + (dr & 3);
        const int k_voff = dkey * p.ldk * 4 + dc, v_voff = dr * p.ldvt * 4 + dc;      // (ldk * 4 / ldvt * 4: bytes from key to key / from row to row of the images)
        const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.k), 0, PAIRKV ? 0x7ffff000 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.vt), 0, PAIRKV ? 0x7ffff000 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(en.kmask), 0, (PAIRKV && MASKS && pass_masked) ? p.Sk : 0, 0x00020000);
        auto dma_tile = [&](int t, int vs) {                    // tile t -> K slot t & 1, V^T slot vs (t >= ntiles: nothing)
            if (!PAIRKV || t >= ntiles) return;
            const int ks = (en.kv_row * p.Sk + t * KT) * p.ldk * 4 + head * 256;
            const int vsoff = (en.kv_row * p.heads * D + head * D) * p.ldvt * 4 + t * 256;
            char* Kd = smem + (t & 1) * SLOT + wave * 1024;
            char* Vd = smem + OFF_V + vs * SLOT + wave * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (att_lptr_t)Kd, 16, k_voff, ks, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (att_lptr_t)(Kd + TILE), 16, k_voff, ks + 128, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (att_lptr_t)Vd, 16, v_voff, vsoff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (att_lptr_t)(Vd + TILE), 16, v_voff, vsoff + 128, 0, 0);
            if (MASKS) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsM, (att_lptr_t)(smem + OFF_M + (t & 1) * 256), 4, lane < 16 ? lane * 4 : OOB, t * KT, 0, 0);
        };
        auto dma_done = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

        // Fragment reads are software-pipelined by hand in groups of four ds_read_b128 (hi and lo images of two fragments: 16 registers)
        // that feed 12 MFMAs; two register buffers alternate, every group is requested while the previous one's MFMAs run.
        u32x4 fa[4], fb[4];
        auto rd_k = [&](int t, int sl, int th, u32x4* F) {      // K fragments tt = 2 th, 2 th + 1 of d-slab sl: F = [hi, lo, hi, lo]
            const char* Kh = smem + (t & 1) * SLOT + rd[sl] + th * 4096;
            F[0] = *reinterpret_cast<const u32x4*>(Kh);
            F[1] = *reinterpret_cast<const u32x4*>(Kh + TILE);
            F[2] = *reinterpret_cast<const u32x4*>(Kh + 2048);
            F[3] = *reinterpret_cast<const u32x4*>(Kh + TILE + 2048);
        };
        auto rd_v = [&](int slot, int c, int ih, u32x4* F) {    // V^T fragments i = 2 ih, 2 ih + 1 of key chunk c
            const char* Vh = smem + OFF_V + slot * SLOT + rd[c] + ih * 4096;
            F[0] = *reinterpret_cast<const u32x4*>(Vh);
            F[1] = *reinterpret_cast<const u32x4*>(Vh + TILE);
            F[2] = *reinterpret_cast<const u32x4*>(Vh + 2048);
            F[3] = *reinterpret_cast<const u32x4*>(Vh + TILE + 2048);
        };
        auto qk_start = [&]() {                                 // S accumulators start at -m (0 while m is undefined); the key mask as one exact k-step
#pragma unroll
            for (int f = 0; f < QF; ++f) {
                unseen[f] = mrun[f] == NEG;
                const float nm = unseen[f] ? 0.f : -mrun[f];
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) st[tt][f] = f32x4{nm, nm, nm, nm};
            }
            if (MASKS && pass_masked) {
                constexpr uint32_t NB = 0xf14au;                // bf16(-1e30)
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    u32x4 kaug = u32x4{0, 0, 0, 0};
                    const bool m1 = ((mbytes >> (8 * tt)) & 0xffu) != 0;
                    if (g == 0) kaug[0] = m1 ? (NB << 16) : NB;   // [mask == 0 -> -BIG | mask != 0 -> -BIG]
#pragma unroll
                    for (int f = 0; f < QF; ++f) x3_mma(kaug, qaug[f], st[tt][f]);
                }
            }
        };
        auto qk_group = [&](auto SL, auto TH, const u32x4* F) {
            constexpr int sl = decltype(SL)::value, th = decltype(TH)::value;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int f = 0; f < QF; ++f) {
                    x3_mma(F[2 * j + 1], qh[f][sl], st[2 * th + j][f]);      // small terms first
                    x3_mma(F[2 * j], ql[f][sl], st[2 * th + j][f]);
                    x3_mma(F[2 * j], qh[f][sl], st[2 * th + j][f]);
                }
        };
        u32x4 ph[NC][QF], pl[NC][QF];
        auto pv_group = [&](auto C, auto IH, const u32x4* F) {
            constexpr int c = decltype(C)::value, ih = decltype(IH)::value;
            if (ih == 0) {
                const u32x4 ones = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
#pragma unroll
                for (int f = 0; f < QF; ++f) {
                    x3_mma(ones, pl[c][f], lacc[f]);
                    x3_mma(ones, ph[c][f], lacc[f]);
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int f = 0; f < QF; ++f) {
                    x3_mma(F[2 * j + 1], ph[c][f], o[2 * ih + j][f]);
                    x3_mma(F[2 * j], pl[c][f], o[2 * ih + j][f]);
                    x3_mma(F[2 * j], ph[c][f], o[2 * ih + j][f]);
                }
        };
        typedef std::integral_constant<int, 0> I0;
        typedef std::integral_constant<int, 1> I1;
#define X3P_SB() __builtin_amdgcn_sched_barrier(0)
        auto softmax = [&]() {                                  // S(t) -> P(t) -> (P_hi, P_lo); same arithmetic as attn_pp_kernel's
#pragma unroll
            for (int f = 0; f < QF; ++f) {
                float tm;
                asm("v_max3_f32 %0, %1, %2, %3\n\tv_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %0, %0, %6, %7\n\tv_max3_f32 %0, %0, %8, %9\n\t"
                    "v_max3_f32 %0, %0, %10, %11\n\tv_max3_f32 %0, %0, %12, %13\n\tv_max3_f32 %0, %0, %14, %15\n\tv_max_f32 %0, %0, %16"
                    : "=&v"(tm)
                    : "v"(st[0][f][0]), "v"(st[0][f][1]), "v"(st[0][f][2]), "v"(st[0][f][3]), "v"(st[1][f][0]), "v"(st[1][f][1]), "v"(st[1][f][2]),
                      "v"(st[1][f][3]), "v"(st[2][f][0]), "v"(st[2][f][1]), "v"(st[2][f][2]), "v"(st[2][f][3]), "v"(st[3][f][0]), "v"(st[3][f][1]),
                      "v"(st[3][f][2]), "v"(st[3][f][3]));
                const float thr = unseen[f] ? -1e29f : FAST_THR;
                if (__builtin_amdgcn_ballot_w64(tm > thr) != 0) {
                    tm = att_max_groups(tm);
                    const bool need = tm > thr;
                    const float delta = need ? tm : 0.f;
                    const float alpha = unseen[f] ? 1.f : __builtin_amdgcn_exp2f(-delta);
                    mrun[f] = need ? (unseen[f] ? delta : mrun[f] + delta) : mrun[f];
                    lacc[f] *= alpha;
#pragma unroll
                    for (int i = 0; i < FD; ++i) o[i][f] *= alpha;
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) st[tt][f][r] -= delta;
                }
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    f32x4 e0, e1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        e0[r] = __builtin_amdgcn_exp2f(st[2 * c][f][r]);
                        e1[r] = __builtin_amdgcn_exp2f(st[2 * c + 1][f][r]);
                    }
                    x3_split8(e0, e1, ph[c][f], pl[c][f]);
                }
            }
        };
        auto lds_done = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };

        // ---- prologue: tiles 0 and 1 staged; S(0) computed ------------------------------------------------------------------------
        if constexpr (PAIRKV) {
            dma_tile(0, 0);
            dma_tile(1, 1);
            dma_done();
        } else {
            fetch_k(0); fetch_v(0);
            fetch_mask(0);
            stage_k(0); stage_v(0);
            fetch_k(1); fetch_v(1);
            stage_k(1); stage_v(1);
            lds_done();
        }
        attpp_barrier();
        attpp_barrier();                                        // both groups' pieces of tiles 0 / 1 are in LDS
        if constexpr (PAIRKV) fetch_mask(0);
        X3P_SB();
        rd_k(0, 0, 0, fa); rd_k(0, 0, 1, fb);
        qk_start();
        qk_group(I0{}, I0{}, fa); X3P_SB(); rd_k(0, 1, 0, fa); X3P_SB();
        qk_group(I0{}, I1{}, fb); X3P_SB(); rd_k(0, 1, 1, fb); X3P_SB();
        qk_group(I1{}, I0{}, fa);
        qk_group(I1{}, I1{}, fb);
        fetch_mask(1);
        __builtin_amdgcn_s_setprio(1);
        X3P_SB();
        attpp_barrier();

        // Hazards (phase = barrier interval; the leading group runs V segment t in phase 2t and M segment t in 2t + 1, the lagging group
        // one phase later).  Tile u is fetched at the start and written at the end of M segment u - 2 (phases 2u - 3, 2u - 2): K(u) into
        // K slot u & 1, whose previous tile u - 2 was last read in M segment u - 3 (phases 2u - 5, 2u - 4); V^T(u) into V^T slot u % 3,
        // whose previous tile u - 3 was last read in M segment u - 3.  First reads: K(u) in M segment u - 1 (phase 2u - 1), V^T(u) at the
        // end of V segment u (phase 2u) -- after both groups' writes.
        // Registers: the fp32 staging pieces live only inside an M segment (where S is dead), the first V^T group is requested when
        // the softmax has consumed S -- the V segment, whose S + P + O + Q already fill most of the 256 registers, holds nothing else.
        int vslot = 0;                                          // V^T slot of tile t
        for (int t = 0; t < ntiles; ++t) {
            const int vnext = vslot == 2 ? 0 : vslot + 1, vwr = vnext == 2 ? 0 : vnext + 1;     // slots of tiles t + 1, t + 2
            // ---- V segment ----
            X3P_SB();
            softmax();
            // P must EXIST before the barrier: it is pure register arithmetic, which neither the barrier nor sched_barrier orders -- left
            // alone, hipcc sinks the 32 v_exp and the whole split next to their first use, i.e. into the M segment (measured: the V
            // segment empty, 45 % MFMA busy)
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int f = 0; f < QF; ++f) asm volatile("" : "+v"(ph[c][f]), "+v"(pl[c][f]));
            X3P_SB();
            rd_v(vslot, 0, 0, fa);                              // lands while the wave waits at the barrier
            X3P_SB();
            attpp_barrier();
            // ---- M segment: 8 groups of 12 (+ 4) MFMAs, the next group's fragments requested behind each ----
            __builtin_amdgcn_s_setprio(0);
            const bool more = t + 1 < ntiles;
            if constexpr (PAIRKV) {
                if (X3P_ABL != 1) dma_tile(t + 2, vwr);
            } else if (X3P_ABL != 1) {
                fetch_k(t + 2);
                fetch_v(t + 2);
            }
            rd_v(vslot, 0, 1, fb); X3P_SB();
            pv_group(I0{}, I0{}, fa); X3P_SB(); rd_v(vslot, 1, 0, fa); X3P_SB();
            pv_group(I0{}, I1{}, fb); X3P_SB(); rd_v(vslot, 1, 1, fb); X3P_SB();
            pv_group(I1{}, I0{}, fa); X3P_SB();
            if (more) rd_k(t + 1, 0, 0, fa);
            X3P_SB();
            pv_group(I1{}, I1{}, fb); X3P_SB();
            if (more) {
                rd_k(t + 1, 0, 1, fb); X3P_SB();
                qk_start();
                qk_group(I0{}, I0{}, fa); X3P_SB(); rd_k(t + 1, 1, 0, fa); X3P_SB();
                qk_group(I0{}, I1{}, fb); X3P_SB(); rd_k(t + 1, 1, 1, fb); X3P_SB();
                qk_group(I1{}, I0{}, fa); X3P_SB();
                qk_group(I1{}, I1{}, fb);
            }
            X3P_SB();
            if constexpr (PAIRKV) {
                dma_done();                                     // this wave's pieces of tile t + 2 have landed (they had the whole segment)
                fetch_mask(t + 2);                              // (every wave brought the tile's mask bytes itself: no barrier between its DMA and this read)
            } else {
                if (X3P_ABL != 1 && t + 2 < ntiles) {
                    stage_k(t + 2);
                    stage_v(vwr);
                }
                fetch_mask(t + 2);
            }
            lds_done();
            __builtin_amdgcn_s_setprio(1);
            X3P_SB();
            attpp_barrier();
            vslot = vnext;
        }
#undef X3P_SB
        // drain: every fragment read of this pass is retired two barriers before the next pass's (or workgroup's) first LDS write
        __builtin_amdgcn_s_setprio(0);
        attpp_barrier();
        attpp_barrier();

        // ---- finish this pass: acc = (previous passes) + w * wq[q] * O / l; the last active pass stores to HBM ------------
        ++nseen;
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            const float l = lacc[f][0];
            const float sc_ = (l > 0.f) ? (w * wq[f] / l) : 0.f;
            const int q = q0 + f * 16 + l15;
#pragma unroll
            for (int i = 0; i < FD; ++i) {
                f32x4 v = o[i][f] * sc_;
                if (nseen > 1) v += totl[(i * QF + f) * 64];
                if (nseen < nactive) {
                    totl[(i * QF + f) * 64] = v;
                } else {
                    float vv[4] = {v[0], v[1], v[2], v[3]};
                    if (q < p.S) store_out(q, i * 16 + 4 * g, vv);
                }
            }
        }
    }
    if (grp == 0) attpp_barrier();                              // balance the lagging group's extra barrier
}

// ---- ffn_attn_presplit: fp32 K / V^T -> the pre-split bf16 images attn_x3p_kernel<., PAIRKV = true> stages by LDS-DMA ------------------------
// K  [R][Sk][ldk] (heads * 64 columns)   -> bf16 [R][Sk][heads][hi(64) | lo(64)]
// V^T [R][heads * 64][ldvt] (Sk % 64 = 0) -> bf16 [R][heads * 64][Sk / 64][hi(64 keys) | lo(64 keys)]
// one thread = 8 values (32 B in, 2 x 16 B out); HBM-bound, once per attention call (the 16 query-block workgroups of a (row, head) used to repeat it)
__global__ __launch_bounds__(256) void attn_presplit_k_kernel(const float* __restrict__ k, bf16* __restrict__ out, long n, int heads, int ldk) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c8 = (int)(i & 7);
        const long rh = i >> 3;
        const int h = (int)(rh % heads);
        const long rs = rh / heads;
        const float* src = k + rs * ldk + h * 64 + c8 * 8;
        u32x4 hi, lo;
        x3_split8(*reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + 4), hi, lo);
        bf16* dst = out + rh * 128 + c8 * 8;
        *reinterpret_cast<u32x4*>(dst) = hi;
        *reinterpret_cast<u32x4*>(dst + 64) = lo;
    }
}
__global__ __launch_bounds__(256) void attn_presplit_vt_kernel(const float* __restrict__ vt, bf16* __restrict__ out, long n, int ntile, int ldvt) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c8 = (int)(i & 7);
        const long rt = i >> 3;
        const int t = (int)(rt % ntile);
        const long rc = rt / ntile;
        const float* src = vt + rc * ldvt + t * 64 + c8 * 8;
        u32x4 hi, lo;
        x3_split8(*reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + 4), hi, lo);
        bf16* dst = out + rt * 128 + c8 * 8;
        *reinterpret_cast<u32x4*>(dst) = hi;
        *reinterpret_cast<u32x4*>(dst + 64) = lo;
    }
}
