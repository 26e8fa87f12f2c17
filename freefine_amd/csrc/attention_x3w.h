// attn_x3w_kernel (round 6): the split-bf16 (FFN_BF16X3) self attention of attention_x3p.h re-built around ONE WAVE PER SIMD.
//
// attn_x3p_kernel pairs two waves per SIMD (one in its MFMA segment, one in its softmax segment) on 16x16x32 MFMAs: a 16x16x32 MFMA
// holds the SIMD's vector issue for 8 of its 16 cycles, the split-bf16 softmax (32 v_exp + the P split, ~800 issue cycles per 32
// queries and 64 keys) needs the other 8 -- the issue port is the bound and the matrix pipe sits at 49 % (profiles/r5_pmc_x3_attn_*).
// Here a workgroup is 4 waves (one per SIMD, the whole 512-register file each), a wave owns 64 queries and all products run on the
// 32x32x16 MFMA, which holds the issue port for 8 of its 32 cycles: per 64 keys a wave issues 96 (+ 4 mask) MFMAs = 3072 matrix-pipe
// cycles that leave ~2300 issue cycles for its OWN vector work (64 v_exp, 64 adds, the P split, 32 maxima: ~1700) -- the softmax of
// tile t+1 and the P split of tile t hide in the gaps of the SAME wave's MFMA chain, no partner, no segment barriers:
//     phase 1 of iteration t:  S(t+1) = K(t+1).Q^T  (48 MFMAs + 4 for the key mask)   beside   P(t) -> (P_hi, P_lo)      (192 VALU)
//     phase 2 of iteration t:  O += V^T(t).P(t)     (48 MFMAs)                        beside   max / exp2 / row sums of S(t+1)
// one workgroup barrier per key tile (the K / V^T ring hand-over).  A K / V^T fragment (one ds_read_b128 per image) now feeds two
// 32-query blocks: half the LDS fragment reads per FLOP of the 32-queries-per-wave kernels.
//
// Same arithmetic as attn_x3p_kernel: fp32 q pre-scaled by scale * log2 e and split in registers, K / V^T PRE-SPLIT bf16 images
// (ffn_attn_presplit, staged by LDS-DMA: buffer_load ... lds, swizzle and row permutation on the per-lane SOURCE offset), three bf16
// MFMAs per product term with the small terms first, fp32 softmax with the accumulators started at -m and deferred re-referencing
// (threshold 2^6), the key mask as one exact extra k-step, pass table / tiled-head rule / duplicate-pass folding of attention.h.
// Differences in the arithmetic: row sums are fp32 VALU adds of the un-split probabilities (attn_x3p: ones-MFMAs over P_hi and P_lo).
//
// Fragment geometry (32x32x16: A[i = lane & 31][k = 8 (lane >> 5) + j], B[k][j = lane & 31], D rows (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)):
//   S^T block (kb, qb) = K rows 32 kb .. + 31  x  queries 32 qb .. + 31;  lane (r, h) holds query r, LDS rows rho = (reg & 3) + 8 (reg >> 2) + 4 h.
//   Its registers 8 s' .. 8 s' + 7 are, converted, the B operand of PV k-step s'' = 2 kb + s' -- MFMA k index 8 h + j <-> LDS row
//   16 s' + 8 (j >> 2) + 4 h + (j & 3).  LDS row rho of a K tile holds key pi(rho) = rho with bits 2 and 3 swapped, so that k index
//   8 h + j is key 16 s' + 8 h + j: the V^T fragment is ONE ds_read_b128 of the natural [d][key] image.
//   LDS images: 64 rows x 128 B, 16-byte chunk c of row w at position c ^ ((w >> 1) & 7): the 16 lanes of a ds_read_b128 group hit 16 slots.
//
// Replaces (same call sites as attention_x3p.h): the TCA / plain self-attention launches of the SD UNet in split-bf16 mode
// (/root/reference/src/utils/attention.py:394-404, 1043-1091, 1284-1324).
#pragma once
#include "attention_pp.h"
#include "attention_x3.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) uint8_t x3w_lds_u8_t;

#ifndef X3W_KRING
#define X3W_KRING 4   // K ring slots (32-key blocks): 4 = a block is requested one iteration ahead of its first read (measured faster on the same box: 1368 vs 1401 us, profiles/r6_x3w_ring_depth_and_dma_placement_ab.txt); 6 = two ahead, first fragments read ahead of the barrier
#endif
#ifndef X3W_SPREAD
#define X3W_SPREAD 0  // 1: the iteration's LDS-DMA requests ride the vector-free MFMA gaps of its two steps; 0: all at its start (faster: see the header of the key loop)
#endif
#ifndef X3W_SUMP1
#define X3W_SUMP1 0   // 1: the row sums of a unit's probabilities ride the phase-1 gaps of the step that SPLITS them instead of the phase-2 gaps beside their
                      // exponentials.  Measured SLOWER on one box (profiles/r6_x3w_row_sums_in_phase1_ab.txt: 1461 vs 1412 us two-pass S = 4096, 832 vs 815 one-pass):
                      // the phase-1 gaps (four split instructions + the fragment reads' issue) have no slack either.  Kept for the A/B only.
#endif
#ifndef X3W_PKSUB
#define X3W_PKSUB 0   // 1: the two subtractions of a P-split pair (lo = x - hi) as ONE packed v_pk_add_f32 -- measured SLOWER (1448 vs 1382 us, profiles/r6_x3w_packed_split_sub_ab.txt); A/B only
#endif
#ifndef X3W_ABL
#define X3W_ABL 0     // timing-only ablations (tools/native/x3w_bench.hip): 1 no softmax / split VALU, 2 no LDS-DMA in the loop, 3 no MFMA
#endif

__device__ __forceinline__ void x3w_mma(const u32x4& a, const u32x4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 x3w_mma0(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
// sum / max over the two lanes r, r + 32 that share a query
__device__ __forceinline__ float x3w_max_halves(float x) {
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return att_max(__uint_as_float(a[0]), __uint_as_float(a[1]));
}
__device__ __forceinline__ float x3w_sum_halves(float x) {
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}

#define X3W_SB() __builtin_amdgcn_sched_barrier(0)
#define X3W_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#define X3W_MFMA 0x008
#define X3W_VALU 0x002
#define X3W_TRANS 0x400
#define X3W_DSRD 0x100

template <bool MASKS>
__global__ __launch_bounds__(256) void attn_x3w_kernel(const AttnParams p) {
    constexpr int D = 64, KT = 64;
    constexpr int KBLK = 2 * 4096;                              // one K block slot: 32 rows x 128 B, hi image | lo image
    constexpr int VIMG = 64 * 128, VSLOT = 2 * VIMG;            // one V^T tile slot: 64 rows (d) x 128 B (64 keys), hi image | lo image
    constexpr int NKB = X3W_KRING, AHEAD = X3W_KRING == 6 ? 2 : 1;                                      // K ring: X3W_KRING block slots; a block is requested AHEAD iterations (= 2 AHEAD blocks) ahead of the iteration that reads it
    constexpr int OFF_V = NKB * KBLK, OFF_M = OFF_V + 2 * VSLOT, OFF_TOT = OFF_M + 1024;   // K ring | V^T ring of 2 tiles | key-mask bytes 4 x 256 | multi-pass sums 64 KB
    constexpr int OOB = (int)0x80000000;
    constexpr float FAST_THR = 6.0f;
    constexpr float NEG = -1e30f;
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int nqb = (p.S + 255) / 256;
    const int Lb = (p.heads * p.Bo >= ATT_XCD_MIN_GROUPS) ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int qblk = Lb % nqb, head = (Lb / nqb) % p.heads, b = Lb / (nqb * p.heads);
    const int q0 = qblk * 256 + wave * 64;
    const float* __restrict__ Qg = reinterpret_cast<const float*>(p.q);
    float* __restrict__ Og = reinterpret_cast<float*>(p.out);
    const float c_pre = p.scale * 1.44269504088896340736f;

    f32x4* totl = reinterpret_cast<f32x4*>(smem + OFF_TOT) + wave * (16 * 64) + lane;       // multi-pass sums, wave private: [16 f32x4][64 lanes]
    int nactive = 0, nseen = 0;
    for (int pass = 0; pass < p.npass; ++pass) {
        const AttnEntry& e0 = p.e[pass * ATT_MAXB + b];
        nactive += (e0.w_const != 0.f || e0.w_slope != 0.f) ? 1 : 0;
    }
    const int dup = att_duplicate_pass(p, b, head);
    if (dup >= 0) nactive = 1;
    auto store_out = [&](int q, int d, const float* vv) {
        if (p.out_pair) store_pair_row4(reinterpret_cast<bf16*>(p.out) + ((long)b * p.S + q) * p.ldo, head * D + d, p.ldo / 2, vv);
        else store4(Og + ((long)b * p.S + q) * p.ldo + head * D + d, vv);
    };
    if (nactive == 0) {   // nothing contributes to this output row: zeros (workgroup-uniform, before any barrier)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const int q = q0 + 32 * qb + r;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float z[4] = {0.f, 0.f, 0.f, 0.f};
                if (q < p.S) store_out(q, 32 * (i >> 2) + 8 * (i & 3) + 4 * h, z);
            }
        }
        return;
    }
    const int ntiles = p.Sk / KT;

    // fragment read offsets (image relative): row r of a 32-row block, chunk 2 s + h
    int rd[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) rd[s] = r * 128 + (((2 * s + h) ^ ((r >> 1) & 7)) << 4);
    // LDS-DMA role of this lane: piece `wave` of every 32-row image = rows 8 wave + lr; position pc of a row holds chunk pc ^ ((row >> 1) & 7)
    const int lr = lane >> 3;
    const int dchunk = (lane & 7) ^ (4 * (wave & 1) + (lr >> 1));
    const int dkey = (lr & 3) + 4 * (wave & 1) + 8 * (lr >> 2) + 16 * ((wave >> 1) & 1);    // pi(8 wave + lr): LDS row of a K block -> key of the block
    const int kst = p.ldk * 4, vst = p.ldvt * 4;                                              // bytes from key to key of the K image, from row to row of the V^T image
    const int k_voff = dkey * kst + dchunk * 16;
    const int v_voff = (8 * wave + lr) * vst + dchunk * 16;
    const int mrow = (r & 0x13) | ((r & 4) << 1) | ((r & 8) >> 1);                            // pi(r): the key whose scores LDS row r holds

    for (int pass = 0; pass < p.npass; ++pass) {
        const AttnEntry& en = p.e[pass * ATT_MAXB + b];
        if (en.w_const == 0.f && en.w_slope == 0.f) continue;   // workgroup-uniform skip
        if (pass == dup) continue;
        float w = en.w_const;
        if (p.w_dev) w += en.w_slope * (*p.w_dev);
        if (dup >= 0) w += att_pass_weight(p, p.e[dup * ATT_MAXB + b]);
        const int hb = en.hr_row > 0 ? en.hr_row - 1 : b;
        const bool pass_masked = MASKS && en.kmask && (!(en.flags & ATT_HEAD_RULE) || (((hb * p.heads + head) & 1) == 0));

        // ---- Q^T fragments (B operand: query r, d elements 16 s + 8 h .. + 7): pre-scaled in fp32, then split --------------------
        u32x4 qh[2][4], ql[2][4];
        u32x4 qaug[2];                                          // B operand of the augmenting k-step: [k0 | k1] = selector flags, [k2 | k3] = -(m_hi, m_lo), rest 0
        float wq[2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const int q = q0 + 32 * qb + r;
            const bool qok = q < p.S;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f}, c = a;
                const float* src = Qg + ((long)en.q_row * p.S + q) * p.ldq + head * D + 16 * s + 8 * h;
                if (qok) {
                    a = *reinterpret_cast<const f32x4*>(src);
                    c = *reinterpret_cast<const f32x4*>(src + 4);
                }
                x3_split8(a * c_pre, c * c_pre, qh[qb][s], ql[qb][s]);
                // loop-invariant MFMA B operands live in the accumulator half of the register file (an AGPR-class value is used by the MFMA as it stands;
                // left to the allocator they stay VGPR-class, get parked in AGPRs and are copied back before EVERY use: 128 v_accvgpr_read per 32 keys)
                asm volatile("" : "+a"(qh[qb][s]));
                asm volatile("" : "+a"(ql[qb][s]));
            }
            wq[qb] = (en.wq && qok) ? en.wq[q] : 1.f;
            qaug[qb] = u32x4{0, 0, 0, 0};
            if (MASKS && pass_masked) {
                const int sel = (en.qsel && qok) ? (en.qsel[q] != 0) : 1;
                qaug[qb][0] = sel ? 0x3f80u : 0x3f800000u;      // k index 0: wants mask != 0 (penalise mask == 0) | k index 1: wants mask == 0
            }
        }
        const bool maskon = MASKS && pass_masked && h == 0;

        f32x16 o[2][2];                                         // O^T blocks (db, qb)
        float lacc[2], mrun[2];                                 // mrun = the reference the scores are taken against = m_hi + m_lo of qaug[.][1], exactly
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            mrun[qb] = NEG;
            lacc[qb] = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                o[0][qb][i] = 0.f;
                o[1][qb][i] = 0.f;
            }
        }

        // ---- LDS-DMA of the pre-split images: K by 32-key blocks (block bk = 2 t + kb -> ring slot bk & 3), V^T and the mask bytes by tiles ----
        const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.k), 0, 0x7ffff000, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.vt), 0, 0x7ffff000, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(en.kmask), 0, (MASKS && pass_masked) ? p.Sk : 0, 0x00020000);
        auto dma_kblk = [&](int bk, bool valid, int img) {      // img 0: hi image, 1: lo image
            const int ks = valid ? (en.kv_row * p.Sk + bk * 32) * kst + head * 256 : 0;
            const int vo = valid ? k_voff : OOB;
            char* Kd = smem + (bk % NKB) * KBLK + wave * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (att_lptr_t)(Kd + img * 4096), 16, vo, ks + img * 128, 0, 0);
        };
        const int m_voff = lane < 16 ? lane * 4 : OOB;
        auto dma_mask = [&](int t, bool valid) {
            if (!MASKS) return;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsM, (att_lptr_t)(smem + OFF_M + (t & 3) * 256), 4, valid ? m_voff : OOB, valid ? t * KT : 0, 0, 0);
        };
        auto dma_v = [&](int t, int piece) {                    // V^T tile t (< ntiles) -> slot t & 1; piece = image (hi / lo) + 2 * row half
            const int vs = (en.kv_row * p.heads * D + head * D) * vst + t * 256;
            char* Vd = smem + OFF_V + (t & 1) * VSLOT + wave * 1024;
            const int img = piece & 1, half = piece >> 1;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (att_lptr_t)(Vd + img * VIMG + half * 4096), 16, v_voff, vs + img * 128 + half * 32 * vst, 0, 0);
        };
        auto dma_done = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
        auto lds_done = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };

        // ---- fragment reads: group = (hi, lo) of one (32-row block, 16-deep k-step) = 8 registers feeding 6 MFMAs; four rotating buffers ----
        u32x4 fr[4][2];
        auto rd_k = [&](int koff, auto S_, auto BUF) {          // K block at LDS offset koff, d k-step s
            constexpr int s = decltype(S_)::value, buf = decltype(BUF)::value;
            const char* base = smem + koff + rd[s];
            fr[buf][0] = *reinterpret_cast<const u32x4*>(base);
            fr[buf][1] = *reinterpret_cast<const u32x4*>(base + 4096);
        };
        auto rd_v = [&](int voff, auto DB, auto S2, auto BUF) { // V^T tile at LDS offset voff, d block db, key k-step s2
            constexpr int db = decltype(DB)::value, s2 = decltype(S2)::value, buf = decltype(BUF)::value;
            const char* base = smem + voff + db * 4096 + rd[s2];
            fr[buf][0] = *reinterpret_cast<const u32x4*>(base);
            fr[buf][1] = *reinterpret_cast<const u32x4*>(base + VIMG);
        };
        // A operand of the augmenting k-step of a K block (this lane's LDS row r <-> the block's key pi(r), whose mask byte sits at LDS offset maddr):
        // k index 0: mask == 0 -> -BIG | k index 1: mask != 0 -> -BIG | k indices 2, 3: 1.0 (they carry -(m_hi + m_lo) of the query into the scores).
        // The byte is requested FIRST in a step (LDS returns in order: a late request would make its wait cover the fragment reads behind it) and used last.
        auto mask_byte = [&](int maddr) -> uint32_t {
            if (!MASKS) return 0u;
            return *(const volatile x3w_lds_u8_t*)(smem + maddr);
        };
        auto kaug_of = [&](uint32_t mb) {
            constexpr uint32_t NB = 0xf14au;                    // bf16(-1e30)
            uint32_t flags = mb ? (NB << 16) : NB;
            flags = maskon ? flags : 0u;
            return u32x4{flags, h == 0 ? 0x3f803f80u : 0u, 0u, 0u};
        };
        auto aug_grp = [&](const u32x4& ka, f32x16 (&nx)[2]) {  // the scores relative to the query's reference; masked keys at -1e30
            x3w_mma(ka, qaug[0], nx[0]);
            x3w_mma(ka, qaug[1], nx[1]);
        };
        // six MFMAs of QK^T group s: small terms first
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        auto qk_grp = [&](auto S_, auto BUF, f32x16 (&nx)[2]) {
            constexpr int s = decltype(S_)::value, buf = decltype(BUF)::value;
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                if (X3W_ABL == 3) { asm volatile("" : "+v"(nx[qb]) : "v"(fr[buf][0]), "v"(fr[buf][1])); continue; }
                if (s == 0) nx[qb] = x3w_mma0(fr[buf][1], qh[qb][s], zero16);
                else x3w_mma(fr[buf][1], qh[qb][s], nx[qb]);
                x3w_mma(fr[buf][0], ql[qb][s], nx[qb]);
                x3w_mma(fr[buf][0], qh[qb][s], nx[qb]);
            }
        };
        u32x4 ph[2][2], pl[2][2];                               // split P^T of the current unit: [s'][qb] = B operand of PV k-step 2 kb + s'
        auto pv_grp = [&](auto DB, auto SP, auto BUF) {
            constexpr int db = decltype(DB)::value, sp = decltype(SP)::value, buf = decltype(BUF)::value;
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                if (X3W_ABL == 3) { asm volatile("" : "+v"(o[db][qb]) : "v"(fr[buf][0]), "v"(fr[buf][1]), "v"(ph[sp][qb]), "v"(pl[sp][qb])); continue; }
                x3w_mma(fr[buf][1], ph[sp][qb], o[db][qb]);
                x3w_mma(fr[buf][0], pl[sp][qb], o[db][qb]);
                x3w_mma(fr[buf][0], ph[sp][qb], o[db][qb]);
            }
        };
        // P split of part g = (s', qb): registers 8 s' .. + 7 of block qb -> (P_hi, P_lo)
        auto split_part = [&](auto G, const f32x16 (&cu)[2]) {
            constexpr int g = decltype(G)::value, sp = g >> 1, qb = g & 1;
            if (X3W_ABL == 1) { ph[sp][qb] = u32x4{0x3f803f80u, 0, 0, 0}; pl[sp][qb] = u32x4{0, 0, 0, 0}; return; }
            const f32x16& x = cu[qb];
            x3_split8(f32x4{x[8 * sp], x[8 * sp + 1], x[8 * sp + 2], x[8 * sp + 3]}, f32x4{x[8 * sp + 4], x[8 * sp + 5], x[8 * sp + 6], x[8 * sp + 7]},
                      ph[sp][qb], pl[sp][qb]);
        };
        // maximum of this lane's 16 scores of query block qb, one asm statement (v_max3 on MFMA outputs without canonicalisation)
        float tm[2];
        auto scan = [&](auto QB, const f32x16 (&nx)[2]) {
            constexpr int qb = decltype(QB)::value;
            const f32x16& x = nx[qb];
            float t_;                                           // (asm operands cannot name a captured variable inside a generic lambda)
            asm("v_max3_f32 %0, %1, %2, %3\n\tv_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %0, %0, %6, %7\n\tv_max3_f32 %0, %0, %8, %9\n\t"
                "v_max3_f32 %0, %0, %10, %11\n\tv_max3_f32 %0, %0, %12, %13\n\tv_max3_f32 %0, %0, %14, %15\n\tv_max_f32 %0, %0, %16"
                : "=&v"(t_)
                : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(x[8]), "v"(x[9]), "v"(x[10]), "v"(x[11]),
                  "v"(x[12]), "v"(x[13]), "v"(x[14]), "v"(x[15]));
            tm[qb] = t_;
        };
        // deferred re-referencing (rare: the first unit of a query, and whenever a score exceeds the reference by more than 2^6)
        // Part 1 (at the decision): the new reference, the scores of nx and the augmenting operand.  Part 2 (rescale, below): O and the row sums --
        // DEFERRED to the end of the step in the key loop, because the probabilities of the CURRENT unit (exponentiated against the old reference)
        // are still being multiplied into O while this decision is taken: everything at the old reference is scaled exactly once, after them.
        float alpha_p[2] = {1.f, 1.f};
        auto reref = [&](f32x16 (&nx)[2]) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                const bool unseen = mrun[qb] == NEG;
                const float thr = unseen ? -1e29f : FAST_THR;
                const float tmx = x3w_max_halves(tm[qb]);
                const bool need = tmx > thr;
                const float mraw = unseen ? tmx : mrun[qb] + tmx;
                const uint32_t hb_ = pack_bf16x2(mraw, 0.f) & 0xffffu;                      // the new reference, carried as bf16 hi + lo (17 bits: exact in fp32)
                const float mhi = __uint_as_float(hb_ << 16);
                const uint32_t lb_ = pack_bf16x2(mraw - mhi, 0.f) & 0xffffu;
                const float mnew = mhi + __uint_as_float(lb_ << 16);
                const float delta = need ? (unseen ? mnew : mnew - mrun[qb]) : 0.f;
                alpha_p[qb] = unseen ? 1.f : __builtin_amdgcn_exp2f(-delta);
                mrun[qb] = need ? mnew : mrun[qb];
                qaug[qb][1] = need ? ((hb_ | (lb_ << 16)) ^ 0x80008000u) : qaug[qb][1];
#pragma unroll
                for (int i = 0; i < 16; ++i) nx[qb][i] -= delta;
            }
        };
        auto rescale = [&]() {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                lacc[qb] *= alpha_p[qb];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    o[0][qb][i] *= alpha_p[qb];
                    o[1][qb][i] *= alpha_p[qb];
                }
            }
        };
        auto need_reref = [&]() {
            const float thr0 = (mrun[0] == NEG) ? -1e29f : FAST_THR, thr1 = (mrun[1] == NEG) ? -1e29f : FAST_THR;
            return __builtin_amdgcn_ballot_w64(tm[0] > thr0 || tm[1] > thr1) != 0;
        };
        // exp2 + row sum of flat score index idx (qb = idx >> 4, register idx & 15), in place
        float lpart[2][2];
        auto expsum = [&](auto LO, auto HI, f32x16 (&nx)[2]) {
            constexpr int lo = decltype(LO)::value, hi = decltype(HI)::value;
#pragma unroll
            for (int idx = lo; idx < hi; ++idx) {
                const int qb = idx >> 4, i = idx & 15;
                if (X3W_ABL == 1) continue;
                const float e = __builtin_amdgcn_exp2f(nx[qb][i]);
                nx[qb][i] = e;
                lpart[qb][i & 1] += e;
            }
        };

        f32x16 s0[2], s1[2];                                    // the scores / probabilities of even and odd 32-key units, query blocks 0 / 1
        typedef std::integral_constant<int, 2> I2;
        typedef std::integral_constant<int, 3> I3;

        // ---- the hand-placed streams: every MFMA is followed by the vector work of ITS gap (<= ~24 issue cycles: a 32x32x16 MFMA holds the issue port
        // for 8 of its 32 cycles) and a sched_barrier, so the single wave's in-order stream keeps the matrix pipe fed.  (sched_group_barrier pipelines
        // were tried first: this compiler version leaves six MFMAs back to back and the 24 vector instructions behind them.)
        auto mma_qk = [&](auto S_, auto BUF, auto QB, auto TERM, f32x16 (&nx)[2]) {      // term 0: K_lo.q_hi (starts the chain at s = 0), 1: K_hi.q_lo, 2: K_hi.q_hi
            constexpr int s = decltype(S_)::value, buf = decltype(BUF)::value, qb = decltype(QB)::value, term = decltype(TERM)::value;
            if (X3W_ABL == 3) { asm volatile("" : "+v"(nx[qb]) : "v"(fr[buf][0]), "v"(fr[buf][1])); return; }
            if (term == 0) {
                if (s == 0) nx[qb] = x3w_mma0(fr[buf][1], qh[qb][s], zero16);
                else x3w_mma(fr[buf][1], qh[qb][s], nx[qb]);
            } else if (term == 1) x3w_mma(fr[buf][0], ql[qb][s], nx[qb]);
            else x3w_mma(fr[buf][0], qh[qb][s], nx[qb]);
        };
        auto mma_pv = [&](auto DB, auto SP, auto BUF, auto QB, auto TERM) {               // term 0: V_lo.P_hi, 1: V_hi.P_lo, 2: V_hi.P_hi
            constexpr int db = decltype(DB)::value, sp = decltype(SP)::value, buf = decltype(BUF)::value, qb = decltype(QB)::value, term = decltype(TERM)::value;
            if (X3W_ABL == 3) { asm volatile("" : "+v"(o[db][qb]) : "v"(fr[buf][0]), "v"(fr[buf][1]), "v"(ph[sp][qb]), "v"(pl[sp][qb])); return; }
            if (term == 0) x3w_mma(fr[buf][1], ph[sp][qb], o[db][qb]);
            else if (term == 1) x3w_mma(fr[buf][0], pl[sp][qb], o[db][qb]);
            else x3w_mma(fr[buf][0], ph[sp][qb], o[db][qb]);
        };
        // phase-1 group g: the six MFMAs of d k-step g beside the split of part g = (s' = g >> 1, qb = g & 1) of cu, four instructions per gap
        auto p1_group = [&](auto G, auto&& next_reads, const f32x16 (&cu)[2], f32x16 (&nx)[2], float (&lp)[2][2]) {
            constexpr int g = std::remove_reference_t<decltype(G)>::value, sp = g >> 1, qp = g & 1;
            typedef std::integral_constant<int, g> GT;
            const f32x16& x = cu[qp];
            uint32_t hw[4], lw[4];
            float d0[4], d1[4];
            auto S1 = [&](int i) { hw[i] = pack_bf16x2(x[8 * sp + 2 * i], x[8 * sp + 2 * i + 1]); };
            auto S2 = [&](int i) {
                if (X3W_PKSUB) {
                    typedef float x3w_f32x2 __attribute__((ext_vector_type(2)));
                    const x3w_f32x2 xv = {x[8 * sp + 2 * i], x[8 * sp + 2 * i + 1]};
                    const x3w_f32x2 hv = {__uint_as_float(hw[i] << 16), __uint_as_float(hw[i] & 0xffff0000u)};
                    x3w_f32x2 dv;                                // (written as `xv - hv` the compiler scalarises it again: two v_add_f32_e64 with neg modifiers)
                    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dv) : "v"(xv), "v"(hv));
                    d0[i] = dv[0]; d1[i] = dv[1];
                } else {
                    d0[i] = x[8 * sp + 2 * i] - __uint_as_float(hw[i] << 16); d1[i] = x[8 * sp + 2 * i + 1] - __uint_as_float(hw[i] & 0xffff0000u);
                }
                if (X3W_SUMP1) {                                 // the row sums of cu, two adds per gap, pinned to it (otherwise sunk to the end of the step)
                    lp[qp][0] += x[8 * sp + 2 * i]; lp[qp][1] += x[8 * sp + 2 * i + 1];
                    asm volatile("" : "+v"(lp[qp][0]), "+v"(lp[qp][1]));
                }
            };
            auto S3 = [&](int i) { lw[i] = pack_bf16x2(d0[i], d1[i]); };
            constexpr bool valu = X3W_ABL != 1;
            next_reads();
            mma_qk(GT{}, GT{}, I0{}, I0{}, nx); if (valu) { S1(0); S1(1); S1(2); S1(3); } X3W_SB();
            mma_qk(GT{}, GT{}, I0{}, I1{}, nx); if (valu) S2(0); X3W_SB();
            mma_qk(GT{}, GT{}, I0{}, I2{}, nx); if (valu) S2(1); X3W_SB();
            mma_qk(GT{}, GT{}, I1{}, I0{}, nx); if (valu) S2(2); X3W_SB();
            mma_qk(GT{}, GT{}, I1{}, I1{}, nx); if (valu) S2(3); X3W_SB();
            mma_qk(GT{}, GT{}, I1{}, I2{}, nx); if (valu) { S3(0); S3(1); S3(2); S3(3); }
            if (valu) { ph[sp][qp] = u32x4{hw[0], hw[1], hw[2], hw[3]}; pl[sp][qp] = u32x4{lw[0], lw[1], lw[2], lw[3]}; }
            else { ph[sp][qp] = u32x4{0x3f803f80u, 0, 0, 0}; pl[sp][qp] = u32x4{0, 0, 0, 0}; }
            asm volatile("" : "+v"(ph[sp][qp]), "+v"(pl[sp][qp]));      // P must EXIST here: pure register arithmetic is otherwise sunk to its first use (phase 2)
            X3W_SB();
        };
        // phase-2 group (db, s'): its six MFMAs beside exp2 + row sum of scores [LO, HI) of nx (flat index: qb = idx >> 4, register idx & 15): two
        // v_exp per gap, the adds one gap behind their exponentials
        auto p2_group = [&](auto DB, auto SP, auto BUF, auto LO, auto HI, auto DOEXP, auto&& next_reads, f32x16 (&nx)[2], float (&lp)[2][2]) {
            constexpr int lo = decltype(LO)::value, hi = decltype(HI)::value;
            constexpr bool doexp = decltype(DOEXP)::value != 0 && X3W_ABL != 1;
            typedef decltype(DB) DBT; typedef decltype(SP) SPT; typedef decltype(BUF) BT;
            auto EX = [&](int idx) { if (doexp && idx >= lo && idx < hi) nx[idx >> 4][idx & 15] = __builtin_amdgcn_exp2f(nx[idx >> 4][idx & 15]); };
            auto AD = [&](int idx) { if (!X3W_SUMP1 && doexp && idx >= lo && idx < hi) lp[idx >> 4][idx & 1] += nx[idx >> 4][idx & 15]; };
            // the sums must EXIST at the end of their gap (pure register arithmetic is otherwise sunk to its first use, the end of the step)
            auto PIN = [&]() { if (doexp && !X3W_SUMP1) asm volatile("" : "+v"(lp[0][0]), "+v"(lp[0][1]), "+v"(lp[1][0]), "+v"(lp[1][1])); };
            next_reads();
            mma_pv(DBT{}, SPT{}, BT{}, I0{}, I0{}); EX(lo); EX(lo + 1); X3W_SB();
            mma_pv(DBT{}, SPT{}, BT{}, I0{}, I1{}); EX(lo + 2); EX(lo + 3); AD(lo); AD(lo + 1); PIN(); X3W_SB();
            mma_pv(DBT{}, SPT{}, BT{}, I0{}, I2{}); EX(lo + 4); EX(lo + 5); AD(lo + 2); AD(lo + 3); PIN(); X3W_SB();
            mma_pv(DBT{}, SPT{}, BT{}, I1{}, I0{}); EX(lo + 6); EX(lo + 7); AD(lo + 4); AD(lo + 5); PIN(); X3W_SB();
            mma_pv(DBT{}, SPT{}, BT{}, I1{}, I1{}); EX(lo + 8); EX(lo + 9); AD(lo + 6); AD(lo + 7); PIN(); X3W_SB();
            mma_pv(DBT{}, SPT{}, BT{}, I1{}, I2{}); EX(lo + 10); AD(lo + 8); AD(lo + 9); AD(lo + 10); PIN(); X3W_SB();
        };
        // half of a score maximum (registers 0 .. 7 / 8 .. 15 of block qb), one asm statement of four instructions
        auto scan_a = [&](auto QB, const f32x16 (&nx)[2]) {
            constexpr int qb = decltype(QB)::value;
            const f32x16& x = nx[qb];
            float t_;
            asm("v_max3_f32 %0, %1, %2, %3\n\tv_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %0, %0, %6, %7\n\tv_max_f32 %0, %0, %8"
                : "=&v"(t_) : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
            tm[qb] = t_;
        };
        auto scan_b = [&](auto QB, const f32x16 (&nx)[2]) {
            constexpr int qb = decltype(QB)::value;
            const f32x16& x = nx[qb];
            float t_ = tm[qb];
            asm("v_max3_f32 %0, %0, %1, %2\n\tv_max3_f32 %0, %0, %3, %4\n\tv_max3_f32 %0, %0, %5, %6\n\tv_max3_f32 %0, %0, %7, %8"
                : "+v"(t_) : "v"(x[8]), "v"(x[9]), "v"(x[10]), "v"(x[11]), "v"(x[12]), "v"(x[13]), "v"(x[14]), "v"(x[15]));
            tm[qb] = t_;
        };

        // One pipeline step for unit u = 2 t + AB (cu = its probabilities, already exponentiated; nx = the next unit's scores):
        //   phase 1:  nx = K(block u + 1, at LDS offset koff).Q^T - m  (24 MFMAs + 2 augmenting)   beside   cu -> (P_hi, P_lo)
        //   phase 2:  O += V^T(tile at voff, key k-steps 2 AB, 2 AB + 1).P  (24 MFMAs)             beside   max / exp2 / row sums of nx
        // KPRE: the first two K groups are already in buffers 0 / 1 (requested by the previous step); KNEXT: request the next step's (block at koff_next).
        // Fragment groups rotate through four buffers, each requested two groups (>= 12 MFMAs) ahead of its first use.
        auto step = [&](auto AB_, auto LAST_, auto KPRE_, auto KNEXT_, int koff, int maddr, int voff, int koff_next, auto&& dmaf) {
            constexpr int ab = decltype(AB_)::value;
            constexpr bool last = decltype(LAST_)::value != 0, kpre = decltype(KPRE_)::value != 0, knext = decltype(KNEXT_)::value != 0;
            typedef std::integral_constant<int, 2 * ab> SV0;
            typedef std::integral_constant<int, 2 * ab + 1> SV1;
            auto& cu = ab ? s1 : s0;
            auto& nx = ab ? s0 : s1;
            auto none = [&]() {};
            if (!last) {
                const uint32_t mb = mask_byte(maddr);
                if (!kpre) { rd_k(koff, I0{}, I0{}); rd_k(koff, I1{}, I1{}); }
                if (X3W_SUMP1) lpart[0][0] = lpart[0][1] = lpart[1][0] = lpart[1][1] = 0.f;
                X3W_SB();
                p1_group(I0{}, [&]() { rd_k(koff, I2{}, I2{}); }, cu, nx, lpart);
                p1_group(I1{}, [&]() { rd_k(koff, I3{}, I3{}); }, cu, nx, lpart);
                p1_group(I2{}, [&]() { rd_v(voff, I0{}, SV0{}, I0{}); }, cu, nx, lpart);
                p1_group(I3{}, [&]() { rd_v(voff, I0{}, SV1{}, I1{}); }, cu, nx, lpart);
                // the augmenting k-step closes the chains: scores relative to the query's reference, masked keys at -1e30; the third V^T group is requested here
                rd_v(voff, I1{}, SV0{}, I2{});
                const u32x4 ka = kaug_of(mb);
                if (X3W_ABL != 3) x3w_mma(ka, qaug[0], nx[0]);
                dmaf(I0{}); X3W_SB();                           // (the gaps without vector work carry this step's share of the iteration's LDS-DMA requests)
                if (X3W_ABL != 3) x3w_mma(ka, qaug[1], nx[1]);
                dmaf(I1{}); X3W_SB();
                // phase 2, group 0: the unit maxima ride its MFMAs
                mma_pv(I0{}, I0{}, I0{}, I0{}, I0{}); scan_a(I0{}, nx); X3W_SB();
                mma_pv(I0{}, I0{}, I0{}, I0{}, I1{}); scan_b(I0{}, nx); X3W_SB();
                mma_pv(I0{}, I0{}, I0{}, I0{}, I2{}); scan_a(I1{}, nx); X3W_SB();
                mma_pv(I0{}, I0{}, I0{}, I1{}, I0{}); scan_b(I1{}, nx); X3W_SB();
                mma_pv(I0{}, I0{}, I0{}, I1{}, I1{}); dmaf(I2{}); X3W_SB();
                mma_pv(I0{}, I0{}, I0{}, I1{}, I2{}); dmaf(I3{}); X3W_SB();
                const bool rr = X3W_ABL != 1 && need_reref();
                if (rr) reref(nx);
                if (!X3W_SUMP1) lpart[0][0] = lpart[0][1] = lpart[1][0] = lpart[1][1] = 0.f;
                X3W_SB();
                p2_group(I0{}, I1{}, I1{}, I0{}, std::integral_constant<int, 11>{}, I1{}, [&]() { rd_v(voff, I1{}, SV1{}, I3{}); }, nx, lpart);
                p2_group(I1{}, I0{}, I2{}, std::integral_constant<int, 11>{}, std::integral_constant<int, 22>{}, I1{}, [&]() { if (knext) rd_k(koff_next, I0{}, I0{}); }, nx, lpart);
                p2_group(I1{}, I1{}, I3{}, std::integral_constant<int, 22>{}, std::integral_constant<int, 32>{}, I1{}, [&]() { if (knext) rd_k(koff_next, I1{}, I1{}); }, nx, lpart);
                if (X3W_SUMP1) {                                // lpart = the sums of cu (this unit, at the OLD reference): in before the move
                    lacc[0] += lpart[0][0] + lpart[0][1];
                    lacc[1] += lpart[1][0] + lpart[1][1];
                }
                if (rr) rescale();                              // O and the sums through this unit are complete at the old reference: now they move
                if (!X3W_SUMP1) {                               // lpart = the sums of nx (the next unit, exponentiated at the NEW reference)
                    lacc[0] += lpart[0][0] + lpart[0][1];
                    lacc[1] += lpart[1][0] + lpart[1][1];
                }
            } else {
                if (X3W_SUMP1) {
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb) {
                        float a0 = 0.f, a1 = 0.f;
#pragma unroll
                        for (int i = 0; i < 16; i += 2) { a0 += cu[qb][i]; a1 += cu[qb][i + 1]; }
                        lacc[qb] += a0 + a1;
                    }
                }
                rd_v(voff, I0{}, SV0{}, I0{}); rd_v(voff, I0{}, SV1{}, I1{}); rd_v(voff, I1{}, SV0{}, I2{}); rd_v(voff, I1{}, SV1{}, I3{});
                split_part(I0{}, cu); split_part(I1{}, cu); split_part(I2{}, cu); split_part(I3{}, cu);
                X3W_SB();
                pv_grp(I0{}, I0{}, I0{}); pv_grp(I0{}, I1{}, I1{}); pv_grp(I1{}, I0{}, I2{}); pv_grp(I1{}, I1{}, I3{});
                X3W_SB();
            }
        };

        // ---- prologue: K blocks 0 .. 2 AHEAD, V^T(0), mask bytes of tiles 0 .. AHEAD staged; unit 0's scores and softmax un-overlapped ------------
#pragma unroll
        for (int bk = 0; bk < 2 * AHEAD + 1; ++bk) { dma_kblk(bk, bk < 2 * ntiles, 0); dma_kblk(bk, bk < 2 * ntiles, 1); }
        dma_v(0, 0); dma_v(0, 1); dma_v(0, 2); dma_v(0, 3);
        dma_mask(0, true); dma_mask(1, ntiles > 1);
        if (AHEAD == 2) dma_mask(2, ntiles > 2);
        dma_done();
        attpp_barrier();
        {
            const u32x4 ka = kaug_of(mask_byte(OFF_M + mrow));
            rd_k(0, I0{}, I0{}); rd_k(0, I1{}, I1{}); rd_k(0, I2{}, I2{}); rd_k(0, I3{}, I3{});
            qk_grp(I0{}, I0{}, s0); qk_grp(I1{}, I1{}, s0); qk_grp(I2{}, I2{}, s0); qk_grp(I3{}, I3{}, s0);
            if (X3W_ABL != 3) aug_grp(ka, s0);
            if (AHEAD == 2) { rd_k(KBLK, I0{}, I0{}); rd_k(KBLK, I1{}, I1{}); }   // the first two groups of unit 1's K block, for the first step
            scan(I0{}, s0); scan(I1{}, s0);
            if (X3W_ABL != 1 && need_reref()) { reref(s0); rescale(); }
            lpart[0][0] = lpart[0][1] = lpart[1][0] = lpart[1][1] = 0.f;
            expsum(I0{}, std::integral_constant<int, 32>{}, s0);
            if (!X3W_SUMP1) {                                   // (X3W_SUMP1: unit 0's sums are taken by the step that splits it, like every unit's)
                lacc[0] += lpart[0][0] + lpart[0][1];
                lacc[1] += lpart[1][0] + lpart[1][1];
            }
        }
        if (AHEAD == 1) {
            lds_done();
            attpp_barrier();                                    // every wave has read K block 0: its slot may be re-filled
        }
        typedef std::integral_constant<int, AHEAD == 2> PRE;    // fragments of the next K block requested across the step / iteration boundary

        // Hazards (iteration t = one barrier interval = units 2 t, 2 t + 1): it reads K blocks 2 t + 1, 2 t + 2, the mask bytes of tiles t, t + 1 and V^T(t) -- all
        // complete since the barrier that started it (every wave waits for its own LDS-DMA pieces, vmcnt(0), before the barrier that ends an iteration).
        // Shipped form (ring of 4, AHEAD = 1): at its START it requests K blocks 2 t + 3, 2 t + 4 (the slots of blocks 2 t - 1, 2 t, last read in iteration t - 1),
        // V^T(t + 1) (slot of V^T(t - 1)) and the mask bytes of tile t + 2 (slot of tile t - 2).  Ring of 6 (X3W_KRING = 6, AHEAD = 2; measured slower): blocks
        // 2 t + 5, 2 t + 6 and the mask bytes of tile t + 3, and the first fragments of block 2 t + 3 are read ahead of the barrier that ends the iteration.
        // X3W_SPREAD = 1 puts one request into each vector-free MFMA gap instead of the iteration's start (also slower).  Requests past the end of the keys go out
        // with an out-of-range offset (zeros into a slot nobody reads): no branch in the loop.  No LDS wait at the barrier: every fragment read (except, ring of 6,
        // the four requested for the next iteration) has been consumed by an MFMA.
        int t = 0;
        for (; t + 1 < ntiles; ++t) {
            const int koffA = ((2 * t + 1) % NKB) * KBLK, koffB = ((2 * t + 2) % NKB) * KBLK, koffN = ((2 * t + 3) % NKB) * KBLK, voff = OFF_V + (t & 1) * VSLOT;
            const int maddrA = OFF_M + (t & 3) * 256 + 32 + mrow, maddrB = OFF_M + ((t + 1) & 3) * 256 + mrow;
            const int bA = 2 * t + 2 * AHEAD + 1, bB = bA + 1, tM = t + AHEAD + 1;
            const bool vA = bA < 2 * ntiles, vB = bB < 2 * ntiles;
            if (!X3W_SPREAD && X3W_ABL != 2) {
                dma_kblk(bA, vA, 0); dma_kblk(bA, vA, 1); dma_kblk(bB, vB, 0); dma_kblk(bB, vB, 1);
                dma_v(t + 1, 0); dma_v(t + 1, 1); dma_v(t + 1, 2); dma_v(t + 1, 3);
                dma_mask(tM, tM < ntiles);
            }
            step(I0{}, I0{}, PRE{}, I1{}, koffA, maddrA, voff, koffB, [&](auto I) {
                constexpr int i = decltype(I)::value;
                if (X3W_ABL == 2 || !X3W_SPREAD) return;
                if (i == 0) { dma_kblk(bA, vA, 0); dma_mask(tM, tM < ntiles); }
                else if (i == 1) dma_kblk(bA, vA, 1);
                else if (i == 2) dma_kblk(bB, vB, 0);
                else dma_kblk(bB, vB, 1);
            });
            step(I1{}, I0{}, I1{}, PRE{}, koffB, maddrB, voff, koffN, [&](auto I) {
                if (X3W_ABL == 2 || !X3W_SPREAD) return;
                dma_v(t + 1, decltype(I)::value);
            });
            dma_done();
            if (AHEAD == 1) lds_done();
            attpp_barrier();
        }
        {   // the last tile: unit 2 t + 1 has no successor
            const int koffA = ((2 * t + 1) % NKB) * KBLK, voff = OFF_V + (t & 1) * VSLOT;
            const int maddrA = OFF_M + (t & 3) * 256 + 32 + mrow;
            step(I0{}, I0{}, PRE{}, I0{}, koffA, maddrA, voff, 0, [&](auto) {});
            step(I1{}, I1{}, I0{}, I0{}, 0, 0, voff, 0, [&](auto) {});
            lds_done();
            attpp_barrier();                                    // every fragment read of this pass is retired before the next pass's first DMA
        }

        // ---- finish this pass: acc = (previous passes) + w * wq[q] * O / l; the last active pass stores to HBM ------------
        ++nseen;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const float l = x3w_sum_halves(lacc[qb]);
            const float sc_ = (l > 0.f) ? (w * wq[qb] / l) : 0.f;
            const int q = q0 + 32 * qb + r;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    f32x4 v = f32x4{o[db][qb][4 * g4], o[db][qb][4 * g4 + 1], o[db][qb][4 * g4 + 2], o[db][qb][4 * g4 + 3]} * sc_;
                    const int slot = ((db * 2 + qb) * 4 + g4) * 64;
                    if (nseen > 1) v += totl[slot];
                    if (nseen < nactive) {
                        totl[slot] = v;
                    } else {
                        float vv[4] = {v[0], v[1], v[2], v[3]};
                        if (q < p.S) store_out(q, 32 * db + 8 * g4 + 4 * h, vv);
                    }
                }
        }
    }
}
