// attn_pp_kernel: the "ping-pong" form of the multi-pass masked attention of attention.h for the hot case -- bf16, head dim 64,
// Sk a multiple of 64, no degenerate uniform-softmax entries.  Same maths, same pass table (ffn_attn_desc), same fragment layouts
// and the same FAST softmax (pre-scaled Q, accumulator started at -m, row sums from a ones-operand MFMA, deferred re-referencing,
// key mask folded into one extra MFMA k-step) as attn_kernel<bf16, 64, 2, 64, 2, MASKS>::tile_fast; what changes is the SCHEDULE.
//
// attn_kernel runs 4 waves per workgroup, 2 workgroups per CU; inside a wave a key tile is [K reads, 16 QK^T MFMAs, V reads,
// softmax (~130 VALU incl. 32 v_exp), 20 PV MFMAs, vmcnt(0), barrier]: the MFMA chains sit at both ends of a long VALU stretch, and
// whether the SIMD's other wave fills the gaps is left to chance (measured: 33 % MFMA busy, 743 TFLOP/s at S = 4096).  Here:
//   * 8 waves per workgroup (256 queries, 32 per wave), ONE workgroup per CU; waves 4-7 run one barrier behind waves 0-3, so the
//     two waves of a SIMD alternate: one issues its MFMA segment while the other runs its VALU segment (at s_setprio 1);
//   * the key loop is software-pipelined so that a segment is either all-MFMA or all-VALU:
//       V segment t:  softmax of S(t) -> P(t);  fragment reads of V^T(t) and K(t+1);  LDS-DMA requests of K(t+3), mask(t+3), V^T(t+2)
//       M segment t:  O += V^T(t) P(t) (20 MFMAs);  S(t+1) = K(t+1) Q^T (16 MFMAs, + 8 when the pass carries a key mask)
//   * K / V^T / mask tiles arrive by buffer loads to LDS (scalar tile offsets, no per-lane address arithmetic) into rings of four
//     slots and stay in flight across the barriers; the only wait in the loop is a counted `s_waitcnt vmcnt(3)` (2 without masks);
//   * tiles past the end of the key range are requested with an out-of-range offset (the range check writes zeros, no traffic), so
//     every wave issues the same number of loads in every segment and the counted wait stays exact.
// LDS hazards: as igemm_p8.h -- a wait at the end of V segment q covers fragment reads from V segment q+1 on (both wave groups); a
// slot is re-requested >= 2 segments pairs after its last read (hence rings of four).
//
// Replaces (same call sites as attention.h): the TCA / plain self-attention launches of the SD UNet at S = 256 ... 4096
// (/root/reference/src/utils/attention.py:394-404, 1043-1091, 1284-1324).
#pragma once
#include "attention.h"

#ifndef ATTPP_PRIO
// 2 (shipped): the softmax (V) segment runs at s_setprio 1 and the MFMA segment at 0 -- the MFMA chain is paced by the matrix pipe and
// loses nothing, while a prioritised MFMA wave takes the issue slots the partner's dependent exp -> cvt chains need (measured at
// S = 4096, 16 rows: 353 -> 335 us one pass, 746 -> 729 us two passes; S = 1024: 73.8 -> 70.3 us); 1: the MFMA segment at 1; 0: none
#define ATTPP_PRIO 2
#endif
#ifndef ATTPP_ABL
#define ATTPP_ABL 0   // timing-only ablation builds of tools/native/attn_bench.hip: 1 no LDS-DMA, 2 no counted waits, 3 no softmax, 4 no MFMA, 5 no fragment reads, 6 half the MFMAs
#endif
template <int N>
__device__ __forceinline__ void attpp_wait_vmcnt() {
    if (ATTPP_ABL != 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void attpp_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <bool MASKS>
__global__ __launch_bounds__(512) void attn_pp_kernel(const AttnParams p) {
    typedef bf16 T;
    constexpr int D = 64, KT = 64, QF = 2, NT = 4, FD = 4, DSL = 2, NC = 2;     // NC = 32-key chunks per tile (PV k-steps)
    constexpr int KBUF = KT * 128, VBUF = D * 128, NSLOT = 4;
    constexpr int OFF_V = NSLOT * KBUF, OFF_M = OFF_V + NSLOT * VBUF, OFF_TOT = OFF_M + NSLOT * 256;
    constexpr int OOB = 0x7ffff000;
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
    const T* __restrict__ Qg = reinterpret_cast<const T*>(p.q);
    T* __restrict__ Og = reinterpret_cast<T*>(p.out);
    const float c_pre = p.scale * 1.44269504088896340736f;

    f32x4* totl = reinterpret_cast<f32x4*>(smem + OFF_TOT) + wave * (FD * QF * 64) + lane;     // multi-pass sums, wave private
    int nactive = 0, nseen = 0;
    for (int pass = 0; pass < p.npass; ++pass) {
        const AttnEntry& e0 = p.e[pass * ATT_MAXB + b];
        nactive += (e0.w_const != 0.f || e0.w_slope != 0.f) ? 1 : 0;
    }
    const int dup = att_duplicate_pass(p, b, head);      // a self-referencing row's second pass on a head the mask skips: folded into the first
    if (dup >= 0) nactive = 1;
    if (nactive == 0) {   // nothing contributes to this output row: zeros (workgroup-uniform: no barrier has been executed yet)
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            const int q = q0 + f * 16 + l15;
#pragma unroll
            for (int i = 0; i < FD; ++i) {
                float z[4] = {0.f, 0.f, 0.f, 0.f};
                if (q < p.S) store4(Og + ((long)b * p.S + q) * p.ldo + head * D + i * 16 + 4 * g, z);
            }
        }
        return;
    }
    const int ntiles = p.Sk / KT;

    // loader geometry: wave w requests LDS rows 8w..8w+7 of the K tile and of the V^T tile (d); lane>>3 = row, lane&7 = chunk slot,
    // 16-byte chunks XOR-swizzled by (row & 7) through the source offset.  The K tile's rows are a PERMUTATION of the tile's keys: LDS
    // row R = 16 t + 4 g + r (MFMA fragment t, C-layout row 4g+r of S^T) holds key 32 (t>>1) + 8 g + 4 (t&1) + r, so that after the
    // QK^T MFMAs a lane's eight P values of a 32-key chunk are eight CONSECUTIVE keys and the matching V^T fragment is one
    // ds_read_b128 (16 contiguous bytes of a V^T row) instead of two ds_read_b64.  The permutation costs nothing: the LDS-DMA takes a
    // per-lane source row.
    const int lr = lane >> 3, lp = lane & 7;
    const int k_key = 32 * (wave >> 2) + 8 * (2 * (wave & 1) + (lr >> 2)) + 4 * ((wave >> 1) & 1) + (lr & 3);   // key of LDS row 8w + lr
    const int k_voff = (k_key * p.ldk + ((lp ^ lr) << 3)) * 2;
    const int v_voff = ((8 * wave + lr) * p.ldvt + ((lp ^ lr) << 3)) * 2;
    const int m_voff = lane < 16 ? lane * 4 : OOB;

    // fragment read addresses (tile-slot relative)
    int ka_rd[DSL];
#pragma unroll
    for (int s = 0; s < DSL; ++s) ka_rd[s] = l15 * 128 + (((4 * s + g) ^ (l15 & 7)) << 4);     // fragment t adds t * 2048 (row & 7 = l15 & 7)

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

        // ---- Q^T fragments, pre-scaled by scale * log2(e) ---------------------------------------------------------------
        u32x4 qf[QF][DSL], qaug[QF];
        float wq[QF];
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            const int q = q0 + f * 16 + l15;
            const bool qok = q < p.S;
#pragma unroll
            for (int s = 0; s < DSL; ++s) {
                qf[f][s] = u32x4{0, 0, 0, 0};
                if (qok) qf[f][s] = *reinterpret_cast<const u32x4*>(Qg + ((long)en.q_row * p.S + q) * p.ldq + head * D + (4 * s + g) * 8);
                float tmp[8];
                DT<T>::unpack(qf[f][s], tmp);
#pragma unroll
                for (int e = 0; e < 8; ++e) tmp[e] *= c_pre;
                qf[f][s] = DT<T>::pack(tmp);
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

        // ---- descriptors of this pass: K rows / V^T rows of (kv_row, head), key mask bytes ---------------------------------
        const char* kbase = reinterpret_cast<const char*>(p.k) + ((long)en.kv_row * p.Sk * p.ldk + head * D) * 2;
        const char* vbase = reinterpret_cast<const char*>(p.vt) + ((long)en.kv_row * p.heads * D + head * D) * p.ldvt * 2;
        const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(kbase), 0, OOB, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(vbase), 0, OOB, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(en.kmask), 0, pass_masked ? p.Sk : 0, 0x00020000);
        auto issue_k = [&](int t) {                             // K tile t and its mask bytes -> slot t & 3 (t >= ntiles: zeros)
            const int slot = t & (NSLOT - 1);
            const int so = t < ntiles ? t * KT * p.ldk * 2 : OOB;
            if (ATTPP_ABL == 1) return;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (att_lptr_t)(smem + slot * KBUF + wave * 1024), 16, k_voff, so, 0, 0);
            if (MASKS) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsM, (att_lptr_t)(smem + OFF_M + slot * 256), 4, m_voff, t < ntiles ? t * KT : OOB, 0, 0);
        };
        auto issue_v = [&](int t) {
            const int slot = t & (NSLOT - 1);
            const int so = (t >= 0 && t < ntiles) ? t * KT * 2 : OOB;
            if (ATTPP_ABL == 1) return;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (att_lptr_t)(smem + OFF_V + slot * VBUF + wave * 1024), 16, v_voff, so, 0, 0);
        };

        u32x4 ka[DSL][NT], va[NC][FD], kaug[NT], pb[NC][QF];
        auto read_k = [&](int t) {                              // K fragments (and the mask k-step operand) of tile t
            const char* Kb = smem + (t & (NSLOT - 1)) * KBUF;
#pragma unroll
            for (int s = 0; s < DSL; ++s)
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) ka[s][tt] = *reinterpret_cast<const u32x4*>(Kb + ka_rd[s] + tt * 2048);
            if (MASKS && pass_masked) {
                constexpr uint32_t NB = 0xf14au;                // bf16(-1e30)
                const uint8_t* Mb = reinterpret_cast<const uint8_t*>(smem + OFF_M + (t & (NSLOT - 1)) * 256);
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    const bool m1 = Mb[32 * (tt >> 1) + 8 * (l15 >> 2) + 4 * (tt & 1) + (l15 & 3)] != 0;    // key of LDS row 16 tt + l15
                    kaug[tt] = u32x4{0, 0, 0, 0};
                    if (g == 0) kaug[tt][0] = m1 ? (NB << 16) : NB;   // [mask == 0 -> -BIG | mask != 0 -> -BIG]
                }
            }
        };
        auto read_v = [&](int t) {                              // V^T fragments: rows d = 16 i + l15, chunk 4 c + g (keys 32 c + 8 g ..+7)
            const char* Vb = smem + OFF_V + (t & (NSLOT - 1)) * VBUF;
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int i = 0; i < FD; ++i) va[c][i] = *reinterpret_cast<const u32x4*>(Vb + ka_rd[c] + i * 2048);
        };
        auto mfma_qk = [&]() {                                  // S(t+1) accumulators start at -m (0 while m is undefined)
#pragma unroll
            for (int f = 0; f < QF; ++f) {
                unseen[f] = mrun[f] == NEG;
                const float nm = unseen[f] ? 0.f : -mrun[f];
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) st[tt][f] = f32x4{nm, nm, nm, nm};
            }
            if (MASKS && pass_masked) {
#pragma unroll
                for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                    for (int f = 0; f < QF; ++f) DT<T>::mma(kaug[tt], qaug[f], st[tt][f]);
            }
#pragma unroll
            for (int s = 0; s < DSL; ++s)
#pragma unroll
                for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                    for (int f = 0; f < QF; ++f) {
                        if (ATTPP_ABL == 6 && f) continue;      // timing experiment: half the MFMA issues (results garbage)
                        DT<T>::mma(ka[s][tt], qf[f][s], st[tt][f]);
                    }
        };
        auto mfma_pv = [&]() {
            const u32x4 ones = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
#pragma unroll
            for (int c = 0; c < NC; ++c) {
#pragma unroll
                for (int f = 0; f < QF; ++f) DT<T>::mma(ones, pb[c][f], lacc[f]);
#pragma unroll
                for (int i = 0; i < FD; ++i)
#pragma unroll
                    for (int f = 0; f < QF; ++f) {
                        if (ATTPP_ABL == 6 && f) continue;
                        DT<T>::mma(va[c][i], pb[c][f], o[i][f]);
                    }
            }
        };
        auto softmax = [&]() {                                  // S(t) -> P(t), same arithmetic as attn_kernel::tile_fast
#pragma unroll
            for (int f = 0; f < QF; ++f) {
                // lane-local maximum of the lane's 16 scores of this query: ONE asm statement (hipcc pads every asm statement with a
                // wait state; eight separate v_max3 statements cost eight s_nop)
                float tm;
                asm("v_max3_f32 %0, %1, %2, %3\n\tv_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %0, %0, %6, %7\n\tv_max3_f32 %0, %0, %8, %9\n\t"
                    "v_max3_f32 %0, %0, %10, %11\n\tv_max3_f32 %0, %0, %12, %13\n\tv_max3_f32 %0, %0, %14, %15\n\tv_max_f32 %0, %0, %16"
                    : "=&v"(tm)
                    : "v"(st[0][f][0]), "v"(st[0][f][1]), "v"(st[0][f][2]), "v"(st[0][f][3]), "v"(st[1][f][0]), "v"(st[1][f][1]), "v"(st[1][f][2]),
                      "v"(st[1][f][3]), "v"(st[2][f][0]), "v"(st[2][f][1]), "v"(st[2][f][2]), "v"(st[2][f][3]), "v"(st[3][f][0]), "v"(st[3][f][1]),
                      "v"(st[3][f][2]), "v"(st[3][f][3]));
                // first allowed key(s) of this query: reference m = tile max; later: only when the tile exceeds m by 2^FAST_THR.  A tile
                // whose keys are all disallowed for this query has tm ~ -1e30: nothing happens, its P is exactly 0.  The four lanes of a
                // query agree on m through the cross-group maximum, which is only needed once SOME lane of the wave trips the test.
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
                    float tmp[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) tmp[e] = __builtin_amdgcn_exp2f(st[c * 2 + e / 4][f][e & 3]);
                    pb[c][f] = DT<T>::pack(tmp);
                }
            }
        };

        // ---- prologue: tiles 0, 1 (K, mask, V^T) and K / mask of tile 2 requested; tiles 0 and 1 landed -------------------------
        issue_k(0); issue_v(0);
        issue_k(1); issue_v(1);
        issue_k(2);
        attpp_wait_vmcnt<MASKS ? 2 : 1>();                      // everything but tile 2's K (and mask)
        attpp_barrier();
        attpp_barrier();
        __builtin_amdgcn_sched_barrier(0);
        read_k(0);
        __builtin_amdgcn_sched_barrier(0);
        attpp_barrier();
        mfma_qk();
        if (ATTPP_PRIO == 2) __builtin_amdgcn_s_setprio(1);
        __builtin_amdgcn_sched_barrier(0);
        attpp_barrier();

        for (int t = 0; t < ntiles; ++t) {
            // ---- V segment ----
            __builtin_amdgcn_sched_barrier(0);
            if (ATTPP_ABL != 5) read_v(t);
            if (ATTPP_ABL != 3) softmax();
            else {
#pragma unroll
                for (int f = 0; f < QF; ++f)
#pragma unroll
                    for (int c = 0; c < NC; ++c) asm volatile("" : "+v"(pb[c][f]) : "v"(st[2 * c][f]), "v"(st[2 * c + 1][f]));
            }
            if (ATTPP_ABL != 5) read_k(t + 1);
            issue_k(t + 3);
            issue_v(t + 2);
            attpp_wait_vmcnt<MASKS ? 3 : 2>();                  // K / mask (t+2) and V^T (t+1) have landed: read in the next V segment
            __builtin_amdgcn_sched_barrier(0);
            attpp_barrier();
            // ---- M segment ----
            if (ATTPP_PRIO == 1) __builtin_amdgcn_s_setprio(1);
            if (ATTPP_PRIO == 2) __builtin_amdgcn_s_setprio(0);
            if (ATTPP_ABL != 4) {
                mfma_pv();
                mfma_qk();
            }
            if (ATTPP_PRIO == 1) __builtin_amdgcn_s_setprio(0);
            if (ATTPP_PRIO == 2) __builtin_amdgcn_s_setprio(1);
            __builtin_amdgcn_sched_barrier(0);
            attpp_barrier();
        }
        // drain: nothing of this pass may still be landing when the next pass (or another workgroup's prologue) re-uses the rings,
        // and every fragment read of this pass is retired two barriers before the next request
        __builtin_amdgcn_s_setprio(0);
        attpp_wait_vmcnt<0>();
        attpp_barrier();
        attpp_barrier();

        // ---- finish this pass: acc = (previous passes) + w * wq[q] * O / l; the last active pass stores to HBM ------------
        ++nseen;
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            const float l = lacc[f][0];
            const float sc = (l > 0.f) ? (w * wq[f] / l) : 0.f;
            const int q = q0 + f * 16 + l15;
#pragma unroll
            for (int i = 0; i < FD; ++i) {
                f32x4 v = o[i][f] * sc;
                if (nseen > 1) v += totl[(i * QF + f) * 64];
                if (nseen < nactive) {
                    totl[(i * QF + f) * 64] = v;
                } else {
                    float vv[4] = {v[0], v[1], v[2], v[3]};
                    if (q < p.S) store4(Og + ((long)b * p.S + q) * p.ldo + head * D + i * 16 + 4 * g, vv);
                }
            }
        }
    }
    if (grp == 0) attpp_barrier();                              // balance the lagging group's extra barrier
}
