// igemm_pp_kernel: the 256-row "ping-pong" member of the implicit-GEMM family (bf16 only, same operands as igemm.h).
//
// Why another structure.  The 2-stage kernels of igemm.h wait `vmcnt(0)` + barrier once per 64-deep K tile: every K tile starts with
// an empty load pipe, and the two waves of a SIMD run their MFMA chains and their LDS reads in lockstep.  Measured (profiles/,
// round 1): 39-46 % MFMA busy, ~50 % of wave cycles parked in s_waitcnt / s_barrier, the L2->LDS path idle between stages.  This
// kernel is built the way cdna_hip_programming.md section 5 ("256^2 8-phase template", T3+T4+T5) describes:
//   * BM x BN output tile (BM = 256 or 192 -- the tile HEIGHT is a launch-time choice so that the tile count lands on a multiple of
//     the 256 CUs: M = 98304 gives 384 tiles of 256 rows = 1.5 rounds, but 512 of 192 = 2 full rounds --, BN = 256 or 320), 8 waves
//     as 2 (M) x 4 (N), BM/2 x BN/4 outputs per wave (96-160 accumulator registers), one workgroup per CU, LDS = two whole K tiles
//     (2 x (BM + BN) x 128 B <= 144 KiB) + 12 KiB of column vectors;
//   * a K tile (64 deep) is consumed in FOUR phases (k-substep 0: rows 0-63 of the wave, rows 64-127; k-substep 1: same), 16 or 20
//     MFMAs each; the B fragments of a k-substep stay in registers for its two phases;
//   * the next K tile's operands are requested in the SAME four phases (B, B, A rows {0-63,128-191}, A rows {64-127,192-255}: 2-3
//     LDS-DMA loads per wave and phase) and stay in flight ACROSS the barriers: the only waits are two COUNTED
//     `s_waitcnt vmcnt(N)` per K tile, never 0 in steady state;
//   * waves 4-7 run the same program ONE BARRIER behind waves 0-3 (they share SIMDs pairwise): while one wave of a SIMD issues its
//     MFMA chain (s_setprio 1) the other one reads fragments from LDS and issues the loads of its phase;
//   * persistent over output tiles with the K-tile stream running straight across tile boundaries (the first K tile of the next
//     output tile is requested while the last one of the current tile is multiplied);
//   * NO global load in the epilogue: bias and the time-embedding row bias of the NEXT output tile arrive by LDS-DMA one K tile
//     ahead (wave-private slots), the accumulators START at bias + row bias (+ residual, loaded once at the top of the tile), and
//     the epilogue only converts and stores.  (In igemm_epilogue every fragment's bias / residual loads wait `vmcnt(0)` behind the
//     previous fragment's stores -- the counter is in order --, which with 40 fragments per wave and two wave groups taking turns
//     cost ~50 us per 256 x 320 tile: 154 us of a 385 us launch.)  Rounding: fl(bias + sum) instead of fl(fl(sum) + bias).
//
// LDS hazards (cdna_hip_programming.md section 5, "Read a staged buffer one phase AFTER the wait that retires it"):
//   RAW  a counted wait placed at the END of the load section of phase q (before its first barrier) covers fragment reads from
//        phase q+1 on, for both wave groups (the lagging group's wait precedes the same barrier event the leading group's reads
//        follow).  Waits: end of phase 4 (B and A-low of the next K tile landed: read in phase 1), end of phase 1 (A-high landed:
//        read in phase 2).
//   WAR  a slot is re-staged >= 2 phases after the last phase that reads it: B is last read in phase 3 and re-staged in phases
//        1-2 of the next K tile; A-low last read in phase 3, re-staged in phase 3; A-high last read in phase 4, re-staged in phase 4.
//
// Operand layout in LDS: rows of 128 B (64 bf16 of K), 16-byte chunks XOR-swizzled by (row & 7) through the SOURCE offset of the
// direct-to-LDS loads (same image as igemm_glds_kernel): conflict-free ds_read_b128 fragment reads.
//
// Loads: BUFFER loads to LDS (`buffer_load_dwordx4 ... offen lds`).  The lane's 32-bit byte offset (voffset) says WHICH row / pixel
// it fetches and changes only per output tile (dense A, W) or per conv tap; the position along K is a SCALAR offset (soffset)
// stepped by 128 B per K tile -- no per-lane pointer arithmetic in the K loop.  Lanes whose offset lies outside the descriptor (rows
// past M, and the zero padding of a convolution, encoded as offset 0x80000000) get ZEROS written to LDS by the range check (probed
// on gfx950: tools/native/probe_bufload_lds.hip), so there is no zero page and no select.
//
// Restrictions (checked by the launcher, capi.hip): bf16; K % 64 == 0 and K >= 128 (conv: Cin % 64 == 0); N % BN == 0; alpha == 1;
// flags subset of {GEGLU (BN = 256)}; rowbias only with rows_per_batch >= 128; every operand < 2 GiB.
#pragma once
#include <type_traits>
#include "igemm.h"

#ifndef PP_PRIO
#define PP_PRIO 1     // 1: MFMA sections at s_setprio 1 (shipped); 0: no priorities; 2: load sections at s_setprio 1
#endif
#ifndef PP_ABL
#define PP_ABL 0      // timing-only ablation builds of tools/native/pp_bench.hip: 1 = no global loads, 2 = no counted waits, 3 = GEGLU without the GELU, 4 = no epilogue stores (results garbage)
#endif
#ifndef PP_TAPINNER
#define PP_TAPINNER 1     // split-bf16 convolutions: K walk chunk-major (taps innermost); 0 = tap-major (A/B builds of tools/native/x3_bench.hip)
#endif
#ifdef PP_STAMP
// tools/native/x3_bench.hip -DPP_STAMP: shader-clock stamps (s_memtime) of one workgroup's waves at the 8 segment boundaries of 4 consecutive stages,
// collected in LDS (behind the kernel's own 2 BUF + 12 KiB) and copied out at the end -- where a stage's cycles go.  Never defined in the library.
__device__ unsigned long long pp_stamp_out[8 * 64];
#define PP_STAMP_AT(k_) do { if (blockIdx.x == 7 && s >= 24 && s < 28) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    if (lane == 0) *reinterpret_cast<unsigned long long*>(smem + CV + 12288 + (wave * 64 + (s - 24) * 16 + (k_)) * 8) = t_; } } while (0)
#else
#define PP_STAMP_AT(k_) do {} while (0)
#endif
#ifdef PP_NBE
#define X3_NBE(FN_, NA_) (PP_NBE)
#else
#define X3_NBE(FN_, NA_) ((FN_) + 2 - ((FN_) + (NA_) + 1) / 2)      // early W pieces per wave and stage of the split-bf16 core (see its loop)
#endif
template <int N>
__device__ __forceinline__ void pp_wait_vmcnt() {
    if (PP_ABL != 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void pp_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// SPLIT: split-K launch for problems with too few output tiles to fill the chip (the 8x8 / 16x16 levels of the UNet: M = 3072,
// K up to 23040).  A "tile" of the persistent walk then is (output tile, K slice): slice s of `splitk` multiplies K tiles
// [s * nk / splitk, (s+1) * nk / splitk) and stores its raw fp32 accumulators to slab s of the workspace (p.ws, [splitk][M][N]);
// igemm_splitk_reduce_kernel (igemm.h) sums the slabs and applies the epilogue (bias, row bias, residual, SiLU, f32 output).
// TRANS: transposed output out[b][n][s] (m = b * rows_per_batch + s; the V^T operand of ffn_attn), bias only: the MFMA operands swap
// roles so that a lane holds four consecutive rows m of one column n -> 8-byte stores along s.
#ifdef PP_TRACE
__device__ unsigned long long pp_trace[256 * 64];   // tools/native/pp_bench.hip: 100 MHz timestamps around the low-row stores of each tile
#endif
// X3 (FFN_BF16X3, split-bf16 operands; round 5: "core v2").  Operands arrive in the BLOCKED pair form (include/freefine_hip.h): every 32 elements of
// the contraction are one 128-byte block [hi(32) | lo(32)] -- of an A row / pixel and of a W row alike -- so to the LOADER this is simply a bf16
// GEMM over 2 K elements: a K tile ("stage") is 32 real elements deep, its LDS row holds the hi fragment chunks (16-byte chunks 0-3) and the lo ones
// (chunks 4-7) of the same 32 elements, every 128-byte line is fetched whole and ONCE, and nothing of the K position logic is special.
// The MULTIPLIER differs: a stage is consumed in TWO phases (low rows, high rows) instead of four, and each phase issues the three products
//     A_hi x W_hi,   A_lo x W_hi,   A_hi x W_lo
// of its FH x FN fragment pairs back to back from ONE set of fragment reads (a_hi, a_lo of the phase's rows; w_hi, w_lo read in phase L and kept
// in registers for phase H; the 256 x 320 tile, 160 accumulator registers, holds one A set and fetches a_lo over a_hi behind the second product).  Against round 3/4's form (the bf16 kernel run over a virtual contraction of 3 K in
// chunk order: three K tiles = 12 phases = 24 barriers and six fragment sets per 64 elements) that is 4 phases = 8 barriers and four fragment sets
// per 64 elements, with 45-60 MFMAs (720-960 cycles) per phase beside the partner wave's load section instead of 16-20.
// LDS-DMA schedule per stage and wave: phase L requests the next stage's W pieces and its low A pieces, phase H its high A pieces; waits: end of
// phase H (W + A-low of the next stage landed, the high A pieces just requested may fly), end of phase L (A-high landed) -- every piece has a full
// phase to land.  WAR: the other buffer's W and low A rows were last read in phase L of the previous stage (two phases back), its high A rows in
// phase H of the previous stage (two phases back from this stage's phase H).
// K slices of a split-K launch are whole stages.  Output and residual are fp32: accumulators start at bias + row bias + fp32 residual, the epilogue
// stores 16 bytes per lane through the same lane permutation as the split-K slabs; GEGLU writes the blocked pair form for the next GEMM.
// F8 (FFN_FP8, 3x3 convolutions): fp8 e4m3 operands in the bf16 byte geometry (the library passes a bf16-shaped view: K, Cin, Kpad in
// two-byte units), so a K tile carries 128 real elements and each fragment pair takes TWO fp8 MFMAs -- twice the MFMA work per LDS-DMA
// piece of a kernel whose bound is the issue of those pieces.  Operands are pre-scaled by powers of two (activations 2^4, weights per
// tensor); the accumulators start at (bias + row bias + residual) / alpha and the epilogue multiplies by alpha = 1 / (scale product).
template <int BM, int BN, int AMODE, bool RES, bool GEGLU, bool SPLIT = false, bool TRANS = false, bool X3 = false, bool F8 = false>
__global__ __launch_bounds__(512) void igemm_pp_kernel(const IgemmParams p, int splitk = 1) {
    typedef bf16 T;
    constexpr int HM = BM / 2;              // rows per wave (128 or 96)
    constexpr int FM = HM / 16;             // 16-row fragments per wave (8 or 6), FH of them per phase
    constexpr int FH = FM / 2;
    constexpr int NA = BM / 64;             // A pieces per wave and K tile (4 or 3): 2 requested in phase 3, NA - 2 in phase 4
    constexpr int FN = BN / 64;             // 16-column fragments per wave (4 or 5; 2 on the 256 x 128 tile)
    constexpr int WN = BN / 4;              // columns per wave
    constexpr int ABYTES = BM * 128, BBYTES = BN * 128, BUF = ABYTES + BBYTES;
    constexpr int CV = 2 * BUF;             // column vectors: per wave [bias | row bias of the first image | of the next image] x 128 floats
    constexpr int NB1 = (FN + 1) / 2;       // B pieces requested in phase 1 (the rest in phase 2)
    // store instructions of one half-tile epilogue (every one is issued: no lane predicate around them); bf16 row-major output goes
    // out as pairs of fragments (store_rows)
    // X3 GEGLU writes the bf16 PAIR form (hi and lo planes: two 8-byte stores per fragment pair) -- its only consumer is the next GEMM
    constexpr int NST = (SPLIT || TRANS || X3) ? FH * FN : (GEGLU ? FH * FN / 2 : FH * ((FN + 1) / 2));
    constexpr int OSZ = (X3 && !GEGLU) ? 4 : 2;         // bytes per output / residual element
    constexpr int OOB = (int)0x80000000;
    // (BN = 128, round 6: the VAE's 128-channel 3x3 convolutions in split-bf16 mode -- two column fragments per wave, launched by capi.hip for X3 convolutions only)
    static_assert((BM == 256 || BM == 192) && (BN == 256 || BN == 320 || (BN == 128 && BM == 256 && X3 && AMODE == AMODE_CONV3 && !GEGLU && !SPLIT && !TRANS)), "tiles built for this kernel");
    static_assert(!GEGLU || FN % 2 == 0, "GEGLU pairs hidden / gate column blocks inside a wave");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;          // waves 0-3: upper half of the rows (leading group), waves 4-7: lower half (lagging group)
    const int l15 = lane & 15, g = lane >> 4;
    const int lrow = lane >> 3, csrc = (lane & 7) ^ lrow;

    static_assert(!SPLIT || (!RES && !GEGLU), "split-K slabs carry raw accumulators: the epilogue runs in the reduce kernel");
    static_assert(!TRANS || (!RES && !GEGLU && !SPLIT && AMODE == AMODE_DENSE), "transposed output: dense A, bias only");
    static_assert(!F8 || (!X3 && !TRANS && !GEGLU), "fp8 operands: plain / residual / split-K epilogues");
    const int ntn = p.N / BN;
    const int ntm = (p.M + BM - 1) / BM;
    const int nsl = SPLIT ? splitk : 1;               // K slices per output tile (the launcher picks a divisor of K / 64)
    const int ntiles = ntm * ntn * nsl;               // walk index = output tile * nsl + slice
    const int G = gridDim.x;
    const int nk = (X3 ? p.K / 96 : p.K / 64) / nsl;  // K tiles per walk step (X3: p.K is the virtual 3 K, a K tile = 32 real elements = one 128-byte [hi | lo] block)
    const int first = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, G));
    const int my_tiles = (ntiles - first + G - 1) / G;
    const int S = my_tiles * nk;                      // K tiles this workgroup multiplies, in stream order

    // ---- descriptors ----------------------------------------------------------------------------------------------------------
    const int PIX = X3 ? p.lda * 2 : p.Cin * 2;        // conv: bytes per input pixel (pair format: hi and lo planes side by side)
    const long a_bytes = AMODE == AMODE_DENSE ? (long)p.M * p.lda * 2 : (long)(p.M / (p.Hout * p.Wout)) * p.Hin * p.Win * PIX;
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, (int)((long)p.N * p.Kpad * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcBias = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.bias ? p.N * 4 : 0, 0x00020000);
    const int nbatch = (p.M + p.rows_per_batch - 1) / p.rows_per_batch;
    const __amdgpu_buffer_rsrc_t rsrcRb =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.rowbias), 0, p.rowbias ? ((nbatch - 1) * p.ldrb + p.N) * 4 : 0, 0x00020000);
    const long o_bytes = TRANS ? (long)((p.M + p.rows_per_batch - 1) / p.rows_per_batch) * p.N * p.ldo * OSZ : (long)p.M * p.ldo * OSZ;
    const __amdgpu_buffer_rsrc_t rsrcO = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)o_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcR = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.residual), 0, RES ? (int)((long)p.M * p.ldr * OSZ) : 0, 0x00020000);

    // ---- loader: runs one K tile ahead of the multiplier along the stream (tile, kt) ------------------------------------------
    // A pieces (8 rows x 128 B = one wave-instruction).  The rows each wave multiplies in phases 1 and 3 ("low": the first HM/2 rows
    // of either wave group) must have landed one phase earlier than the rest ("high"), so every low piece is requested in phase 3:
    //   BM = 256: wave w requests pieces {w, 16+w} (low) in phase 3 and {8+w, 24+w} (high) in phase 4;
    //   BM = 192: low = pieces 0-5 and 12-17, high = 6-11 and 18-23; phase 3 takes the 12 low ones and 4 high ones, phase 4 the rest.
    // B pieces: rows 8(w + 8i)..
    int l_tile = first, l_kt = 0;
    int l_k0 = 0;                                     // first K tile of the loader's K slice (split-K launches)
    int a_off[NA];                                    // dense: byte offset of this lane's chunk at K = 0; conv: the same for tap (0,0), ignoring the image border
    int a_yx[NA];                                     // conv: packed (y << 16 | x & 0xffff) of the tap-(0,0) input pixel
    int b_voff;
    const int b_step = 64 * p.Kpad * 2;
    int apiece[NA];
    if constexpr (BM == 256) {
        apiece[0] = wave; apiece[1] = 16 + wave; apiece[2] = 8 + wave; apiece[3] = 24 + wave;
    } else {
        apiece[0] = wave < 6 ? wave : wave + 6;
        apiece[1] = wave < 4 ? 14 + wave : 2 + wave;
        apiece[2] = wave < 2 ? 10 + wave : 16 + wave;
    }
    // The position along K is a pure SCALAR function of the loader's K-tile counter (no loop-carried tap state: with `if (ka >= Cin)`
    // style updates hipcc moved the whole tap state into VGPRs and put a v_readfirstlane + hazard nops, or a waterfall loop, in
    // front of every load).  cpt = 64-channel chunks per tap; tap = l_kt / cpt by a 16-bit reciprocal (exact for l_kt * cpt < 65536).
    // K order of a convolution: tap-major (all 64-channel chunks of tap 0, then tap 1, ...: every activation row and every weight row
    // is streamed front to back).  The nine taps re-read the activations, and the per-XCD footprint between two taps (32 CUs x 256
    // pixels x Cin x 2 B = 5.2 MB at Cin = 320) exceeds the 4 MiB L2: the dominant launch fetches 1.15 GB for 127 MB of activations
    // (profiles/r2_pmc_*; 379 MB with the 192-row tile).  Two walks that keep the re-reads in L2 were built and measured -- chunk-major
    // (the nine taps of a chunk back to back) and (ky, chunk, kx) -- and both run 5-13 % SLOWER (310 / 313 vs 289 us at M = 196608,
    // Cin = 320; 373 vs 339 us at 32x32 x 640): the launch is not bound by where its lines come from, and the strided walks lose more
    // (requests to lines still in flight, weight rows no longer streamed) than the L2 hits return.
    // X3: a pixel's Cin channels are Cin / 32 blocks [hi(32) | lo(32)] of 128 bytes = Cin / 32 K tiles per tap (PIX = 4 Cin bytes per pixel)
    const int cpt = AMODE == AMODE_DENSE ? 1 : (X3 ? p.Cin / 32 : p.Cin / 64);      // K tiles per tap
    const int cpt_rcp = (65536 + cpt - 1) / cpt;
    // Split-bf16 convolutions walk K CHUNK-major (round 5): the taps of one 32-channel block back to back, then the next block.  With three MFMAs per
    // product the stage's load sections, not its MFMA sections, set the pace (in-kernel stamps, profiles/r5_x3_stamps.txt: 990 + 780 cycles of
    // load sections beside 2 x 720 of MFMA on the 192 x 320 tile), and what they wait for is the issue of LDS-DMA pieces whose lines come from
    // beyond L2: tap-major, the nine taps re-read a tile's pixels 10 stages apart and the per-XCD footprint between two taps exceeds the 4 MiB L2
    // (40 % of the launch's requests miss it: TCC_HIT 1.5e7 / MISS 1.0e7, FETCH_SIZE = 9 x the activations); chunk-major, eight of the nine taps
    // find their lines where the previous stage left them.  (Round 2 measured the same walk 5-13 % SLOWER on the bf16 kernel: there the MFMA section
    // was a third as long and the loader's per-stage tap arithmetic was the pole.)  W rows keep their tap-major packing -- only the walk changes:
    // stage kt reads W columns (tap * cpt + chunk) * 64.
    constexpr bool TAPIN = X3 && AMODE != AMODE_DENSE && PP_TAPINNER;
    const int ntap = p.conv == 2 ? 4 : 9;
    auto kt_split = [&](int kt, int& tap, int& c) {
        if (TAPIN) {
            c = p.conv == 2 ? (kt >> 2) : ((kt * 7282) >> 16);      // kt / 9: exact below 32768
            tap = kt - c * ntap;
        } else {
            tap = (kt * cpt_rcp) >> 16;
            c = kt - tap * cpt;
        }
    };
    int ka = 0, tap_ky = 0, tap_kx = 0, tap_off = 0;  // of K tile l_kt; set by k_position()
    auto k_position = [&]() {
        const int kt = l_k0 + l_kt;
        if (AMODE == AMODE_DENSE) {
            ka = kt * 128;
        } else {
            int tap, c;
            kt_split(kt, tap, c);
            ka = c * 128;
            tap_ky = p.conv == 2 ? (tap >> 1) : ((tap * 21846) >> 16);       // taps per window row: 2 (conv == 2) or 3
            tap_kx = tap - (p.conv == 2 ? 2 : 3) * tap_ky;
            tap_off = (tap_ky * p.Win + tap_kx) * PIX;
        }
    };
    auto w_position = [&]() {                         // byte offset of the loader's K tile inside a W row
        const int kt = l_k0 + l_kt;
        if (!TAPIN) return kt * 128;
        int tap, c;
        kt_split(kt, tap, c);
        return (tap * cpt + c) * 128;
    };
    auto prep = [&](int wtile) {                      // loader state at the first K tile of walk step `wtile`
        const int tile = wtile / nsl;
        l_k0 = (wtile - tile * nsl) * nk;
        const int mt = (tile / ntn) * BM, nt = (tile % ntn) * BN;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int m = mt + 8 * apiece[i] + lrow;
            if (AMODE == AMODE_DENSE) {
                a_off[i] = m < p.M ? (PP_ABL == 10 ? m & 255 : m) * (p.lda * 2) + csrc * 16 : OOB;      // 10: every tile reads the same 256 rows (L2-hot A; results garbage)
            } else {
                const int hw = p.Hout * p.Wout;
                const int ma = PP_ABL == 10 ? (m & 255) : m;       // 10: every tile reads the same 256 pixels (L2-hot A; results garbage)
                const int b = ma / hw, rem = ma - b * hw;
                const int yo = rem / p.Wout, xo = rem - yo * p.Wout;
                // conv == 2 (2x2 taps, one sub-pixel class of a 3x3 conv on a nearest-2x upsampled input): the window starts pad_y = pad >> 1
                // rows above and pad_x = pad & 1 columns left of the output pixel
                const int y0 = yo * p.stride - (p.conv == 2 ? (p.pad >> 1) : p.pad), x0 = xo * p.stride - (p.conv == 2 ? (p.pad & 1) : p.pad);       // in (nearest-2x upsampled) input coordinates
                a_off[i] = ((b * p.Hin + (y0 >> p.upsample)) * p.Win + (x0 >> p.upsample)) * PIX + csrc * 16;
                a_yx[i] = m < p.M ? ((y0 << 16) | (x0 & 0xffff)) : (int)0x80008000;       // rows past M: every tap out of the image
            }
        }
        b_voff = (nt + 8 * wave + lrow) * (p.Kpad * 2) + csrc * 16;
    };
    auto issue_a = [&](int i, int buf) {
        int voff = a_off[i];
        if (AMODE != AMODE_DENSE) {
            const int y0 = a_yx[i] >> 16, x0 = (int)(short)(a_yx[i] & 0xffff);
            const int yy = y0 + tap_ky, xx = x0 + tap_kx;
            const bool inb = (unsigned)yy < (unsigned)(p.Hin << p.upsample) && (unsigned)xx < (unsigned)(p.Win << p.upsample);
            int toff = tap_off;
            if (p.upsample) {     // fused nearest-2x upsample: source pixel (yy >> 1, xx >> 1); the step from tap (0,0) depends on the parity of y0 / x0
                const int dy = (tap_ky + (y0 & 1)) >> 1, dx = (tap_kx + (x0 & 1)) >> 1;
                toff = (dy * p.Win + dx) * PIX;
            }
            voff = inb ? voff + toff : OOB;
        }
        if (PP_ABL != 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lptr_t)(smem + buf * BUF + apiece[i] * 1024), 16, voff, ka, 0, 0);
    };
    auto issue_b = [&](int i, int buf) {
        const int so = w_position() + i * b_step;     // (a lambda call inside the builtin's argument list made hipcc 7.2 drop the whole kernel without a diagnostic)
        if (PP_ABL != 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lptr_t)(smem + buf * BUF + ABYTES + (wave + 8 * i) * 1024), 16, b_voff, so, 0, 0);
    };
    // column vectors of output tile `tile` for THIS wave: bias[n0 + wc*WN ..) and the row bias of the (at most two: the launcher
    // requires rows_per_batch >= HM) images the wave's HM rows belong to, 2 x 64 floats each (lanes past WN fetch nothing: zeros),
    // into the wave's private 1.5 KiB slot.  6 loads per wave.
    auto issue_colvec = [&](int tile) {
        if (SPLIT) return;                            // slabs carry raw sums: no column vectors
        const int mt = (tile / ntn) * BM + wr * HM, nt = (tile % ntn) * BN + wc * WN;
        const int bb = mt / p.rows_per_batch;
        char* slot = smem + CV + wave * 1536;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = h * 64 + lane;
            const int vb = c < WN ? (nt + c) * 4 : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcBias, (lptr_t)(slot + h * 256), 4, vb, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcRb, (lptr_t)(slot + 512 + h * 256), 4, vb, bb * p.ldrb * 4, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcRb, (lptr_t)(slot + 1024 + h * 256), 4, vb, (bb + 1) * p.ldrb * 4, 0, 0);
        }
    };
    int switched = 0;
    auto advance = [&]() {                            // after the last piece of a K tile was requested: step the stream
        ++l_kt;
        switched = 0;
        if (l_kt == nk) {
            l_kt = 0;
            l_tile += G;
            if (l_tile < ntiles) {
                prep(l_tile);
                issue_colvec(l_tile);
                switched = 1;
            }
        }
    };

    // ---- multiplier state ------------------------------------------------------------------------------------------------
    const int swz = l15 & 7;
    const int a_rd0 = (wr * HM + l15) * 128 + ((g ^ swz) << 4);            // k-substep 0; fragment i adds i * 2048
    const int a_rd1 = (wr * HM + l15) * 128 + (((4 + g) ^ swz) << 4);
    const int b_rd0 = ABYTES + (wc * WN + l15) * 128 + ((g ^ swz) << 4);
    const int b_rd1 = ABYTES + (wc * WN + l15) * 128 + (((4 + g) ^ swz) << 4);

    f32x4 acc[FM][FN];
    u32x4 fa[FH], fb[FN];

    auto read_a = [&](int buf, int rd, int i0) {
#pragma unroll
        for (int i = 0; i < FH; ++i) fa[i] = *reinterpret_cast<const u32x4*>(smem + buf * BUF + rd + (i0 + i) * 2048);
    };
    auto read_b = [&](int buf, int rd) {
#pragma unroll
        for (int j = 0; j < FN; ++j) fb[j] = *reinterpret_cast<const u32x4*>(smem + buf * BUF + rd + j * 2048);
    };
    auto mfma_rows = [&](auto I0) {
        constexpr int i0 = decltype(I0)::value;
        if (PP_PRIO == 1) __builtin_amdgcn_s_setprio(1);
        if (PP_PRIO == 2) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int i = 0; i < FH; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                if constexpr (F8) mma_fp8(fb[j], fa[i], acc[i0 + i][j]);
                else if constexpr (TRANS) DT<T>::mma(fa[i], fb[j], acc[i0 + i][j]);      // C[m = 4g + r][n = l15]
                else DT<T>::mma(fb[j], fa[i], acc[i0 + i][j]);                      // C[m = l15][n = 4g + r]
            }
        if (PP_PRIO == 1) __builtin_amdgcn_s_setprio(0);
        if (PP_PRIO == 2) __builtin_amdgcn_s_setprio(1);
    };
    typedef std::integral_constant<int, 0> I0_t;
    typedef std::integral_constant<int, FH> I4_t;

    // accumulators of a new output tile start at bias + row bias (+ residual): lane holds C[m = 16 i + l15][n = 16 j + 4 g + r]
    // The epilogue of a tile is split into its low and high row halves: the low rows are final after phase 3 of the tile's last K
    // tile and are stored (and re-started for the next tile) in that K tile's phase-4 load section; the high rows are stored and
    // re-started in the phase-2 load section of the next tile's first K tile -- each half beside an MFMA phase of the other wave
    // group, instead of both groups running their whole epilogue back to back with the matrix pipe idle.
    const float f8_inv = F8 ? 1.0f / p.alpha : 1.0f;       // alpha is a power of two: exact
    auto init_rows = [&](int tile, auto I0) {
        constexpr int i0 = decltype(I0)::value;
        if constexpr (SPLIT) {
#pragma unroll
            for (int i = i0; i < i0 + FH; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            return;
        }
        const char* slot = smem + CV + wave * 1536;
        if constexpr (TRANS) {                        // bias of column n = 16 j + l15, the same for the lane's four rows
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const float bv = *reinterpret_cast<const float*>(slot + (j * 16 + l15) * 4);
#pragma unroll
                for (int i = i0; i < i0 + FH; ++i) acc[i][j] = f32x4{bv, bv, bv, bv};
            }
            return;
        }
        const int m0 = (tile / ntn) * BM + wr * HM, n0 = (tile % ntn) * BN + wc * WN;
        const int E = (m0 / p.rows_per_batch + 1) * p.rows_per_batch - m0;      // wave rows >= E belong to the next image (E >= HM: none)
        // the row bias slot (first / next image) is a per-lane LDS address; bias + row bias are re-read per fragment row rather than
        // held in registers across the rows (the accumulators, the fragments just read and the loader state leave ~25 free registers)
        // residual: all loads of the half tile first (their registers are the half's own, dead, accumulators' worth), ONE wait, then
        // the sums -- hipcc waits vmcnt(0) at the first use of a register-destination load while LDS-DMA loads are in flight, so a
        // load-use pair per fragment row would drain the K-tile stream once per row
        if constexpr (RES && X3) {
            // fp32 residual (16 bytes per lane and fragment): loaded STRAIGHT INTO the half tile's own, dead, accumulators -- no staging registers at
            // all -- with one wait for the whole half, then bias + row bias are added in place.  (Round 4: the staged form, one fragment row per
            // batch, still made hipcc spill 24-138 registers per lane in the RES x X3 instantiations and drained the K-tile stream four times per half.)
            const int voff = (l15 * p.ldr + 4 * g) * 4;
#pragma unroll
            for (int i = i0; i < i0 + FH; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcR, voff, ((m0 + i * 16) * p.ldr + n0 + j * 16) * 4, 0));
            // column by column: (bias + row bias of the first image) and (bias + row bias of the next image) of the lane's four columns, selected per
            // fragment row -- 8 live registers instead of two LDS reads per fragment (round 5: the split-bf16 core holds up to 72 fragment registers)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(slot + (j * 16 + 4 * g) * 4);
                const f32x4 c0 = bv + *reinterpret_cast<const f32x4*>(slot + 512 + (j * 16 + 4 * g) * 4);
                const f32x4 c1 = bv + *reinterpret_cast<const f32x4*>(slot + 1024 + (j * 16 + 4 * g) * 4);
#pragma unroll
                for (int i = i0; i < i0 + FH; ++i) acc[i][j] += (i * 16 + l15 >= E) ? c1 : c0;
                __builtin_amdgcn_sched_barrier(0);
            }
            return;
        }
        constexpr int RB = X3 ? 1 : ((BM == 256 && BN == 320) ? 2 : FH);      // fragment rows per batch (register budget of the largest tile)
#pragma unroll
        for (int ib = i0; ib < i0 + FH; ib += RB) {
            u32x2 w[(RES && !X3) ? RB : 1][(RES && !X3) ? FN : 1];
            u32x4 w4[(RES && X3) ? RB : 1][(RES && X3) ? FN : 1];
            if constexpr (RES && X3) {               // fp32 residual: 16 bytes per lane
                const int voff = (l15 * p.ldr + 4 * g) * 4;
#pragma unroll
                for (int i = 0; i < RB; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        w4[i][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcR, voff, ((m0 + (ib + i) * 16) * p.ldr + n0 + j * 16) * 4, 0));
            } else if constexpr (RES) {
                const int voff = (l15 * p.ldr + 4 * g) * 2;
#pragma unroll
                for (int i = 0; i < RB; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        w[i][j] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrcR, voff, ((m0 + (ib + i) * 16) * p.ldr + n0 + j * 16) * 2, 0));
            }
#pragma unroll
            for (int i = ib; i < ib + RB && i < i0 + FH; ++i) {
                const int rb_off = (i * 16 + l15 >= E) ? 1024 : 512;
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    f32x4 c = *reinterpret_cast<const f32x4*>(slot + (j * 16 + 4 * g) * 4) + *reinterpret_cast<const f32x4*>(slot + rb_off + (j * 16 + 4 * g) * 4);
                    if constexpr (F8) c *= f8_inv;
                    if constexpr (RES && X3) {
                        acc[i][j] = c + __builtin_bit_cast(f32x4, w4[i - ib][j]);
                    } else if constexpr (RES) {
                        const unsigned w0 = w[i - ib][j][0], w1 = w[i - ib][j][1];
                        if constexpr (F8) {
                            acc[i][j][0] = fmaf(__uint_as_float(w0 << 16), f8_inv, c[0]);
                            acc[i][j][1] = fmaf(__uint_as_float(w0 & 0xffff0000u), f8_inv, c[1]);
                            acc[i][j][2] = fmaf(__uint_as_float(w1 << 16), f8_inv, c[2]);
                            acc[i][j][3] = fmaf(__uint_as_float(w1 & 0xffff0000u), f8_inv, c[3]);
                        } else {
                        acc[i][j][0] = c[0] + __uint_as_float(w0 << 16);
                        acc[i][j][1] = c[1] + __uint_as_float(w0 & 0xffff0000u);
                        acc[i][j][2] = c[2] + __uint_as_float(w1 << 16);
                        acc[i][j][3] = c[3] + __uint_as_float(w1 & 0xffff0000u);
                        }
                    } else {
                        acc[i][j] = c;
                    }
                }
            }
        }
    };
    // epilogue: convert and store (rows past M fall outside the descriptor and are dropped by the range check)
    const __amdgpu_buffer_rsrc_t rsrcWs = __builtin_amdgcn_make_buffer_rsrc(p.ws, 0, SPLIT ? (int)((long)nsl * p.M * p.N * 4) : 0, 0x00020000);
    auto store_rows = [&](int wtile, auto I0) {
        constexpr int i0 = decltype(I0)::value;
        const int tile = wtile / nsl;
        const int m0 = (tile / ntn) * BM + wr * HM, n0 = (tile % ntn) * BN + wc * WN;
        // Lane order of the stores.  In the MFMA result a lane holds 4 consecutive columns of row l15, so the 64 lanes of a store
        // instruction touch 16 rows and NO two neighbouring lanes are neighbours in memory: the address unit takes such an
        // instruction one lane at a time (measured, tools/native/store_bw.hip: 7 B/clk/CU, and it serves the CU's loads in the same
        // queue, so the K-tile stream stands still meanwhile: 20k cycles per 256 x 320 tile).  Every packed register is therefore
        // permuted across the wave first (ds_bpermute: LDS crossbar, no LDS memory) so that lane 4 r + c holds columns 4c..4c+3 of
        // row r: runs of four lanes write 32 contiguous bytes, 24 B/clk/CU.
        int ln = lane;
        asm volatile("" : "+v"(ln));                  // lane constants of the epilogue are recomputed here, not carried through the K loop (registers)
        const int pr = ln >> 2, pg = ln & 3;          // row and column group this lane stores
        const int paddr = (pr + 16 * pg) * 4;         // ds_bpermute address: the lane that computed them
        auto perm = [&](unsigned v) { return (unsigned)__builtin_amdgcn_ds_bpermute(paddr, (int)v); };
        if constexpr (SPLIT) {                        // raw fp32 accumulators to slab (wtile % nsl): rows past M are dropped by the range check
            const int slice = wtile - tile * nsl;
            const int voff = (pr * p.N + 4 * pg) * 4;
#pragma unroll
            for (int i = i0; i < i0 + FH; ++i) {
                const int vo = m0 + i * 16 + pr < p.M ? voff : OOB;           // a row past M would land in the next slab: out of range instead
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    u32x4 w = __builtin_bit_cast(u32x4, acc[i][j]);
                    w[0] = perm(w[0]); w[1] = perm(w[1]); w[2] = perm(w[2]); w[3] = perm(w[3]);
                    __builtin_amdgcn_raw_buffer_store_b128(w, rsrcWs, vo, ((slice * p.M + m0 + i * 16) * p.N + n0 + j * 16) * 4, 0);
                }
            }
            return;
        }
        if constexpr (TRANS) {                        // out[(b * N + n) * ldo + s], four consecutive s per lane (rows_per_batch % 16 == 0)
            const int voff = (pr * p.ldo + 4 * pg) * OSZ;
#pragma unroll
            for (int i = i0; i < i0 + FH; ++i) {
                const int mb = m0 + i * 16;
                const int bb = mb / p.rows_per_batch, sb = mb - bb * p.rows_per_batch;
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    if constexpr (X3) {               // fp32 V^T (the split-bf16 attention kernels read fp32 operands): 16 bytes per lane
                        u32x4 w4 = __builtin_bit_cast(u32x4, acc[i][j]);
                        w4[0] = perm(w4[0]); w4[1] = perm(w4[1]); w4[2] = perm(w4[2]); w4[3] = perm(w4[3]);
                        if (p.flags & FFN_IG_OUT_KV64) {      // the attention kernels' pre-split V^T image: positions s .. s + 3 of key tile s / 64 -> hi 8 bytes, lo 8 bytes
                            const float f0 = __uint_as_float(w4[0]), f1 = __uint_as_float(w4[1]), f2 = __uint_as_float(w4[2]), f3 = __uint_as_float(w4[3]);
                            u32x2 hi, lo;
                            hi[0] = pack_bf16x2(f0, f1);
                            hi[1] = pack_bf16x2(f2, f3);
                            lo[0] = pack_bf16x2(f0 - __uint_as_float(hi[0] << 16), f1 - __uint_as_float(hi[0] & 0xffff0000u));
                            lo[1] = pack_bf16x2(f2 - __uint_as_float(hi[1] << 16), f3 - __uint_as_float(hi[1] & 0xffff0000u));
                            const int vo2 = mb + 4 * pg < p.M ? (pr * p.ldo * 4 + 4 * pg * 2) : OOB;
                            const int so2 = (bb * p.N + n0 + j * 16) * p.ldo * 4 + (sb >> 6) * 256 + (sb & 63) * 2;
                            __builtin_amdgcn_raw_buffer_store_b64(hi, rsrcO, vo2, so2, 0);
                            __builtin_amdgcn_raw_buffer_store_b64(lo, rsrcO, vo2, so2 + 128, 0);
                            continue;
                        }
                        __builtin_amdgcn_raw_buffer_store_b128(w4, rsrcO, mb + 4 * pg < p.M ? voff : OOB, ((bb * p.N + n0 + j * 16) * p.ldo + sb) * 4, 0);
                        continue;
                    }
                    u32x2 w;
                    w[0] = perm(pack_bf16x2(acc[i][j][0], acc[i][j][1]));
                    w[1] = perm(pack_bf16x2(acc[i][j][2], acc[i][j][3]));
                    __builtin_amdgcn_raw_buffer_store_b64(w, rsrcO, mb + 4 * pg < p.M ? voff : OOB, ((bb * p.N + n0 + j * 16) * p.ldo + sb) * 2, 0);
                }
            }
            return;
        }
        if (PP_ABL == 4) {
#pragma unroll
            for (int i = i0; i < i0 + FH; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) asm volatile("" ::"v"(acc[i][j]));
            return;
        }
        if constexpr (X3) {                           // fp32 row-major output: lane (pr, pg) stores columns 4 pg .. 4 pg + 3 of row pr, 16 bytes
            const int voff = (pr * p.ldo + 4 * pg) * 4;
#pragma unroll
            for (int i = i0; i < i0 + FH; ++i) {
                const int vo = m0 + i * 16 + pr < p.M ? voff : OOB;
                if constexpr (GEGLU) {                // blocked pair form: out is bf16 [M][ldo]; the wave's 32 output columns n0/2 .. n0/2 + 31 are ONE
                                                      // 128-byte block [hi(32) | lo(32)] at element (n0/2 / 32) * 64 (ldo / 2 = output columns, % 32 == 0)
                    const int vo2 = m0 + i * 16 + pr < p.M ? (pr * p.ldo + 4 * pg) * 2 : OOB;
#pragma unroll
                    for (int j = 0; j + 1 < FN; j += 2) {
                        const f32x2_t g01 = (PP_ABL == 3 ? f32x2_t{acc[i][j + 1][0], acc[i][j + 1][1]} : gelu_erf26_2(f32x2_t{acc[i][j + 1][0], acc[i][j + 1][1]})) * f32x2_t{acc[i][j][0], acc[i][j][1]};
                        const f32x2_t g23 = (PP_ABL == 3 ? f32x2_t{acc[i][j + 1][2], acc[i][j + 1][3]} : gelu_erf26_2(f32x2_t{acc[i][j + 1][2], acc[i][j + 1][3]})) * f32x2_t{acc[i][j][2], acc[i][j][3]};
                        const float v[4] = {g01[0], g01[1], g23[0], g23[1]};
                        u32x2 hi, lo;
                        hi[0] = pack_bf16x2(v[0], v[1]);
                        hi[1] = pack_bf16x2(v[2], v[3]);
                        lo[0] = pack_bf16x2(v[0] - __uint_as_float(hi[0] << 16), v[1] - __uint_as_float(hi[0] & 0xffff0000u));
                        lo[1] = pack_bf16x2(v[2] - __uint_as_float(hi[1] << 16), v[3] - __uint_as_float(hi[1] & 0xffff0000u));
                        hi[0] = perm(hi[0]); hi[1] = perm(hi[1]); lo[0] = perm(lo[0]); lo[1] = perm(lo[1]);
                        const int so = ((m0 + i * 16) * p.ldo + n0 + (j / 2) * 16) * 2;      // block base = 2 * (n0 / 2) elements (n0 / 2 % 32 == 0)
                        __builtin_amdgcn_raw_buffer_store_b64(hi, rsrcO, vo2, so, 0);
                        __builtin_amdgcn_raw_buffer_store_b64(lo, rsrcO, vo2, so + 64, 0);         // lo: 32 elements behind hi
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < FN; ++j) {
                        u32x4 w = __builtin_bit_cast(u32x4, acc[i][j]);
                        w[0] = perm(w[0]); w[1] = perm(w[1]); w[2] = perm(w[2]); w[3] = perm(w[3]);
                        const int nf = n0 + j * 16;           // first column of the fragment (a fragment never straddles a 32- or 64-column block)
                        if (p.flags & FFN_IG_OUT_PAIR) {      // blocked pair rows for the next split-bf16 GEMM (bias / residual already in the accumulators): out is
                                                              // bf16 [M][ldo], columns 32 b .. 32 b + 31 = ONE 128-byte block [hi(32) | lo(32)] at element 64 b
                            const float f0 = __uint_as_float(w[0]), f1 = __uint_as_float(w[1]), f2 = __uint_as_float(w[2]), f3 = __uint_as_float(w[3]);
                            u32x2 hi, lo;
                            hi[0] = pack_bf16x2(f0, f1);
                            hi[1] = pack_bf16x2(f2, f3);
                            lo[0] = pack_bf16x2(f0 - __uint_as_float(hi[0] << 16), f1 - __uint_as_float(hi[0] & 0xffff0000u));
                            lo[1] = pack_bf16x2(f2 - __uint_as_float(hi[1] << 16), f3 - __uint_as_float(hi[1] & 0xffff0000u));
                            const int vo2 = m0 + i * 16 + pr < p.M ? (pr * p.ldo + 4 * pg) * 2 : OOB;
                            const int so2 = (m0 + i * 16) * p.ldo * 2 + (nf >> 5) * 128 + (nf & 31) * 2;
                            __builtin_amdgcn_raw_buffer_store_b64(hi, rsrcO, vo2, so2, 0);
                            __builtin_amdgcn_raw_buffer_store_b64(lo, rsrcO, vo2, so2 + 64, 0);
                            continue;
                        }
                        if ((p.flags & FFN_IG_OUT_KV64) && nf >= p.kv64_from) {      // the attention kernels' pre-split K image: [hi(64) | lo(64)] per head
                            const float f0 = __uint_as_float(w[0]), f1 = __uint_as_float(w[1]), f2 = __uint_as_float(w[2]), f3 = __uint_as_float(w[3]);
                            u32x2 hi, lo;
                            hi[0] = pack_bf16x2(f0, f1);
                            hi[1] = pack_bf16x2(f2, f3);
                            lo[0] = pack_bf16x2(f0 - __uint_as_float(hi[0] << 16), f1 - __uint_as_float(hi[0] & 0xffff0000u));
                            lo[1] = pack_bf16x2(f2 - __uint_as_float(hi[1] << 16), f3 - __uint_as_float(hi[1] & 0xffff0000u));
                            const int crel = nf - p.kv64_from;
                            const int vo2 = m0 + i * 16 + pr < p.M ? (pr * p.ldo * 4 + 4 * pg * 2) : OOB;
                            const int so2 = ((m0 + i * 16) * p.ldo + p.kv64_from) * 4 + (crel >> 6) * 256 + (crel & 63) * 2;
                            __builtin_amdgcn_raw_buffer_store_b64(hi, rsrcO, vo2, so2, 0);
                            __builtin_amdgcn_raw_buffer_store_b64(lo, rsrcO, vo2, so2 + 128, 0);
                            continue;
                        }
                        __builtin_amdgcn_raw_buffer_store_b128(w, rsrcO, vo, ((m0 + i * 16) * p.ldo + n0 + j * 16) * 4, 0);
                    }
                }
            }
            return;
        }
        // bf16 row-major output.  Output fragments (16 columns = 32 B per row) are taken in PAIRS: v_permlane16_swap exchanges
        // the odd 16-lane rows of one fragment's register with the even rows of its neighbour's, after which lane (row, g) holds
        // 16 contiguous bytes (g = 0: columns 0-7 of the first fragment, 1: 0-7 of the second, 2: 8-15 of the first, 3: 8-15 of the
        // second); the wave permutation then puts the four 16-byte pieces of a row into neighbouring lanes (4 r + c, c = memory
        // order): one 16-byte store per lane, 64 contiguous bytes per row (42 B/clk/CU).  An odd fragment left over goes out as
        // 8 bytes per lane (runs of four lanes = 32 B).
        if constexpr (GEGLU) {                        // half as many stores beside twice the arithmetic: the permutation does not pay here (measured)
            const int voff = (l15 * p.ldo + 4 * g) * 2;
#pragma unroll
            for (int i = i0; i < i0 + FH; ++i)
#pragma unroll
                for (int j = 0; j + 1 < FN; j += 2) {
                    const f32x2_t g01 = (PP_ABL == 3 ? f32x2_t{acc[i][j + 1][0], acc[i][j + 1][1]} : gelu_erf_fast2(f32x2_t{acc[i][j + 1][0], acc[i][j + 1][1]})) * f32x2_t{acc[i][j][0], acc[i][j][1]};
                    const f32x2_t g23 = (PP_ABL == 3 ? f32x2_t{acc[i][j + 1][2], acc[i][j + 1][3]} : gelu_erf_fast2(f32x2_t{acc[i][j + 1][2], acc[i][j + 1][3]})) * f32x2_t{acc[i][j][2], acc[i][j][3]};
                    u32x2 w;
                    w[0] = pack_bf16x2(g01[0], g01[1]);
                    w[1] = pack_bf16x2(g23[0], g23[1]);
                    __builtin_amdgcn_raw_buffer_store_b64(w, rsrcO, voff, ((m0 + i * 16) * p.ldo + n0 / 2 + (j / 2) * 16) * 2, 0);
                }
            return;
        }
        if constexpr (F8) {                           // un-scale the half tile IN PLACE first (the accumulators restart right after the stores);
                                                      // with the multiply fused into the pack, rows l15 % 4 == 3 of a row's first fragment pair
                                                      // came out as garbage on gfx950 (the scaled value went pk_mul -> cvt -> v_permlane16_swap
                                                      // back to back): the conversion below now reads plain registers like the bf16 path
#pragma unroll
            for (int i = i0; i < i0 + FH; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    acc[i][j] *= p.alpha;
                    asm volatile("" : "+v"(acc[i][j]));
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (!RES && !F8) {                  // plain epilogue activations (the accumulators started at the bias): the DINOv2 MLP's GELU, the DPT head's ReLU
            if (p.flags & (FFN_IG_OUT_GELU | FFN_IG_OUT_RELU)) {
                const bool gelu = p.flags & FFN_IG_OUT_GELU;
#pragma unroll
                for (int i = i0; i < i0 + FH; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[i][j][r] = gelu ? gelu_erf_fast(acc[i][j][r]) : fmaxf(acc[i][j][r], 0.f);
                        asm volatile("" : "+v"(acc[i][j]));      // (plain registers in front of the pack / lane permutation, like the fp8 un-scale)
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        constexpr int NOF = FN;                       // output fragments per fragment row
        const int voff8 = (pr * p.ldo + 4 * pg) * 2;
        const int voff16 = pr * p.ldo * 2 + pg * 16;
        const int paddr16 = (pr + 16 * (((pg & 1) << 1) | (pg >> 1))) * 4;
        const int ncol0 = n0;
#pragma unroll
        for (int i = i0; i < i0 + FH; ++i) {
            auto outfrag = [&](int jo, unsigned& w0, unsigned& w1) {
                w0 = pack_bf16x2(acc[i][jo][0], acc[i][jo][1]);
                w1 = pack_bf16x2(acc[i][jo][2], acc[i][jo][3]);
            };
            const int row_off = PP_ABL == 5 ? ((m0 + i * 16) & 255) * p.ldo + ncol0 % 320 : (m0 + i * 16) * p.ldo + ncol0;      // 5: every tile to the same 256 x 320 window
#pragma unroll
            for (int jo = 0; jo + 1 < NOF; jo += 2) {
                unsigned a0, a1, b0, b1;
                outfrag(jo, a0, a1);
                outfrag(jo + 1, b0, b1);
                const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
                u32x4 v;
                v[0] = (unsigned)__builtin_amdgcn_ds_bpermute(paddr16, (int)s0[0]);
                v[1] = (unsigned)__builtin_amdgcn_ds_bpermute(paddr16, (int)s1[0]);
                v[2] = (unsigned)__builtin_amdgcn_ds_bpermute(paddr16, (int)s0[1]);
                v[3] = (unsigned)__builtin_amdgcn_ds_bpermute(paddr16, (int)s1[1]);
                if (PP_ABL == 8) asm volatile("" ::"v"(v));
                else __builtin_amdgcn_raw_buffer_store_b128(v, rsrcO, voff16, (row_off + jo * 16) * 2, 0);
            }
            if constexpr (NOF % 2 == 1) {
                unsigned a0, a1;
                outfrag(NOF - 1, a0, a1);
                u32x2 v;
                v[0] = perm(a0);
                v[1] = perm(a1);
                if (PP_ABL == 8) asm volatile("" ::"v"(v));
                else __builtin_amdgcn_raw_buffer_store_b64(v, rsrcO, voff8, (row_off + (NOF - 1) * 16) * 2, 0);
            }
        }
    };

    // ---- prologue: K tile 0 of the stream and the first tile's column vectors, fully landed -----------------------------------
    prep(l_tile);
    issue_colvec(l_tile);
#pragma unroll
    for (int i = 0; i < FN; ++i) issue_b(i, 0);
    k_position();
#pragma unroll
    for (int i = 0; i < NA; ++i) issue_a(i, 0);
    advance();
    pp_wait_vmcnt<0>();
    if constexpr (X3) {                               // split-bf16 core: the early W pieces of stage 1 (see the loop), in flight across the first barrier
        constexpr int NBE0 = X3_NBE(BN / 64, BM / 64);
        if (S > 1) {
#pragma unroll
            for (int i = 0; i < NBE0; ++i) issue_b(i, 1);
        }
    }
    init_rows(first, I0_t{});
    init_rows(first, I4_t{});
    pp_barrier();
    if (wr == 1) pp_barrier();                        // the lagging group starts one barrier late

    int c_tile = first, c_kt = 0, buf = 0;
    int p_tile = -1;                                  // tile whose high rows still sit in the accumulators (-1: none)
    if constexpr (X3) {
        // ---- split-bf16 core v2: two phases per stage (32 real K elements), three products per fragment pair from one set of fragment reads ----
        // KEEPA: a_hi and a_lo of a phase's rows both sit in registers.  The 256 x 320 tile (160 accumulator registers) holds ONE A set and
        // fetches a_lo over a_hi's registers behind the second product (A rows are re-staged two phases after their phase: safe for a read inside
        // the MFMA section; W rows are re-staged in the very next phase L, so W fragments are only ever read in phase L's load section).
        constexpr bool KEEPA = !(BM == 256 && BN == 320);
        constexpr int NAL = 2, NAH = NA - 2;                   // A pieces requested in phase L (every low piece is among them) / in phase H
        // W pieces of stage s + 2 requested EARLY, in phase H of stage s (right behind the loader's step to s + 2, into the CURRENT buffer, whose W
        // rows were last read in this stage's phase L: retired by the lgkmcnt(0) in front of that phase's barrier), the rest in phase L of stage
        // s + 1: the LDS-DMA issue, the long pole of a load section (100-185 cycles per piece beside the partner's MFMAs), is then spread evenly
        // over the two phases (192 x 320: 4 + 4 pieces instead of 7 + 1) instead of making phase L's load section twice its MFMA section
        constexpr int NBE = X3_NBE(FN, NA);
        constexpr int NBL = FN - NBE;
        (void)NBL;
        static_assert(NBE >= 0 && NBE <= FN, "early W pieces");
        u32x4 fal[KEEPA ? FH : 1], fbl[FN];                    // lo fragments (fa / fb hold the hi ones); w_hi / w_lo stay for both phases of a stage
        // one product: C[m = l15][n = 4g + r] (row-major epilogues) or, TRANS, C[m = 4g + r][n = l15] (operands swap roles)
        auto mm = [&](const u32x4& w_, const u32x4& a_, f32x4& c_) {
            if constexpr (TRANS) DT<T>::mma(a_, w_, c_);
            else DT<T>::mma(w_, a_, c_);
        };
        auto mfma3 = [&](auto I0, int b_) {
            constexpr int i0 = decltype(I0)::value;
            if (PP_PRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < FH; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) mm(fb[j], fa[i], acc[i0 + i][j]);                  // A_hi x W_hi
            if constexpr (KEEPA) {
#pragma unroll
                for (int i = 0; i < FH; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) mm(fb[j], fal[i], acc[i0 + i][j]);             // A_lo x W_hi
#pragma unroll
                for (int i = 0; i < FH; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) mm(fbl[j], fa[i], acc[i0 + i][j]);             // A_hi x W_lo
            } else {
#pragma unroll
                for (int i = 0; i < FH; ++i) {
#pragma unroll
                    for (int j = 0; j < FN; ++j) mm(fbl[j], fa[i], acc[i0 + i][j]);             // A_hi x W_lo (row i), then a_lo[i] over a_hi[i]
                    fa[i] = *reinterpret_cast<const u32x4*>(smem + b_ * BUF + a_rd1 + (i0 + i) * 2048);
                }
#pragma unroll
                for (int i = 0; i < FH; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) mm(fb[j], fa[i], acc[i0 + i][j]);              // A_lo x W_hi
            }
            if (PP_PRIO == 1) __builtin_amdgcn_s_setprio(0);
        };
        auto read_a_set = [&](int b_, int i0) {                // a_hi (, a_lo) of fragment rows i0 .. i0 + FH - 1
            read_a(b_, a_rd0, i0);
            if constexpr (KEEPA) {
#pragma unroll
                for (int i = 0; i < FH; ++i) fal[i] = *reinterpret_cast<const u32x4*>(smem + b_ * BUF + a_rd1 + (i0 + i) * 2048);
            }
        };
        auto read_low_set = [&](int b_) {                      // phase L's fragments: the low rows' A set, w_hi and w_lo
            read_a_set(b_, 0);
            read_b(b_, b_rd0);
#pragma unroll
            for (int j = 0; j < FN; ++j) fbl[j] = *reinterpret_cast<const u32x4*>(smem + b_ * BUF + b_rd1 + j * 2048);
        };
        for (int s = 0; s < S; ++s) {
            const bool more = s + 1 < S;              // another stage follows in this workgroup's stream: request it during this one
            const int nb = buf ^ 1;

            // ---- phase L: low rows ----
            __builtin_amdgcn_sched_barrier(0);
            PP_STAMP_AT(0);
            if (p_tile >= 0) {
                // first stage of a new output tile (never the last stage of the stream: nk >= 2).  The previous tile's high rows, final since
                // the phase-H MFMAs, leave now; its low rows left in that phase H.  VM queue, oldest first:
                //   A-high of this stage | this tile's column vectors | low stores | high stores | W x FN | A-low x NAL     (see the bf16 loop below)
                store_rows(p_tile, I4_t{});
#pragma unroll
                for (int i = NBE; i < FN; ++i) issue_b(i, nb);
                k_position();
                issue_a(0, nb);
                issue_a(1, nb);
                pp_wait_vmcnt<FN + NAL + 2 * NST>();  // through A-high and the column vectors (younger: early W, the stores, A-low, late W)
                init_rows(c_tile, I0_t{});
                __builtin_amdgcn_sched_barrier(0);    // the fragment reads last: their registers are free for the epilogue's temporaries
                read_low_set(buf);
            } else {
                read_low_set(buf);
                PP_STAMP_AT(8);
                if (more) {
#pragma unroll
                    for (int i = NBE; i < FN; ++i) issue_b(i, nb);
                    k_position();
                    issue_a(0, nb);
                    issue_a(1, nb);
                    PP_STAMP_AT(9);
                    pp_wait_vmcnt<FN + NAL>();        // A-high of THIS stage (requested in phase H of the previous one, in front of the early W pieces) has landed
                } else {
                    pp_wait_vmcnt<0>();
                }
                PP_STAMP_AT(10);
            }
            if (NBE > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this stage's W rows are re-staged from the next phase on: their reads retire HERE
            __builtin_amdgcn_sched_barrier(0);
            PP_STAMP_AT(1);
            pp_barrier();
            PP_STAMP_AT(2);
            mfma3(I0_t{}, buf);
            __builtin_amdgcn_sched_barrier(0);
            PP_STAMP_AT(3);
            pp_barrier();
            PP_STAMP_AT(4);

            // ---- phase H: high rows ----  (tile-boundary work BEFORE the fragment reads: the A set's registers are free for its temporaries.
            // Reading first in the stages without boundary work -- so that the fragments have landed when the MFMA section opens -- was measured:
            // +-1 % on the tiles that hold both A sets, and the 256 x 320 residual tiles spill inside the loop and lose 30-45 %: not kept)
            if (more) {
#pragma unroll
                for (int i = 2; i < NA; ++i) issue_a(i, nb);
                advance();                            // the loader now stands on stage s + 2
                const bool more2 = s + 2 < S;
                if (more2) {
#pragma unroll
                    for (int i = 0; i < NBE; ++i) issue_b(i, buf);
                }
                // W and A-low of the next stage have landed (this phase's A-high and early W pieces, and the next output tile's 6 column-vector loads, may fly)
                if (more2) {
                    if (switched) pp_wait_vmcnt<NAH + NBE + (SPLIT ? 0 : 6)>();
                    else pp_wait_vmcnt<NAH + NBE>();
                } else {
                    pp_wait_vmcnt<NAH>();             // (the loader never switches tiles behind the last stage)
                }
            }
            if (p_tile >= 0) {                        // the high rows of the new tile start
                init_rows(c_tile, I4_t{});
                p_tile = -1;
            }
            if (c_kt == nk - 1) store_rows(c_tile, I0_t{});      // last stage of the output tile: its low rows are final since phase L (they restart in the next phase L)
            __builtin_amdgcn_sched_barrier(0);
            read_a_set(buf, FH);
            __builtin_amdgcn_sched_barrier(0);
            PP_STAMP_AT(5);
            pp_barrier();
            PP_STAMP_AT(6);
            mfma3(I4_t{}, buf);
            __builtin_amdgcn_sched_barrier(0);
            PP_STAMP_AT(7);
            pp_barrier();

            buf = nb;
            if (++c_kt == nk) {                       // output tile complete (high rows are stored in the next stage's phase L, or below)
                p_tile = c_tile;
                c_kt = 0;
                c_tile += G;
            }
        }
    } else {
        for (int s = 0; s < S; ++s) {
            const bool more = s + 1 < S;                  // another K tile follows in this workgroup's stream: request it during this one
            const int nb = buf ^ 1;
            const int ra = buf, rb = buf, la = nb, lb = nb;      // LDS buffers this K tile reads / the next one is staged into
            const bool needA = more, needB = more;

            // ---- phase 1: low rows, k-substep 0 ----
            __builtin_amdgcn_sched_barrier(0);
            if (p_tile < 0) {
                read_a(ra, a_rd0, 0);
                read_b(rb, b_rd0);
            }
            if (p_tile >= 0) {
                // first K tile of a new output tile (never the last K tile of the stream: nk >= 2).  The previous tile's high rows, final
                // since the phase-4 MFMAs, leave now; its low rows left in that phase 4.  The VM counter is in order and counts stores:
                // a wait that only needs loads OLDER than the stores names the stores (and the residual loads behind them) as allowed
                // in flight, so no wait of the K loop ever sits behind a store's round trip.  Queue, oldest first:
                //   A-high of this K tile | next tile's column vectors | low stores | high stores | B x NB1
                // (the residual loads of init_rows are the compiler's: it drains the counter at their first use)
                store_rows(p_tile, I4_t{});
#pragma unroll
                for (int i = 0; i < NB1; ++i) issue_b(i, lb);
                pp_wait_vmcnt<NB1 + 2 * NST>();           // through A-high
                init_rows(c_tile, I0_t{});
                __builtin_amdgcn_sched_barrier(0);        // the fragment reads last: their registers are free for the epilogue's temporaries
                read_a(ra, a_rd0, 0);
                read_b(rb, b_rd0);
            } else if (needB) {
#pragma unroll
                for (int i = 0; i < NB1; ++i) issue_b(i, lb);
                pp_wait_vmcnt<NB1>();                     // A-high of THIS K tile (requested in phase 4 of the previous one) has landed
            } else {
                pp_wait_vmcnt<0>();                       // last K tile of the stream
            }
            __builtin_amdgcn_sched_barrier(0);
            pp_barrier();
            mfma_rows(I0_t{});
            __builtin_amdgcn_sched_barrier(0);
            pp_barrier();

            // ---- phase 2: high rows, k-substep 0 ----
            read_a(ra, a_rd0, FH);
            if (needB) {
#pragma unroll
                for (int i = NB1; i < FN; ++i) issue_b(i, lb);
            }
            if (p_tile >= 0) {                            // the high rows of the new tile start
                init_rows(c_tile, I4_t{});
                p_tile = -1;
            }
            __builtin_amdgcn_sched_barrier(0);
            pp_barrier();
            mfma_rows(I4_t{});
            __builtin_amdgcn_sched_barrier(0);
            pp_barrier();

            // ---- phase 3: low rows, k-substep 1 ----
            read_a(ra, a_rd1, 0);
            read_b(rb, b_rd1);
            if (needA) {
                k_position();
                issue_a(0, la);
                issue_a(1, la);
            }
            __builtin_amdgcn_sched_barrier(0);
            pp_barrier();
            mfma_rows(I0_t{});
            __builtin_amdgcn_sched_barrier(0);
            pp_barrier();

            // ---- phase 4: high rows, k-substep 1 ----
            read_a(ra, a_rd1, FH);
            if (more) {
                if (needA) {
#pragma unroll
                    for (int i = 2; i < NA; ++i) issue_a(i, la);
                }
                advance();
                // B and A-low of the next K tile have landed (this phase's A-high pieces, and the next output tile's 6 column-vector
                // loads, may still be in flight)
                if (PP_ABL == 9 && c_kt < 2) {}           // timing experiment: no wait in the two K tiles behind the stores (results garbage)
                else if (switched) pp_wait_vmcnt<NA - 2 + (SPLIT ? 0 : 6)>();
                else pp_wait_vmcnt<NA - 2>();
            }
            if (c_kt == nk - 1) {                         // last K tile of the output tile: its low rows are final since phase 3
#ifdef PP_TRACE
                if (tid == 0) pp_trace[blockIdx.x * 64 + 2 * ((c_tile - first) / G & 31)] = __builtin_amdgcn_s_memrealtime();
#endif
                store_rows(c_tile, I0_t{});               // (they restart in phase 1 of the next K tile)
#ifdef PP_TRACE
                if (tid == 0) pp_trace[blockIdx.x * 64 + 2 * ((c_tile - first) / G & 31) + 1] = __builtin_amdgcn_s_memrealtime();
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
            pp_barrier();
            mfma_rows(I4_t{});
            __builtin_amdgcn_sched_barrier(0);
            pp_barrier();

            buf = nb;
            if (++c_kt == nk) {                           // output tile complete (high rows are stored in the next K tile's phase 2, or below)
                p_tile = c_tile;
                c_kt = 0;
                c_tile += G;
            }
        }
    }
    if (p_tile >= 0) store_rows(p_tile, I4_t{});
    if (wr == 0) pp_barrier();                        // balance the lagging group's extra barrier
#ifdef PP_STAMP
    if (blockIdx.x == 7) pp_stamp_out[tid] = *reinterpret_cast<const unsigned long long*>(smem + CV + 12288 + tid * 8);
#endif
}
