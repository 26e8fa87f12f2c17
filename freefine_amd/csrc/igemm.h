// Implicit-GEMM kernel family for the SD UNet / VAE hot path on gfx950.
//
//   out[m, n] = epilogue( sum_k A(m, k) * W[n, k] )
//
// A(m, k) is either a dense row-major matrix (Linear layers, 1x1 convs: AMODE_DENSE) or the im2col
// view of an NHWC activation for a 3x3 convolution (AMODE_CONV3: k = (ky*3+kx)*Cin + ci, optional
// stride 2, optional fused nearest-2x upsample of the input, zero padding) -- the gather happens in
// the global->LDS loader, nothing is materialised.  W is always [N][Kpad] with K contiguous, i.e. the
// layout torch.nn.Linear stores and the layout conv weights are repacked to once at load time.
//
// Replaces (reference call sites): every diffusers ResnetBlock2D / Downsample2D / Upsample2D conv and every
// Linear reached from override_forward (/root/reference/src/utils/attention.py:105-214) and the to_q/to_k/
// to_v/to_out projections of the hooked Attention.forward (attention.py:372-407).
//
// Tile: BM x BN outputs per 256-thread workgroup (4 waves as 2x2), 128 bytes of K per stage (64 bf16 / 32
// f32), two LDS stages, register-staged global loads issued one stage ahead (issue-early / write-late).
// LDS rows are 128 B with a 16-byte-chunk XOR swizzle (chunk ^= row & 7): conflict-free for both the
// ds_write_b128 of the loader and the ds_read_b128 fragment reads.
#pragma once
#include <type_traits>
#include "common.h"
#include "../../include/freefine_hip.h"

enum { AMODE_DENSE = 0, AMODE_CONV3 = 1 };
enum { IG_OUT_SILU = FFN_IG_OUT_SILU, IG_OUT_F32 = FFN_IG_OUT_F32, IG_GEGLU = FFN_IG_GEGLU, IG_OUT_TRANSPOSED = FFN_IG_OUT_TRANSPOSED, IG_OUT_PAIR = FFN_IG_OUT_PAIR,
       IG_OUT_GELU = FFN_IG_OUT_GELU, IG_OUT_RELU = FFN_IG_OUT_RELU, IG_OUT_KV64 = FFN_IG_OUT_KV64 };
// the plain epilogue's activation (flags are launch-uniform)
__device__ __forceinline__ float ig_activation(float x, int flags) {
    if (flags & IG_OUT_SILU) return silu_exact(x);
    if (flags & IG_OUT_GELU) return gelu_erf(x);
    if (flags & IG_OUT_RELU) return fmaxf(x, 0.f);
    return x;
}
// four fp32 values, columns c .. c + 3 (c % 4 == 0) of a row of C -> the bf16 pair form of that row (common.h pair_pos), 8 bytes each
__device__ __forceinline__ void store_pair_row4(bf16* row, int c, int C, const float* v) {
    bf16* p = row + pair_pos(c, C);
    const int lo_off = pair_lo(C);
    u32x2 hi, lo;
    hi[0] = pack_bf16x2(v[0], v[1]);
    hi[1] = pack_bf16x2(v[2], v[3]);
    lo[0] = pack_bf16x2(v[0] - __uint_as_float(hi[0] << 16), v[1] - __uint_as_float(hi[0] & 0xffff0000u));
    lo[1] = pack_bf16x2(v[2] - __uint_as_float(hi[1] << 16), v[3] - __uint_as_float(hi[1] & 0xffff0000u));
    *reinterpret_cast<u32x2*>(p) = hi;
    *reinterpret_cast<u32x2*>(p + lo_off) = lo;
}
// FFN_IG_OUT_KV64 (include/freefine_hip.h): four values of 64-column block positions d .. d + 3 (d % 4 == 0) -> [hi(64) | lo(64)] bf16 at `blk` (the
// 256 bytes the block's fp32 values would occupy): the pre-split K / V^T image of the split-bf16 attention kernels, written by the projection itself
__device__ __forceinline__ void store_kv64_4(float* blk, int d, const float* v) {
    bf16* p = reinterpret_cast<bf16*>(blk) + d;
    u32x2 hi, lo;
    hi[0] = pack_bf16x2(v[0], v[1]);
    hi[1] = pack_bf16x2(v[2], v[3]);
    lo[0] = pack_bf16x2(v[0] - __uint_as_float(hi[0] << 16), v[1] - __uint_as_float(hi[0] & 0xffff0000u));
    lo[1] = pack_bf16x2(v[2] - __uint_as_float(hi[1] << 16), v[3] - __uint_as_float(hi[1] & 0xffff0000u));
    *reinterpret_cast<u32x2*>(p) = hi;
    *reinterpret_cast<u32x2*>(p + 64) = lo;
}
typedef ffn_igemm_desc IgemmParams;

// epilogue shared by the igemm kernels: acc[i][j] is the 16x16 fragment (i, j) of this wave's (BM/2) x (BN/2) sub-tile
// (mbase, nbase) = global row / column of the wave's sub-tile
template <typename T, int FM, int FN, bool SWAP>
__device__ __forceinline__ void igemm_epilogue(const IgemmParams& p, f32x4 (&acc)[FM][FN], int mbase, int nbase, int l15, int g) {
    // ---- epilogue --------------------------------------------------------------------------------
    T* __restrict__ outT = reinterpret_cast<T*>(p.out);
    float* __restrict__ outF = reinterpret_cast<float*>(p.out);
    const T* __restrict__ res = reinterpret_cast<const T*>(p.residual);
    const bool out_f32 = p.flags & IG_OUT_F32;
    const bool out_act = p.flags & (IG_OUT_SILU | IG_OUT_GELU | IG_OUT_RELU);

    if (SWAP && gridDim.y > 1) {
        // split-K partial: raw fp32 accumulators to slab blockIdx.y of the workspace
        float* __restrict__ slab = reinterpret_cast<float*>(p.ws) + (long)blockIdx.y * p.M * p.N;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int m = mbase + i * 16 + l15;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int n = nbase + j * 16 + 4 * g;
                if (m < p.M && n < p.N) {
                    float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    store4(slab + (long)m * p.N + n, v);
                }
            }
        }
        return;
    }
    if (SWAP) {
        // lane holds C[m = ..+l15][n = ..+4g+r], r = 0..3: four consecutive columns of one row
        if ((FN % 2 == 0) && (p.flags & IG_GEGLU)) {      // hidden / gate column blocks come in pairs: even FN only
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int m = mbase + i * 16 + l15;
#pragma unroll
                for (int j = 0; j + 1 < FN; j += 2) {
                    const int nh = nbase + j * 16 + 4 * g;  // packed column of the hidden half
                    if (m < p.M && nh < p.N) {
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float h = acc[i][j][r] * p.alpha, gt = acc[i][j + 1][r] * p.alpha;
                            if (p.bias) {
                                h += p.bias[nh + r];
                                gt += p.bias[nh + 16 + r];
                            }
                            v[r] = h * gelu_for<T>(gt);
                        }
                        const int no = (nbase) / 2 + (j / 2) * 16 + 4 * g;
                        if (p.flags & IG_OUT_PAIR) store_pair_row4(reinterpret_cast<bf16*>(p.out) + (long)m * p.ldo, no, p.ldo / 2, v);
                        else store4(outT + (long)m * p.ldo + no, v);
                    }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int m = mbase + i * 16 + l15;
                const int bb = m / p.rows_per_batch;
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int n = nbase + j * 16 + 4 * g;
                    if (m < p.M && n < p.N) {
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float x = acc[i][j][r] * p.alpha;
                            if (p.bias) x += p.bias[n + r];
                            if (p.rowbias) x += p.rowbias[(long)bb * p.ldrb + n + r];
                            if (out_act) x = ig_activation(x, p.flags);
                            v[r] = x;
                        }
                        if (res) {
                            float rr[4];
                            load4(res + (long)m * p.ldr + n, rr);
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] += rr[r];
                        }
                        if (p.flags & IG_OUT_PAIR)
                            store_pair_row4(reinterpret_cast<bf16*>(p.out) + (long)m * p.ldo, n, p.ldo / 2, v);
                        else if ((p.flags & IG_OUT_KV64) && n >= p.kv64_from)
                            store_kv64_4(outF + (long)m * p.ldo + p.kv64_from + ((n - p.kv64_from) & ~63), (n - p.kv64_from) & 63, v);
                        else if (out_f32)
                            store4(outF + (long)m * p.ldo + n, v);
                        else
                            store4(outT + (long)m * p.ldo + n, v);
                    }
                }
            }
        }
    } else {
        // lane holds C[m = ..+4g+r][n = ..+l15]: four consecutive rows of one column -> transposed store
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int mb = mbase + i * 16 + 4 * g;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int n = nbase + j * 16 + l15;
                if (n >= p.N) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = acc[i][j][r] * p.alpha;
                    if (p.bias) x += p.bias[n];
                    v[r] = x;
                }
                if (p.flags & IG_OUT_TRANSPOSED) {
                    const int bb = mb / p.rows_per_batch, s = mb - bb * p.rows_per_batch;
                    if ((p.flags & IG_OUT_KV64) && mb < p.M) {      // (fp32-typed launches only; rows_per_batch % 64 == 0, so M % 4 == 0)
                        store_kv64_4(outF + ((long)bb * p.N + n) * p.ldo + (s & ~63), s & 63, v);
                    } else if ((p.rows_per_batch & 3) == 0 && mb + 3 < p.M) {
                        store4(outT + ((long)bb * p.N + n) * p.ldo + s, v);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int m = mb + r;
                            if (m < p.M) {
                                const int b2 = m / p.rows_per_batch, s2 = m - b2 * p.rows_per_batch;
                                DT<T>::st(outT + ((long)b2 * p.N + n) * p.ldo + s2, v[r]);
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = mb + r;
                        if (m < p.M) DT<T>::st(outT + (long)m * p.ldo + n, v[r]);
                    }
                }
            }
        }
    }
}

template <typename T, int BM, int BN, int AMODE, bool SWAP>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmParams p) {
    constexpr int EPC = DT<T>::EPC;
    constexpr int BKE = 8 * EPC;  // K elements per stage (128 bytes)
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int FM = WM / 16, FN = WN / 16;
    constexpr int NA = BM / 32, NB = BN / 32;  // 16-byte chunks per thread per stage

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                    // [2][BM][128]
    char* Bs = smem + 2 * BM * 128;     // [2][BN][128]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, g = lane >> 4;

    const int ntn = (p.N + BN - 1) / BN;
    const int ntm = (p.M + BM - 1) / BM;
    const int L = xcd_remap(blockIdx.x, ntm * ntn);
    const int m0 = (L / ntn) * BM, n0 = (L % ntn) * BN;

    const int lc = tid & 7, lr = tid >> 3;

    // ---- per-thread loader state -------------------------------------------------------------
    const T* __restrict__ Ag = reinterpret_cast<const T*>(p.A);
    const T* __restrict__ Wg = reinterpret_cast<const T*>(p.W);
    long a_base[NA];
    int a_y[NA], a_x[NA];
    bool a_ok[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int m = m0 + lr + 32 * i;
        a_ok[i] = m < p.M;
        if (AMODE == AMODE_DENSE) {
            a_base[i] = (long)m * p.lda;
            a_y[i] = a_x[i] = 0;
        } else {
            const int hw = p.Hout * p.Wout;
            const int b = m / hw, rem = m - b * hw;
            const int yo = rem / p.Wout, xo = rem - yo * p.Wout;
            a_base[i] = (long)b * p.Hin * p.Win;
            a_y[i] = yo * p.stride - p.pad;
            a_x[i] = xo * p.stride - p.pad;
        }
    }
    long b_base[NB];
    bool b_ok[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int n = n0 + lr + 32 * i;
        b_ok[i] = n < p.N;
        b_base[i] = (long)n * p.Kpad;
    }
    const int He = p.Hin << p.upsample, We = p.Win << p.upsample;

    u32x4 ra[NA], rb[NB];
    auto issue_loads = [&](int k0) {
        const int kk = k0 + lc * EPC;
        if (AMODE == AMODE_DENSE) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                ra[i] = u32x4{0, 0, 0, 0};
                if (a_ok[i] && kk < p.K) ra[i] = *reinterpret_cast<const u32x4*>(Ag + a_base[i] + kk);
            }
        } else {
            const int tap = kk / p.Cin, ci = kk - tap * p.Cin;
            const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                int yy = a_y[i] + ky, xx = a_x[i] + kx;
                const bool inb = a_ok[i] && (kk < p.K) && yy >= 0 && yy < He && xx >= 0 && xx < We;
                yy >>= p.upsample;
                xx >>= p.upsample;
                ra[i] = u32x4{0, 0, 0, 0};
                if (inb) ra[i] = *reinterpret_cast<const u32x4*>(Ag + (a_base[i] + (long)yy * p.Win + xx) * p.Cin + ci);
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            rb[i] = u32x4{0, 0, 0, 0};
            if (b_ok[i] && kk < p.K) rb[i] = *reinterpret_cast<const u32x4*>(Wg + b_base[i] + kk);  // Kpad = row stride of W
        }
    };
    auto write_lds = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int row = lr + 32 * i;
            *reinterpret_cast<u32x4*>(As + buf * BM * 128 + row * 128 + ((lc ^ (row & 7)) << 4)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int row = lr + 32 * i;
            *reinterpret_cast<u32x4*>(Bs + buf * BN * 128 + row * 128 + ((lc ^ (row & 7)) << 4)) = rb[i];
        }
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // split-K: blockIdx.y owns the K stages [kt0, nk); partial sums go to fp32 slabs, ffn's reduce kernel finishes
    const int nk_all = (p.K + BKE - 1) / BKE;
    const int spp = (nk_all + (int)gridDim.y - 1) / (int)gridDim.y;
    const int kt0 = blockIdx.y * spp;
    const int nk = min(nk_all, kt0 + spp);
    issue_loads(kt0 * BKE);
    write_lds(kt0 & 1);
    __syncthreads();

    for (int kt = kt0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) issue_loads((kt + 1) * BKE);
        const char* Ab = As + buf * BM * 128;
        const char* Bb = Bs + buf * BN * 128;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4 fa[FM], fb[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int row = wm * WM + i * 16 + l15;
                fa[i] = *reinterpret_cast<const u32x4*>(Ab + row * 128 + (((4 * s + g) ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int row = wn * WN + j * 16 + l15;
                fb[j] = *reinterpret_cast<const u32x4*>(Bb + row * 128 + (((4 * s + g) ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    if (SWAP)
                        DT<T>::mma(fb[j], fa[i], acc[i][j]);
                    else
                        DT<T>::mma(fa[i], fb[j], acc[i][j]);
                }
        }
        if (kt + 1 < nk) write_lds(buf ^ 1);
        __syncthreads();
    }

    igemm_epilogue<T, FM, FN, SWAP>(p, acc, m0 + wm * WM, n0 + wn * WN, l15, g);
}

// ---------------------------------------------------------------------------------------------------------------------
// igemm_glds_kernel: same tile / fragment / epilogue scheme as igemm_kernel, but both operand tiles are staged with
// direct-to-LDS loads (global_load_lds_dwordx4): no VGPR round trip and no ds_write (on gfx950 a ds_write_b128 stream
// delivers only ~80 B/clk/CU, which made the register-staged loader's LDS writes as expensive as the MFMAs).
// The LDS image must be lane-linear per wave instruction (8 rows x 128 B = 1 KiB), so the XOR swizzle is applied to
// the SOURCE address: the lane that owns LDS slot c of row r fetches global chunk c ^ (r & 7).  Out-of-range chunks
// (conv zero padding, M/N/K tails) are fetched from a 16-byte zero page.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) const uint32_t g_zero_chunk[4] = {0, 0, 0, 0};
// the streaming loader's out-of-range pointers walk forward 128 B per stage like the real ones: a zero PAGE long enough for one
// conv tap / one dense K row (64 KiB covers 32768 bf16 or 16384 f32 elements), so no per-piece stride register is needed
__device__ __attribute__((aligned(128))) const uint32_t g_zero_page[16384] = {0};

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// NS = LDS ring depth (NS-1 stages in flight), NWM x NWN = wave grid over the BM x BN tile (4 or 8 waves)
// FASTK: the caller guarantees K % 64-byte-stage == 0 (dense) / Cin % stage == 0 (conv): the streaming loader below is used
// X3 (FFN_BF16X3, "split-bf16"): every fp32 value is carried as hi + lo (two bf16) and a product as hi*hi + hi*lo + lo*hi on the bf16
// MFMA with fp32 accumulation.  Nothing changes in the multiplier: the GEMM simply runs over a VIRTUAL contraction of 3K --
//   "plane order" (p.x3 == 1; any K % 8 == 0):  A pair rows [hi(K) | lo(K)] read as segments [A_hi | A_hi | A_lo], W packed [N][3K] =
//      [W_hi | W_lo | W_hi] (conv: per tap)                                                            (p.K holds 3K, Kr = K)
//   "blocked" (p.x3 == 2; K, conv: Cin, % 32 == 0):  A and W both as 128-byte blocks [hi(32) | lo(32)] per 32 elements (W: [N][2K]); the three
//      segments are taken per block (see the loader).  This is the layout the ping-pong kernel's split-bf16 core (igemm_p8.h) streams.
// so only the loader's K -> (tap, column) map differs.  Output / residual are fp32 (epilogue instantiated for float).
// F8 (FFN_FP8): A and W hold OCP e4m3 bytes; the library hands the kernels a bf16-SHAPED view of the problem (K, Cin, lda, Kpad in units of
// two bytes), so nothing in the addressing changes -- only the multiply (mma_fp8: two fp8 MFMAs per 16-byte chunk pair) and the scale
// p.alpha the epilogue already applies.  out / residual are bf16.
template <typename T, int BM, int BN, int AMODE, bool SWAP, int NS = 2, int NWM = 2, int NWN = 2, bool FASTK = false, bool X3 = false, bool F8 = false>
__global__ __launch_bounds__(64 * NWM * NWN) void igemm_glds_kernel(const IgemmParams p) {
    typedef typename std::conditional<X3, float, T>::type TO;      // element type of out / residual
    static_assert(!X3 || (sizeof(T) == 2 && !FASTK && NS == 2), "split-bf16: bf16 operands, recomputing loader");
    static_assert(!F8 || (sizeof(T) == 2 && !X3), "fp8 operands ride the bf16 byte geometry");
    constexpr int EPC = DT<T>::EPC;
    constexpr int BKE = 8 * EPC;  // K elements per stage (128 bytes)
    constexpr int NW = NWM * NWN;
    constexpr int WM = BM / NWM, WN = BN / NWN;
    constexpr int FM = WM / 16, FN = WN / 16;
    constexpr int GA = BM / 8, GB = BN / 8;                      // 8-row groups (one wave-instruction each) per stage
    constexpr int NA = (GA + NW - 1) / NW, NB = (GB + NW - 1) / NW;  // wave-instructions per wave per stage (last one may be idle)
    static_assert(BM % 16 == 0 && BN % 16 == 0 && BM % NWM == 0 && BN % NWN == 0 && WM % 16 == 0 && WN % 16 == 0, "tile / wave grid mismatch");
    static_assert(FM >= 1 && FN >= 1, "tile too small for this wave grid");
    static_assert(NS == 2 || (GA % NW == 0 && GB % NW == 0), "counted vmcnt waits need the same number of loads in every wave");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                    // [NS][BM][128]
    char* Bs = smem + NS * BM * 128;    // [NS][BN][128]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;
    const int l15 = lane & 15, g = lane >> 4;

    // PERSISTENT over output tiles: block b handles logical tiles xcd_remap(b, G) + j*G, j = 0, 1, ... (G = gridDim.x): at every
    // step j the G resident blocks cover a contiguous run of logical tiles, XCD-contiguous inside it (neighbours share operand
    // panels in that XCD's L2).  The stage ring runs straight across tile boundaries: the first stage of the NEXT tile is
    // requested before the last stage of the current one is multiplied, so the epilogue's stores and the next tile's first load
    // overlap (measured: 0-7% on the K = 320 / 640 Linear layers, neutral on the long-K convolutions).
    const int ntn = (p.N + BN - 1) / BN;
    const int ntm = (p.M + BM - 1) / BM;
    const int ntiles = ntm * ntn;
    const int G = gridDim.x;
    int tile = xcd_remap(blockIdx.x, G);
    int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;

    const int lrow = lane >> 3;             // row inside this wave-instruction's 8-row group
    const int csrc = (lane & 7) ^ lrow;     // global chunk this lane fetches (source-side swizzle)
    const char* zero = reinterpret_cast<const char*>(g_zero_chunk);

    const T* __restrict__ Ag = reinterpret_cast<const T*>(p.A);
    const T* __restrict__ Wg = reinterpret_cast<const T*>(p.W);
    long a_base[NA];
    int a_y[NA], a_x[NA];
    bool a_ok[NA];
    long b_base[NB];
    bool b_ok[NB];
    auto prep = [&](int mt, int nt) {       // loader state of the tile whose stages are requested next
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int m = mt + 8 * (wave + NW * i) + lrow;
            a_ok[i] = m < p.M;
            if (AMODE == AMODE_DENSE) {
                a_base[i] = (long)m * p.lda;
                a_y[i] = a_x[i] = 0;
            } else {
                const int hw = p.Hout * p.Wout;
                const int b = m / hw, rem = m - b * hw;
                const int yo = rem / p.Wout, xo = rem - yo * p.Wout;
                a_base[i] = (long)b * p.Hin * p.Win;
                a_y[i] = yo * p.stride - p.pad;
                a_x[i] = xo * p.stride - p.pad;
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int n = nt + 8 * (wave + NW * i) + lrow;
            b_ok[i] = n < p.N;
            b_base[i] = (long)n * p.Kpad;
        }
    };
    prep(m0, n0);
    const int He = p.Hin << p.upsample, We = p.Win << p.upsample;

    const int Kr = X3 ? p.K / 3 : p.K;           // the real contraction length (dense)
    const int pix = X3 ? p.lda : p.Cin;           // conv: elements per input pixel (pair format: hi and lo planes side by side)
    auto issue = [&](int kt, int buf) {
        const int kk = kt * BKE + csrc * EPC;
        const bool kin = kk < p.K;
        // X3 with blocked operands (p.x3 == 2): the virtual contraction visits, per 32-element block b of the real K, the three segments
        //   seg 0: A_hi[b] x W_hi[b]     seg 1: A_hi[b] x W_lo[b]     seg 2: A_lo[b] x W_hi[b]
        // (96 virtual elements per block); A and W both store the block as [hi(32) | lo(32)] (64 elements), so kk -> (b, seg, e) gives
        // A offset 64 b + e + (seg == 2 ? 32 : 0) and W offset 64 b + e + (seg == 1 ? 32 : 0).  A 16-byte chunk (8 elements) never straddles.
        int kb = kk;                                   // W column of this lane's chunk
        if (AMODE == AMODE_DENSE) {
            int ka = kk;
            if (X3 && p.x3 == 2) {
                const int blk = kk / 96, w = kk - blk * 96, seg = w >> 5, e = w & 31;
                ka = blk * 64 + e + (seg == 2 ? 32 : 0);
                kb = blk * 64 + e + (seg == 1 ? 32 : 0);
            } else if (X3) {                      // plane order: segment 0, 1: hi plane; segment 2: lo plane (a chunk never straddles: Kr % 8 == 0)
                const int seg = (kk >= Kr) + (kk >= 2 * Kr);
                ka = kk - seg * Kr + (seg == 2 ? p.a_lo : 0);
            }
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                if (GA % NW != 0 && wave + NW * i >= GA) continue;
                const char* src = (a_ok[i] && kin) ? reinterpret_cast<const char*>(Ag + a_base[i] + ka) : zero;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(As + buf * BM * 128 + (8 * (wave + NW * i)) * 128), 16, 0, 0);
            }
        } else {
            int tap, ci;
            if (X3 && p.x3 == 2) {
                const int c3 = 3 * p.Cin;
                tap = kk / c3;
                const int r = kk - tap * c3, blk = r / 96, w = r - blk * 96, seg = w >> 5, e = w & 31;
                ci = blk * 64 + e + (seg == 2 ? 32 : 0);
                kb = tap * 2 * p.Cin + blk * 64 + e + (seg == 1 ? 32 : 0);
            } else if (X3) {
                const int c3 = 3 * p.Cin;
                tap = kk / c3;
                const int r = kk - tap * c3;
                const int seg = (r >= p.Cin) + (r >= 2 * p.Cin);
                ci = r - seg * p.Cin + (seg == 2 ? p.a_lo : 0);
            } else {
                tap = kk / p.Cin;
                ci = kk - tap * p.Cin;
            }
            const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                if (GA % NW != 0 && wave + NW * i >= GA) continue;
                int yy = a_y[i] + ky, xx = a_x[i] + kx;
                const bool inb = a_ok[i] && kin && yy >= 0 && yy < He && xx >= 0 && xx < We;
                yy >>= p.upsample;
                xx >>= p.upsample;
                const char* src = inb ? reinterpret_cast<const char*>(Ag + (a_base[i] + (long)yy * p.Win + xx) * pix + ci) : zero;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(As + buf * BM * 128 + (8 * (wave + NW * i)) * 128), 16, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (GB % NW != 0 && wave + NW * i >= GB) continue;
            const char* src = (b_ok[i] && kin) ? reinterpret_cast<const char*>(Wg + b_base[i] + kb) : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Bs + buf * BN * 128 + (8 * (wave + NW * i)) * 128), 16, 0, 0);
        }
    };

    // ---- streaming loader (FASTK: every 128-byte K stage lies inside one conv tap / inside K): per wave-instruction the lane keeps
    // a running source pointer and a stride (128 B, or 0 while it reads the zero page), so a stage costs one 64-bit add per piece
    // instead of an integer division, the pixel / bounds arithmetic and two selects (the ISA of the recomputing loader showed ~120
    // scalar+vector instructions per stage per wave beside 20 MFMAs: the convolutions were ISSUE bound, 39% MFMA busy).
    constexpr bool fastk = FASTK && NS == 2;
    const char* a_cur[NA];
    const char* b_cur[NB];
    const char* zpage = reinterpret_cast<const char*>(g_zero_page) + csrc * 16;
    int tap_cur = 0, ci_cur = 0;                     // conv: tap of the stage the pointers stand on, channel offset inside it
    auto set_tap = [&](int tap, int ci) {            // conv: pointers of every A row for (tap, channel offset ci)
        const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            int yy = a_y[i] + ky, xx = a_x[i] + kx;   // rows past M carry a_y = INT_MIN/2: they fail the bounds test
            const bool inb = (unsigned)yy < (unsigned)He && (unsigned)xx < (unsigned)We;
            yy >>= p.upsample;
            xx >>= p.upsample;
            const long pix = a_base[i] + (long)yy * p.Win + xx;
            a_cur[i] = inb ? reinterpret_cast<const char*>(Ag + pix * p.Cin + ci + csrc * EPC) : zpage;
        }
        tap_cur = tap;
        ci_cur = ci;
    };
    auto seek = [&](int kt) {                        // pointers of the tile prepared by prep() at K stage kt
        const int kk = kt * BKE;
        if (AMODE == AMODE_DENSE) {
#pragma unroll
            for (int i = 0; i < NA; ++i) a_cur[i] = a_ok[i] ? reinterpret_cast<const char*>(Ag + a_base[i] + kk + csrc * EPC) : zpage;
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i)
                if (!a_ok[i]) a_y[i] = -(1 << 30);
            const int tap = kk / p.Cin;
            set_tap(tap, kk - tap * p.Cin);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) b_cur[i] = b_ok[i] ? reinterpret_cast<const char*>(Wg + b_base[i] + kk + csrc * EPC) : zpage;
    };
    auto issue_next = [&](int buf) {                 // request the stage the pointers stand on, then advance them by one stage
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (GA % NW != 0 && wave + NW * i >= GA) continue;
            __builtin_amdgcn_global_load_lds((gptr_t)a_cur[i], (lptr_t)(As + buf * BM * 128 + (8 * (wave + NW * i)) * 128), 16, 0, 0);
            a_cur[i] += 128;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (GB % NW != 0 && wave + NW * i >= GB) continue;
            __builtin_amdgcn_global_load_lds((gptr_t)b_cur[i], (lptr_t)(Bs + buf * BN * 128 + (8 * (wave + NW * i)) * 128), 16, 0, 0);
            b_cur[i] += 128;
        }
        if (AMODE != AMODE_DENSE) {
            ci_cur += BKE;
            if (ci_cur >= p.Cin) set_tap(tap_cur + 1, 0);      // wave-uniform: every Cin/64 stages
        }
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf) {
        const char* Ab = As + buf * BM * 128;
        const char* Bb = Bs + buf * BN * 128;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4 fa[FM], fb[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int row = wm * WM + i * 16 + l15;
                fa[i] = *reinterpret_cast<const u32x4*>(Ab + row * 128 + (((4 * s + g) ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int row = wn * WN + j * 16 + l15;
                fb[j] = *reinterpret_cast<const u32x4*>(Bb + row * 128 + (((4 * s + g) ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    if constexpr (F8) {
                        if (SWAP) mma_fp8(fb[j], fa[i], acc[i][j]);
                        else mma_fp8(fa[i], fb[j], acc[i][j]);
                    } else if (SWAP)
                        DT<T>::mma(fb[j], fa[i], acc[i][j]);
                    else
                        DT<T>::mma(fa[i], fb[j], acc[i][j]);
                }
        }
    };

    const int nk_all = (p.K + BKE - 1) / BKE;
    const int spp = (nk_all + (int)gridDim.y - 1) / (int)gridDim.y;
    const int kt0 = blockIdx.y * spp;
    const int nk = min(nk_all, kt0 + spp);

    // ring of NS LDS stages, NS-1 of them in flight; every issue() is exactly NLD wave-instructions, so "stage kt has landed"
    // is a COUNTED wait (vmcnt = NLD x stages issued after it), the DMA of later stages stays in flight across the barrier
    constexpr int NLD = NA + NB;
    if constexpr (NS == 2) {
        if (kt0 < nk) {
            if constexpr (fastk) {
                seek(kt0);
                issue_next(0);
            } else {
                issue(kt0, 0);
            }
        }
        int buf = 0;
        while (true) {
            const int next = tile + G;
            for (int kt = kt0; kt < nk; ++kt) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();   // stage kt visible to all waves; all waves are done with the previous stage -> its buffer is free
                if (kt + 1 < nk) {
                    if constexpr (fastk) issue_next(buf ^ 1);
                    else issue(kt + 1, buf ^ 1);
                } else if (next < ntiles) {     // last stage of this tile: request the next tile's first stage before multiplying
                    prep((next / ntn) * BM, (next % ntn) * BN);
                    if constexpr (fastk) {
                        seek(kt0);
                        issue_next(buf ^ 1);
                    } else {
                        issue(kt0, buf ^ 1);
                    }
                }
                compute(buf);
                buf ^= 1;
            }
            igemm_epilogue<TO, FM, FN, SWAP>(p, acc, m0 + wm * WM, n0 + wn * WN, l15, g);
            if (next >= ntiles) break;
            tile = next;
            m0 = (tile / ntn) * BM;
            n0 = (tile % ntn) * BN;
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    } else {                                    // deeper rings: one tile per block (launched with gridDim.x == number of tiles)
#pragma unroll
        for (int s = 0; s < NS - 1; ++s)
            if (kt0 + s < nk) issue(kt0 + s, s);
        int buf = 0;
        for (int kt = kt0; kt < nk; ++kt) {
            const int ahead = min(NS - 2, nk - 1 - kt);     // stages issued after stage kt
            if (NS >= 4 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NLD) : "memory");
            else if (NS >= 3 && ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + NS - 1 < nk) issue(kt + NS - 1, (buf + NS - 1) % NS);
            compute(buf);
            buf = (buf + 1 == NS) ? 0 : buf + 1;
        }
        igemm_epilogue<TO, FM, FN, SWAP>(p, acc, m0 + wm * WM, n0 + wn * WN, l15, g);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// igemm_halo_kernel: 3x3 / stride 1 / pad 1 convolution whose A operand is staged ONCE per 64-channel chunk as the tile's
// pixel HALO and re-read from LDS by all nine taps.  The im2col kernels above move (BM + BN) * 128 B into LDS per (tap, chunk)
// stage although the nine taps of a chunk read almost the same input pixels; they are bound by the L2 -> LDS path (measured:
// 10.9 TB/s, 44 % MFMA busy on the 128x320 tile), so bytes are what count.  Per chunk this kernel moves
// (rows+2)*(width+2)*128 B of A instead of 9*BM*128 B (128-pixel tile on a 64-wide image: 33 KB instead of 144 KB).
//   tile   = BM consecutive output pixels of ONE image = `TR` whole image rows (W <= BM, BM % W == 0, H*W % BM == 0) or a
//            BM-wide piece of one row (W % BM == 0); the caller checks this
//   K loop = (chunk c, tap t), tap fastest: the weight stage of (c, t) is W[:, t*Cin + c*64 .. +64) of the usual packed
//            layout -- no repacking, only a different walk; fp32 summation order differs from the im2col kernels
//   LDS    = 2 halo buffers [slots][128 B] (slot = halo pixel, 16-byte chunks XOR-swizzled by slot & 7 like the tile rows of
//            the other kernels; chunk c+1 loads while chunk c is multiplied) + the 2-stage weight ring [BN][128 B]
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int BM, int BN, int NWM, int NWN>
__global__ __launch_bounds__(64 * NWM * NWN) void igemm_halo_kernel(const IgemmParams p, int halo_bytes) {
    constexpr int EPC = DT<T>::EPC;
    constexpr int BKE = 8 * EPC;
    constexpr int NW = NWM * NWN;
    constexpr int WM = BM / NWM, WN = BN / NWN;
    constexpr int FM = WM / 16, FN = WN / 16;
    constexpr int GB = BN / 8, NB = (GB + NW - 1) / NW;
    constexpr int NI = 4;                       // halo wave-instructions per wave per chunk (caller guarantees enough)
    static_assert(WM % 16 == 0 && WN % 16 == 0, "tile / wave grid mismatch");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Hs = smem;                            // [2][halo_bytes]
    char* Bs = smem + 2 * halo_bytes;           // [2][BN][128]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;
    const int l15 = lane & 15, g = lane >> 4;
    const int lrow = lane >> 3;
    const int H = p.Hin, W = p.Win, Cin = p.Cin;

    const int ntn = (p.N + BN - 1) / BN;
    const int ntm = p.M / BM;
    const int L = xcd_remap(blockIdx.x, ntm * ntn);
    const int m0 = (L / ntn) * BM, n0 = (L % ntn) * BN;
    // tile geometry
    const int TW = W <= BM ? W : BM, TR = BM / TW;
    const int HW2 = TW + 2, NSLOT = (TR + 2) * HW2, NQ = (NSLOT + 7) >> 3;
    const int img = m0 / (H * W), rem = m0 - img * (H * W);
    const int y0 = rem / W, x0 = rem - y0 * W;

    const T* __restrict__ Ag = reinterpret_cast<const T*>(p.A);
    const T* __restrict__ Wg = reinterpret_cast<const T*>(p.W);
    const char* zpage = reinterpret_cast<const char*>(g_zero_page);

    // halo loader: wave-instruction q covers slots 8q .. 8q+7 (lane>>3 picks the slot, lane&7 the chunk position; the lane
    // fetches source chunk (lane&7) ^ (slot&7)); running pointers, +128 B per chunk, like the streaming loader
    const char* h_cur[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int slot = 8 * (wave + NW * i) + lrow;
        const int hy = slot / HW2, hx = slot - hy * HW2;
        const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
        const bool ok = slot < NSLOT && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
        const int csrc = (lane & 7) ^ (slot & 7);
        h_cur[i] = ok ? reinterpret_cast<const char*>(Ag + ((long)(img * H + yy) * W + xx) * Cin + csrc * EPC) : zpage + csrc * 16;
    }
    auto issue_halo = [&](int hbuf) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (wave + NW * i >= NQ) continue;
            __builtin_amdgcn_global_load_lds((gptr_t)h_cur[i], (lptr_t)(Hs + hbuf * halo_bytes + (8 * (wave + NW * i)) * 128), 16, 0, 0);
            h_cur[i] += 128;
        }
    };
    // weight loader: stage (c, t) = columns [t*Cin + c*64, +64); pointer walk +Cin per tap, -(8*Cin) + 64 at the chunk wrap
    const char* b_cur[NB];
    const int wsrc = (lane & 7) ^ lrow;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int n = n0 + 8 * (wave + NW * i) + lrow;
        b_cur[i] = (n < p.N) ? reinterpret_cast<const char*>(Wg + (long)n * p.Kpad + wsrc * EPC) : nullptr;
    }
    const long tap_step = (long)Cin * sizeof(T), wrap_step = 128 - 8 * tap_step;
    auto issue_w = [&](int wbuf, bool last_tap) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (GB % NW != 0 && wave + NW * i >= GB) continue;
            const char* src = b_cur[i] ? b_cur[i] : zpage + wsrc * 16;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Bs + wbuf * BN * 128 + (8 * (wave + NW * i)) * 128), 16, 0, 0);
            if (b_cur[i]) b_cur[i] += last_tap ? wrap_step : tap_step;
        }
    };

    // per-fragment halo slot of this lane's output pixel at tap (0,0)
    int slot0[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int ml = wm * WM + i * 16 + l15;
        const int r = ml / TW;
        slot0[i] = r * HW2 + (ml - r * TW);
    }

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int hbuf, int wbuf, int tapoff) {
        const char* Hb = Hs + hbuf * halo_bytes;
        const char* Bb = Bs + wbuf * BN * 128;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4 fa[FM], fb[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int slot = slot0[i] + tapoff;
                fa[i] = *reinterpret_cast<const u32x4*>(Hb + slot * 128 + (((4 * s + g) ^ (slot & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int row = wn * WN + j * 16 + l15;
                fb[j] = *reinterpret_cast<const u32x4*>(Bb + row * 128 + (((4 * s + g) ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) DT<T>::mma(fb[j], fa[i], acc[i][j]);
        }
    };

    const int nc = Cin / BKE;
    issue_halo(0);
    issue_w(0, false);
    int hbuf = 0, wbuf = 0;
    for (int c = 0; c < nc; ++c) {
#pragma unroll 1
        for (int t = 0; t < 9; ++t) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();       // this stage's weights (and, at t == 0, this chunk's halo) visible; previous stage's buffers free
            if (t < 8 || c + 1 < nc) issue_w(wbuf ^ 1, t == 7);      // the stage being requested is (c, t+1), or (c+1, 0) after tap 8
            if (t == 0 && c + 1 < nc) issue_halo(hbuf ^ 1);
            const int ky = t / 3;
            compute(hbuf, wbuf, ky * HW2 + (t - 3 * ky));
            wbuf ^= 1;
        }
        hbuf ^= 1;
    }
    igemm_epilogue<T, FM, FN, true>(p, acc, m0 + wm * WM, n0 + wn * WN, l15, g);
}

// split-K finish: out[m, n..n+3] = epilogue(sum_s slab[s][m][n..n+3])  (same epilogue as the SWAP path of igemm_kernel)
template <typename T>
__global__ __launch_bounds__(256) void igemm_splitk_reduce_kernel(const IgemmParams p, int splitk) {
    const long nq = (long)p.M * (p.N / 4);
    const float* __restrict__ ws = reinterpret_cast<const float*>(p.ws);
    T* __restrict__ outT = reinterpret_cast<T*>(p.out);
    float* __restrict__ outF = reinterpret_cast<float*>(p.out);
    const T* __restrict__ res = reinterpret_cast<const T*>(p.residual);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nq; i += (long)gridDim.x * 256) {
        const int m = (int)(i / (p.N / 4)), n = (int)(i % (p.N / 4)) * 4;
        f32x4 a = *reinterpret_cast<const f32x4*>(ws + (long)m * p.N + n);
        for (int s = 1; s < splitk; ++s) a += *reinterpret_cast<const f32x4*>(ws + ((long)s * p.M + m) * p.N + n);
        const int bb = m / p.rows_per_batch;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float x = a[r] * p.alpha;
            if (p.bias) x += p.bias[n + r];
            if (p.rowbias) x += p.rowbias[(long)bb * p.ldrb + n + r];
            x = ig_activation(x, p.flags);
            v[r] = x;
        }
        if (res) {
            float rr[4];
            load4(res + (long)m * p.ldr + n, rr);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += rr[r];
        }
        if (p.flags & IG_OUT_PAIR)
            store_pair_row4(reinterpret_cast<bf16*>(p.out) + (long)m * p.ldo, n, p.ldo / 2, v);
        else if (p.flags & IG_OUT_F32)
            store4(outF + (long)m * p.ldo + n, v);
        else
            store4(outT + (long)m * p.ldo + n, v);
    }
}
