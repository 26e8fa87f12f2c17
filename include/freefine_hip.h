/* freefine_hip.h -- C ABI of libfreefine_hip.so: the MI355X (gfx950) kernels behind the FreeFine
 * DDIM-inversion + guided-denoising hot path.
 *
 * The reference (CIawevy/FreeFine) has no FFI of its own: its "operator API" for this path is Python that
 * monkey-patches diffusers (src/utils/attention.py:226-564) and calls torch ops.  Each entry point below names
 * the reference code it replaces (paths relative to /root/reference).  INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (e.g. the torch caching allocator); the library never
 *     allocates, frees or retains memory; scratch is passed in explicitly;
 *   - every call enqueues on the hipStream_t given as `stream` (void* so the header needs no HIP include), is
 *     asynchronous with respect to the host and performs no synchronisation -> capturable in a hipGraph.  ONE exception:
 *     ffn_igemm in bf16, the FIRST time it sees a problem shape outside stream capture, times its candidate configurations on the
 *     stream (it synchronises the stream, rewrites `out` several times and holds a process-wide lock while it does); call
 *     ffn_igemm_tune() for every shape during warm-up -- or set FFN_IGEMM_TUNE=0 -- where that must not happen later (worker threads,
 *     latency-sensitive replays, runs that need identical tile choices on every rank);
 *   - return value 0 = success, negative = error (FFN_E*); ffn_last_error() returns a thread-local message;
 *   - dtype selects the element type of activations / weights: FFN_F32 (exact fp32 MFMA, "parity mode") or
 *     FFN_BF16 (bf16 MFMA operands, fp32 accumulation, "fast mode").  Norm parameters, biases, masks-as-weights,
 *     latents and scheduler tensors are always fp32;
 *   - activations are channel-contiguous: [batch, H*W, C].
 */
#ifndef FREEFINE_HIP_H
#define FREEFINE_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { FFN_F32 = 0, FFN_BF16 = 1, FFN_BF16X3 = 2 /* ffn_igemm / ffn_attn: split-bf16 arithmetic, fp32 results (see ffn_igemm) */,
       FFN_FP8 = 3 /* ffn_igemm, 3x3 convolutions only: OCP e4m3 operands, bf16 results (see ffn_igemm) */ };
enum { FFN_OK = 0, FFN_EINVAL = -22, FFN_ENOSYS = -38, FFN_EHIP = -5 };

int ffn_version(void);
const char* ffn_last_error(void);
/* fills name (<=64 bytes) with the gcnArchName of `device`, returns CU count or negative error. */
int ffn_device_info(int device, char* name, int name_len);
/* hipGraphLaunch(graph_exec, stream) through this library's HIP runtime.  Why it is in the ABI: a foreign-function call releases the host
 * language's interpreter lock, torch.cuda.CUDAGraph.replay() does not -- and on ROCm 7.2 a launch of a ~380-node UNet-forward graph holds the
 * calling thread for milliseconds, so the one-image-per-call layout (several host threads, one HIP stream each: the reference's
 * src/demo/model.py:577-617 loop run as it stands) serialised on that lock.  graph_exec = the hipGraphExec_t of a captured forward
 * (torch: CUDAGraph.raw_cuda_graph_exec()). */
int ffn_graph_launch(void* stream, void* graph_exec);

/* ---- implicit GEMM: Linear / 1x1 conv (dense A) and 3x3 conv (im2col gather) ------------------------------
 * out[m,n] = epi( alpha * sum_k A(m,k) W[n,k] ), W is [N][Kpad]: K contiguous, Kpad = row stride of W (>= K, multiple of
 * the 16-byte chunk: 4 f32 / 8 bf16); columns >= K are never read, so an activation can serve as W (attention-as-GEMM).
 * Replaces: diffusers ResnetBlock2D/Downsample2D/Upsample2D/Linear modules reached from override_forward
 * (src/utils/attention.py:105-214) and to_q/to_k/to_v/to_out in the hooked Attention.forward (attention.py:372-407). */
enum {
    FFN_IG_OUT_SILU = 1 << 0,
    FFN_IG_OUT_F32 = 1 << 1,
    FFN_IG_GEGLU = 1 << 2,         /* N = 2*Nout; 16-column blocks alternate hidden/gate; out = hidden * gelu(gate) */
    FFN_IG_OUT_TRANSPOSED = 1 << 3, /* out[b][n][s], row stride ldo, m = b*rows_per_batch + s (V^T for ffn_attn) */
    FFN_IG_OUT_PAIR = 1 << 4,       /* FFN_BF16X3 only: out is the bf16 PAIR form [M][ldo] of rows of C = ldo/2 columns (layout: FFN_BF16X3 below) -- the A
                                       operand of the next FFN_BF16X3 GEMM (the GEGLU projection feeding ff.net.2; round 6: ff.net.2 + residual feeding proj_out -- bias, row
                                       bias and the fp32 residual are added before the split; out must not alias the residual) */
    FFN_IG_OUT_GELU = 1 << 5,       /* out = gelu_erf(acc + bias) (+ residual): the MLP of the DINOv2 blocks (dinov2/layers/mlp.py:31-38) */
    FFN_IG_OUT_KV64 = 1 << 7,       /* FFN_BF16X3 only (round 6): the projection writes the attention kernels' PRE-SPLIT K / V^T images itself (what ffn_attn_presplit
                                       produces from an fp32 K / V^T: one fp32 round trip and one HBM-bound pass less per self-attention block).  Row-major output:
                                       columns c >= kv64_from of a row are stored, per 64 columns (one head), as [hi(64) | lo(64)] bf16 in the bytes their fp32 values
                                       would occupy (columns below kv64_from stay fp32: the q half of the fused q | k projection; kv64_from and N %% 64 == 0).
                                       FFN_IG_OUT_TRANSPOSED: every run of 64 consecutive positions s of a row out[b][n][.] is stored as [hi(64) | lo(64)]
                                       (rows_per_batch %% 64 == 0).  Plain epilogue only (bias allowed), no split-K.  ffn_attn reads them with kv_pair = 1. */
    FFN_IG_OUT_RELU = 1 << 6        /* out = max(acc + bias, 0) (+ residual): the DPT head's ResidualConvUnit / output convs (depth_anything/blocks.py:68-78,
                                       dpt.py:93-98).  SILU / GELU / RELU are mutually exclusive and exclude GEGLU and the transposed output */
};
typedef struct ffn_igemm_desc {
    const void* A;        /* dense: [M][lda];  conv: NHWC input [B][Hin][Win][Cin] */
    const void* W;        /* [N][Kpad] */
    void* out;            /* [M][ldo] (or transposed) */
    const float* bias;    /* [N] or NULL (GEGLU: packed like W rows) */
    const float* rowbias; /* [batch][ldrb] or NULL; batch = m / rows_per_batch (time-embedding projection) */
    const void* residual; /* [M][ldr] or NULL */
    int M, N, K, Kpad;
    int lda, ldo, ldr, ldrb;
    int rows_per_batch;
    int Hin, Win, Cin, Hout, Wout, stride, pad, upsample; /* conv only; upsample=1 fuses nearest-2x of the input */
    int flags;
    float alpha;
    int conv; /* 0 = dense, 1 = 3x3 conv, 2 = 2x2 conv: taps (dy, dx) in {0,1}^2 read input pixel (yo - (pad >> 1) + dy, xo - (pad & 1) + dx),
                 K = 4 Cin, stride 1 -- one sub-pixel class of a 3x3 conv behind a nearest-2x upsample (output pixel (2y+a, 2x+b) sees a 2x2
                 window of the low-resolution input, pad = 2 (1 - a) + (1 - b), with the 3x3 taps that coincide summed: 4/9 of the FLOPs);
                 bf16 ping-pong tiles only (Cin % 64 == 0, N % 256 or % 320 == 0, M >= 192) */
    int splitk;       /* 0 = let the library choose (needs ws), 1 = never split, k = force k K-slices */
    void* ws;         /* optional fp32 scratch for split-K partial slabs (>= splitk*M*N*4 bytes) or NULL */
    long ws_bytes;
    int a_lo;         /* FFN_BF16X3: x3 = 0 / 1: column (conv: channel) offset of the lo plane inside a row (pixel) of A; x3 = 2: 32 (the block); else ignored */
    int x3;           /* FFN_BF16X3: operand layout -- 0 / 1 = planes (A [hi(K) | lo(K)], W [W_hi | W_lo | W_hi] over the whole K, conv: per tap);
                         2 = blocked: A and W as 128-byte blocks [hi(32) | lo(32)] per 32 elements of K (needs K, conv: Cin, % 32 == 0; what the
                         ping-pong tile's split-bf16 core streams) */
    int f8;           /* set by the library from `dtype` (callers leave it 0) */
    int kv64_from;    /* FFN_IG_OUT_KV64, row-major output: first column written in the pre-split form (0 = the whole row) */
} ffn_igemm_desc;
int ffn_igemm(void* stream, int dtype, const ffn_igemm_desc* d);
/* FFN_BF16X3 ("split-bf16", the fast mode that keeps fp32-level results): every fp32 operand value v is carried as hi = bf16(v) and
 * lo = bf16(v - hi) (16-17 significant bits together) and a product a*w is evaluated as a_hi*w_hi + a_hi*w_lo + a_lo*w_hi on the bf16
 * MFMA with fp32 accumulation -- 3 MFMAs per product term (ceiling 2.5 PFLOP/s / 3 = 833 TFLOP/s against 157 TFLOP/s of the fp32
 * MFMA), dropped term a_lo*w_lo ~ 2^-18.  Operand formats:
 *   the PAIR form of a row (pixel) of C fp32 values = 2C bf16.  C % 32 == 0: BLOCKED -- every 32 columns are one 128-byte block
 *        [hi(32) | lo(32)] (column c: hi at element 64 (c / 32) + c % 32, lo 32 elements behind it), so that a 32-deep K stage of a GEMM is
 *        ONE whole line carrying both planes (round 5; rounds 3-4 kept two planes and streamed the hi plane twice).  Otherwise one block of C
 *        columns, i.e. the planes [hi(C) | lo(C)].  ffn_split_pair, the *_pair norms, FFN_IG_OUT_PAIR and ffn_attn's out_pair all write it;
 *   A    pair rows of K columns (conv: pixels of Cin channels), lda = row (pixel) stride in bf16 elements.  desc.x3 = 2 (blocked, K resp.
 *        Cin % 32 == 0): a_lo = 32.  desc.x3 = 0 / 1 (planes): lo plane at columns [a_lo, a_lo + K);
 *   W    bf16 [N][Kpad].  x3 = 2: the pair form of each fp32 weight row, conv: of each tap's Cin columns (Kpad >= 2K; column
 *        k' = (tap*Cin/32 + block)*64 + {0, 32} + e).  x3 = 0 / 1: Kpad >= 3K, [W_hi | W_lo | W_hi] (conv: per tap, k' = (tap*3 + seg)*Cin + ci);
 *   out, residual   fp32 (FFN_IG_OUT_F32 is implied); bias / rowbias fp32 as always; GEGLU, SILU, transposed output, split-K as in bf16.
 * K in the descriptor is the REAL contraction length (dense K, conv 9*Cin). */
int ffn_split_pair(void* stream, const float* src, void* dst, long rows, int C, int ld_src);
/* 3x3 / stride 1 / pad 1 convolution with FOUR output channels (the UNet's conv_out): x NHWC [B][H*W][Cin] (FFN_F32 or FFN_BF16 elements), w fp32
 * [4][9][Cin] ((ky, kx, ci) order), bias fp32 [4] or NULL, out fp32 [B][H*W][4].  Direct fp32 convolution on the vector ALU (exact fp32 products in
 * every mode) instead of 1/32 of an MFMA tile; Cin % 16 == 0, Cin <= 448 (weights in LDS). */
int ffn_conv3x3_n4(void* stream, int dtype, const void* x, const float* w, const float* bias, float* out, int B, int H, int W, int Cin);
/* FFN_FP8 (3x3 convolutions of the bf16 fast mode, reported beside it -- not a parity mode): A = e4m3 bytes NHWC [B][Hin][Win][Cin] with
 * Cin a multiple of 16 (the ping-pong tile: of 128; ffn_groupnorm_f8 writes such a tensor, channels zero-padded), W = e4m3 [N][Kpad]
 * in the usual (ky, kx, ci) order, K = 9 * Cin; out / residual bf16, bias / rowbias fp32.  Both operands carry power-of-two scales (the
 * activation's is ffn_groupnorm_f8's `qscale`, the weight's is chosen at pack time); `alpha` = 1 / (their product) un-scales the result.
 * The kernels are the bf16 ones: e4m3 rides the same 16-byte chunk geometry (16 values per chunk, 128 per K row), each fragment pair
 * takes two v_mfma_f32_16x16x32_fp8_fp8 -- twice the contraction per staged byte at the bf16 MFMA rate. */
int ffn_groupnorm_f8(void* stream, const void* x_bf16, void* y_e4m3, const float* gamma, const float* beta, int B, int HW, int C, int Cp,
                     int G, float eps, int silu, float qscale, float* partial_ws, float* scale, float* shift);
/* Kernel families behind ffn_igemm (freefine_amd/csrc): igemm_pp_kernel (igemm_p8.h; bf16 -- 256- or 192-row "ping-pong" tiles with
 * LDS-DMA operands in flight across barriers, the default wherever N is a multiple of 256 or 320 and K a multiple of 64),
 * igemm_glds_kernel / igemm_halo_kernel (igemm.h; every other bf16 shape and all of f32).
 * bf16 problems: the first time a problem shape is seen outside stream capture, ffn_igemm times its few plausible (tile, K-split)
 * configurations on the caller's stream (this one call synchronises the stream and launches the kernel several times: `out` must
 * not alias `residual`) and caches the winner; FFN_IGEMM_TUNE=0 in the environment keeps the deterministic rule-based choice
 * (f32 always uses it).  Testing hooks: the number of bf16 configurations, and forcing one (-1 = off; returns the previous value). */
/* explicit warm-up entry point: tunes (or looks up) the configuration for `d` now, exactly as the first ffn_igemm call would --
 * it IS a full ffn_igemm call: `out` is written (several times while candidates are timed) and holds the winner's result */
int ffn_igemm_tune(void* stream, int dtype, const ffn_igemm_desc* d);
/* the tuned table as data: entries of ffn_igemm_tune_entry_ints() ints (problem key, configuration, K split).  export returns the
 * number of entries in the table (copies at most max_entries); import merges entries (unknown configurations are skipped) and returns
 * how many it took.  Use: persist the table across processes, or broadcast rank 0's table in a multi-GPU run so that every rank
 * launches identical configurations. */
int ffn_igemm_tune_entry_ints(void);
int ffn_igemm_tune_export(int* buf, int max_entries);
int ffn_igemm_tune_import(const int* buf, int n_entries);
/* Entries carry a stamp of the build that wrote them (table layout, configuration list, gfx950): import ignores foreign ones, and
 * an imported entry is launched only after its (configuration, K split) was found among the candidates THIS build offers for the
 * actual problem (workspace capacity, split legality, tile applicability); otherwise it is dropped and the problem re-tuned.
 * ffn_igemm_tune_clear empties the table (returns the number of entries dropped); ffn_igemm_tune_enable(0) stops timing-based
 * tuning in this process -- problems not in the table then take the deterministic rule (returns the previous setting).  A sharded
 * run wanting bit-identical bf16 results on every rank: clear + import rank 0's table, then ffn_igemm_tune_enable(0). */
int ffn_igemm_tune_clear(void);
int ffn_igemm_tune_enable(int on);
int ffn_igemm_tune_stamp(void);   /* first int of every entry this build exports */
int ffn_igemm_num_configs(void);
int ffn_igemm_force_config(int cfg);
/* which tile (BM x BN) ffn_igemm dispatches for this problem -- lets a profiler name the kernel instantiation */
int ffn_igemm_variant(const ffn_igemm_desc* d, int* bm, int* bn);
/* the kernel instantiation ffn_igemm launches for this problem, spelled like rocprofv3's kernel trace */
int ffn_igemm_kernel_name(int dtype, const ffn_igemm_desc* d, char* buf, int len);

/* ---- multi-pass masked attention (the FreeFine attention modulation) -----------------------------------------
 * out[b,q,h,:] = sum_p w_p(b) * wq_p[q] * softmax_k(scale*<Q[qrow_p(b),q,h],K[kvrow_p(b),k,h]> + mask_p(q,k)) V[kvrow_p(b),k,h]
 * Replaces: Attention_Modulator.Temporal_contextal_attention{,_bg,_compose}, modulate_local_cross_attn{,_bg,_compose},
 * style_align_share_attention, mask_attention, get_attention_scores, get_cross_hidden_state and the prepare_*_mask
 * builders (src/utils/attention.py:774-1432) plus the plain branch of ca_forward (attention.py:394-404). */
#define FFN_ATT_MAXP 4
#define FFN_ATT_MAXB 16
enum { FFN_ATT_HEAD_RULE = 1, FFN_ATT_UNIFORM_SEL1 = 2, FFN_ATT_UNIFORM_SEL0 = 4 };
typedef struct ffn_attn_entry {
    int q_row, kv_row;      /* batch rows supplying Q and K/V for this (pass, output row) */
    float w_const, w_slope; /* weight = w_const + w_slope * (*w_dev); both 0 -> entry skipped */
    const float* wq;        /* per-query weight [S] or NULL */
    const uint8_t* kmask;   /* per-key byte mask [Sk] or NULL */
    const uint8_t* qsel;    /* per-query selector [S] or NULL (=1): allowed(q,k) = (kmask[k]!=0) == (qsel[q]!=0) */
    int flags;              /* FFN_ATT_HEAD_RULE: mask applies only where (b*heads+head) is even (attention.py:859 vs 761);
                               FFN_ATT_UNIFORM_SEL1/0: the allowed set for sel=1/0 is empty -> uniform over all keys */
    int hr_row;             /* 0: the tiled-head rule uses the OUTPUT row index; k>0: it uses row k-1 (row-deduplicated batches) */
} ffn_attn_entry;
typedef struct ffn_attn_desc {
    const void* q;      /* [Bq][S][ldq], head h at column h*D */
    const void* k;      /* [Bk][Sk][ldk] */
    const void* vt;     /* [Bk][heads*D][ldvt]  V transposed (ldvt >= Sk, multiple of 8, padding finite) */
    void* out;          /* [Bo][S][ldo] */
    const float* w_dev; /* device scalar (context_guidance) or NULL */
    int Bo, S, Sk, heads, D;
    int ldq, ldk, ldvt, ldo;
    float scale;
    int npass;
    int out_pair;       /* FFN_BF16X3 with D <= 64 only: out is the bf16 PAIR form [Bo][S][ldo] of rows of ldo/2 columns (layout: FFN_BF16X3 above), the A operand of
                           the to_out projection's FFN_BF16X3 GEMM; 0 = fp32 rows */
    int kv_pair;        /* FFN_BF16X3 launches that run attn_x3p_kernel (ffn_attn_kernel_name with kv_pair = 0 says so) only: k / vt are the PRE-SPLIT bf16
                           images ffn_attn_presplit (or a projection with FFN_IG_OUT_KV64) wrote: k[row][key] = heads blocks of [hi(64) | lo(64)], ldk * 4
                           bytes from key to key; vt[row][head * 64 + d] = Sk / 64 blocks of [hi(64 keys) | lo(64 keys)], ldvt * 4 bytes from row to row
                           (i.e. ldk / ldvt are what they would be for the fp32 tensors the images replace: heads * 64 / Sk when compact).  The launch runs
                           attn_x3w_kernel (attention_x3w.h, round 6: one wave per SIMD on 32x32x16 MFMAs) or, with FFN_ATTN_X3W=0 in the environment,
                           attn_x3p_kernel<., PAIRKV>; 0 = fp32 k / vt */
    ffn_attn_entry e[FFN_ATT_MAXP * FFN_ATT_MAXB]; /* entry (p,b) at p*FFN_ATT_MAXB + b */
} ffn_attn_desc;
int ffn_attn(void* stream, int dtype, const ffn_attn_desc* d);
/* dtype FFN_BF16X3: q / k / vt / out are fp32 exactly as with FFN_F32; head sizes D <= 64 run attn_x3_kernel (attention_x3.h: both
 * products in split-bf16 arithmetic, three bf16 MFMAs per term, fp32 softmax) -- or, under attn_pp_kernel's preconditions (D = 64,
 * Sk % 64 == 0, S >= 128, no uniform-softmax entry), attn_x3p_kernel (attention_x3p.h: the same arithmetic in the ping-pong schedule;
 * FFN_ATTN_PP=0 disables it too); larger heads fall back to the exact fp32 kernel. */
/* bf16 launches with D = 64, Sk % 64 == 0, S >= 128 and no degenerate (uniform-softmax) entry run attn_pp_kernel (attention_pp.h:
 * software-pipelined, 8 waves in two alternating groups); everything else attn_kernel (attention.h).  Same results up to fp32
 * summation order.  FFN_ATTN_PP=0 in the environment forces attn_kernel.
 * bf16 launches with D = 64, Sk <= 96 and ONE pass whose entries carry no key mask / selector / per-query weight (the text
 * cross-attention) run xattn_kernel (attention_x.h: K and V^T of a (row, head) in a wave's registers); FFN_ATTN_X=0 disables it. */
/* fp32 K [rows][Sk][ldk] (heads * 64 columns) and V^T [rows][heads * 64][ldvt] -> the bf16 images attn_x3p_kernel stages by LDS-DMA when
 * desc.kv_pair = 1: k_pair [rows][Sk][heads][hi(64) | lo(64)], vt_pair [rows][heads * 64][Sk / 64][hi(64 keys) | lo(64 keys)] (each as many bytes
 * as its fp32 source; Sk % 64 == 0, head dim 64).  Once per attention call: the kernel's 16 query-block workgroups per (row, head) otherwise each
 * split the whole K / V^T in their key loops (16-25 % of the launch).  Reference call sites: the self-attention of the hooked Attention.forward,
 * src/utils/attention.py:394-404, 1043-1091. */
int ffn_attn_presplit(void* stream, const float* k, const float* vt, void* k_pair, void* vt_pair, int rows, int Sk, int heads, int ldk, int ldvt);
/* padded head dim / query fragments per wave of the instantiation ffn_attn dispatches for head dim D */
int ffn_attn_variant(int dtype, int D, int* dp, int* qf);
/* the kernel instantiation ffn_attn launches for this problem, spelled like rocprofv3's kernel trace */
int ffn_attn_kernel_name(int dtype, const ffn_attn_desc* d, char* buf, int len);

/* ---- elementwise / resampling helpers of the depth front end ------------------------------------------ */
/* y = max(a, 0) (FFN_ELT_RELU, b ignored) or y = a + b (FFN_ELT_ADD) over n elements (n % 4 == 0; fp32 or bf16).  Replaces the
 * activation in front of ResidualConvUnit.conv1 and FeatureFusionBlock's skip_add (depth_anything/blocks.py:68, 137-139). */
enum { FFN_ELT_RELU = 0, FFN_ELT_ADD = 1 };
int ffn_eltwise(void* stream, int dtype, int op, const void* a, const void* b, void* y, long n);
/* bilinear resampling of an NHWC tensor [B][Hin][Win][C] -> [B][Hout][Wout][C] with align_corners = True (source coordinate
 * y * (Hin - 1) / (Hout - 1), fp32 weights; Hout == 1 reads row 0): torch.nn.functional.interpolate(mode="bilinear",
 * align_corners=True) as called by depth_anything/blocks.py:147-149 and dpt.py:132, 165.  relu != 0 applies max(., 0) (dpt.py:166). */
int ffn_resize_bilinear(void* stream, int dtype, const void* x, void* y, int B, int Hin, int Win, int Hout, int Wout, int C, int relu);

/* ---- point-cloud warp of the 3-D coarse edit ------------------------------------------------------------
 * The arithmetic of IntegratedP3DTransRasterBlendingFull (src/utils/geo_utils.py:427-528): depth-lifted object pixels -> rigid transform about
 * the cloud's centre -> FoV-perspective projection -> pytorch3d-style disc splat (PointsRasterizer: the K nearest covering points per pixel,
 * AlphaCompositor with weights 1 - d^2 / r^2).  pytorch3d is absent from the build image: the restatement is PARITY UNPINNED (property-tested
 * and checked against oracle/warp3d.py).  All buffers fp32 / int32, device memory.
 *   ffn_splat_lift     pts[n][4] = (-(i - W/2) z / fx, -(j - H/2) z / fy, z, 0) for the masked pixels idx[n] = j * W + i (geo_utils.py:436-457)
 *   ffn_splat_project  proj[n][4] = (x_ndc, y_ndc, z_view, 0): view = ((p - center + translate) . rotate) * scale + center, row vectors,
 *                      ndc = view.xy / (view.z * tan_half_fov)  (geo_utils.py:343-378, 399-425, 478-481)
 *   ffn_splat_bin      tile binning (16 x 16 pixel tiles, tiles = ceil(W/16) * ceil(H/16)): fill = 0 adds to counts[tile] the number of discs
 *                      whose bounding box touches the tile; fill = 1 writes the point ids to list[offs[tile] + k] (counts = zeroed cursor,
 *                      offs = exclusive scan of the counting pass with the total at offs[tiles])
 *   ffn_splat_render   image[H][W][3] (fp32, composited colours), idx_sum[H][W] (sum of the K point ids, -1 per empty slot: what the
 *                      reference's mask test reads, geo_utils.py:517) and covered[H][W] (any point)  (geo_utils.py:482-517) */
typedef struct ffn_splat_xform {
    float center[3], translate[3], rotate[9] /* row-major, applied as p . R */, scale[3];
    float tan_half_fov;
} ffn_splat_xform;
int ffn_splat_lift(void* stream, const float* depth, const int* idx, float* pts, int n, int W, int H, float fx, float fy);
int ffn_splat_project(void* stream, const float* pts, float* proj, int n, const ffn_splat_xform* x);
int ffn_splat_bin(void* stream, int fill, const float* proj, int n, float radius, int W, int H, int* counts, const int* offs, int* list);
int ffn_splat_render(void* stream, const float* proj, const float* rgb, const int* offs, const int* list, float radius, int K, int W, int H,
                     float* image, int* idx_sum, uint8_t* covered);

/* ---- normalisation ------------------------------------------------------------------------------------------- */
/* `silu` of ffn_groupnorm / ffn_gn_apply is a flag word: FFN_NORM_SILU applies SiLU; FFN_NORM_OUT_PAIR (fp32 input only) writes y as the
 * bf16 PAIR rows [B*HW][2C] an FFN_BF16X3 GEMM reads (layout: FFN_BF16X3 above; no separate ffn_split_pair pass).  ffn_layernorm_pair: the same for LayerNorm. */
enum { FFN_NORM_SILU = 1, FFN_NORM_OUT_PAIR = 2 };
/* GroupNorm statistics -> per-(batch,channel) scale/shift (fp32).  partial_ws: >= B*nchunk*2*C floats where
 * nchunk = ffn_gn_nchunk(HW).  Replaces torch GroupNorm inside diffusers blocks (attention.py:105-214). */
int ffn_gn_nchunk(int HW);
/* 1 if ffn_groupnorm handles this problem in one fused launch (no workspace needed), 0 if it needs the partial / scale / shift workspace */
int ffn_gn_fused(int B, int HW, int C, int G);
int ffn_gn_stats(void* stream, int dtype, const void* x, const float* gamma, const float* beta, int B, int HW, int C, int G,
                 float eps, float* partial_ws, float* scale, float* shift);
int ffn_gn_apply(void* stream, int dtype, const void* x, void* y, const float* scale, const float* shift, int B, int HW,
                 int C, int silu);
/* GroupNorm (+ optional SiLU) in one call: a single fused launch when a (batch, group) slice has <= 131072 elements (every UNet
 * GroupNorm at 64x64 latents), else ffn_gn_stats + ffn_gn_apply (workspace pointers may be NULL in the fused case). */
int ffn_groupnorm(void* stream, int dtype, const void* x, void* y, const float* gamma, const float* beta, int B, int HW, int C, int G,
                  float eps, int silu, float* partial_ws, float* scale, float* shift);
/* round 6: GroupNorm (three-launch form, fp32 x) whose apply pass writes BOTH pair tensors a ResBlock with a 1x1 shortcut reads: y = pair(act(GN(x))) for conv1
 * and yraw = pair(x) for the shortcut GEMM (what ffn_split_pair(x) would produce, bit for bit) -- x is read once less.  C % 8 == 0; the workspace is required
 * (also for shapes ffn_gn_fused() would take in one launch); `silu`: FFN_NORM_SILU or 0.  Reference: the norm1 / conv_shortcut inputs of diffusers' ResnetBlock2D
 * (in-tree copy /root/reference/evaluation/DragonDiffusion/src/unet/resnet_2d.py:110). */
int ffn_groupnorm_pair_raw(void* stream, const void* x, void* y_pair, void* yraw_pair, const float* gamma, const float* beta, int B, int HW, int C, int G,
                           float eps, int silu, float* partial_ws, float* scale, float* shift);
int ffn_layernorm(void* stream, int dtype, const void* x, void* y, const float* gamma, const float* beta, int M, int C,
                  float eps);
int ffn_layernorm_pair(void* stream, const float* x, void* y, const float* gamma, const float* beta, int M, int C, float eps);
int ffn_softmax_rows(void* stream, int dtype, const void* x, void* y, long M, int N, float scale);

/* ---- scheduler / guidance elementwise (fp32, NCHW like the reference) ------------------------------------------ */
/* eps = eu + cfg*(ec-eu)*mask   (src/demo/model.py:605-611; mask NULL -> :608) */
int ffn_cfg_masked(void* stream, const float* eps_u, const float* eps_c, const float* mask, float cfg, float* eps, long n,
                   int HW);
/* inv_step (src/demo/model.py:109-132); coefficients are the fp32 values torch computes: c_bt=sqrt(1-a_t),
 * c_at=sqrt(a_t), c_an=sqrt(a_next), c_bn=sqrt(1-a_next) */
int ffn_ddim_inv_step(void* stream, const float* eps, const float* x, float c_bt, float c_at, float c_an, float c_bn,
                      float* x_next, float* pred_x0, long n);
/* ctrl_step (src/demo/model.py:134-198) */
typedef struct ffn_ctrl_step_desc {
    const float* eps;
    const float* x;
    const float* noise; /* NULL when eta == 0 */
    const float* m;     /* float(mask) [HW] */
    const float* om;    /* float(1 - mask) evaluated in the mask's own dtype (uint8 wrap kept) [HW] */
    float* x_prev;
    float* pred_x0; /* may be NULL */
    float c_bt, c_at, c_ap, c_dir;
    float c_dirm[8], stdv[8];
    int row_masked[8];
    int rows, CHW, HW;
} ffn_ctrl_step_desc;
int ffn_ddim_ctrl_step(void* stream, const ffn_ctrl_step_desc* d);

/* ---- layout / misc ----------------------------------------------------------------------------------------------- */
typedef struct ffn_pack_desc {
    const float* src; /* fp32 NCHW [Bsrc][Cl][HW] */
    void* dst;        /* T NHWC [B][HW][CP], channels >= Cl zero */
    int src_row[16];
    int B, Cl, CP, HW;
} ffn_pack_desc;
int ffn_pack_nchw(void* stream, int dtype, const ffn_pack_desc* d);
int ffn_nhwc_to_nchw_f32(void* stream, const float* src, float* dst, int B, int HW, int C, int ld);
/* out[r][0:C1] = a[r], out[r][C1:C1+C2] = b[r].  a == NULL: the left block is already in place (its producer wrote it with
 * ldo = C1 + C2), only b is copied */
int ffn_concat(void* stream, int dtype, const void* a, const void* b, void* out, long rows, int C1, int C2);
int ffn_timestep_embed(void* stream, int dtype, const float* t_dev, const float* freq, void* out, int B, int half, int flip);
int ffn_transpose(void* stream, int dtype, const void* src, void* dst, int B, int R, int C, int ld_src, int ld_dst);
int ffn_cast(void* stream, int src_dtype, int dst_dtype, const void* src, void* dst, long n);
int ffn_image_to_nhwc(void* stream, int dtype, const uint8_t* img, void* dst, long npix, int CP);
int ffn_nhwc_to_image(void* stream, int dtype, const void* src, float* dst, int B, int HW, int ld);

#ifdef __cplusplus
}
#endif
#endif
