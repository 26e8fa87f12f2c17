// C ABI of libfreefine_hip.so (see include/freefine_hip.h).  Host-side launch glue only: argument validation,
// tile selection, dynamic-LDS opt-in.  No allocation, no synchronisation, everything on the caller's stream.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <mutex>
#include <unordered_map>
#include <stdlib.h>
#include <string.h>

#include "../../include/freefine_hip.h"
#include "attention.h"
#include "attention_pp.h"
#include "attention_x.h"
#include "attention_x3.h"
#include "attention_x3p.h"
#include "attention_xx3.h"
#include "conv_small.h"
#include "elementwise.h"
#include "igemm.h"
#include "igemm_p8.h"
#include "norms.h"
#include "splat.h"

static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
static int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FFN_EHIP, "%s: %s", what, hipGetErrorString(e));
    return FFN_OK;
}
// a stale error left behind by an unrelated earlier HIP call (e.g. a tool's probe) must not be blamed on our launch
#define LAUNCH(...)                  \
    do {                             \
        (void)hipGetLastError();     \
        hipLaunchKernelGGL(__VA_ARGS__); \
    } while (0)
#define REQUIRE(cond, ...) \
    do {                   \
        if (!(cond)) return fail(FFN_EINVAL, __VA_ARGS__); \
    } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline int grid_for(long n, int per_block = 256, int cap = 4096) {
    long g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

template <typename K>
static int set_lds(K kernel, int bytes) {
    static std::mutex mu;
    static std::unordered_map<const void*, int> done;   // one opt-in per (kernel, size)
    if (bytes > 48 * 1024) {
        std::lock_guard<std::mutex> lk(mu);
        int& have = done[reinterpret_cast<const void*>(kernel)];
        if (have >= bytes) return FFN_OK;
        have = bytes;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return fail(FFN_EHIP, "hipFuncSetAttribute(%d): %s", bytes, hipGetErrorString(e));
    }
    return FFN_OK;
}

extern "C" int ffn_version(void) { return 1; }
extern "C" const char* ffn_last_error(void) { return g_err; }
extern "C" int ffn_device_info(int device, char* name, int name_len) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return fail(FFN_EHIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (name && name_len > 0) {
        strncpy(name, prop.gcnArchName, name_len - 1);
        name[name_len - 1] = 0;
    }
    return prop.multiProcessorCount;
}

extern "C" int ffn_graph_launch(void* stream, void* graph_exec) {
    REQUIRE(graph_exec, "graph_launch: null graph");
    hipError_t e = hipGraphLaunch(reinterpret_cast<hipGraphExec_t>(graph_exec), reinterpret_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(FFN_EHIP, "hipGraphLaunch: %s", hipGetErrorString(e));
    return FFN_OK;
}

// ---- igemm -------------------------------------------------------------------------------------------------------
static void igemm_plan_for(int dtype, const ffn_igemm_desc& d, int* bm, int* bn, int* splitk);
static int igemm_stages_env() {
    static const int v = [] {
        const char* e = getenv("FFN_IGEMM_STAGES");     // 0/unset = heuristic, 1 = register-staged legacy loader, 2..4 = LDS ring depth
        return e ? atoi(e) : 0;
    }();
    return v;
}
static int device_cus() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    return n;
}
static bool persist_enabled() {
    static const bool on = [] { const char* e = getenv("FFN_IGEMM_PERSIST"); return !(e && atoi(e) == 0); }();
    return on;
}
// persist: the 2-stage glds kernels walk several output tiles per workgroup (igemm.h); launch only as many workgroups as are
// co-resident (LDS / thread limits per CU) and let each stride over the tile list
template <typename K>
static int launch_igemm_kernel(K kern, int lds, hipStream_t s, const ffn_igemm_desc& d, int ntiles, int splitk, int threads = 256,
                               bool persist = false) {
    int rc = set_lds(kern, lds);
    if (rc) return rc;
    int gx = ntiles;
    if (persist && persist_enabled()) {
        int per_cu = (160 * 1024) / (lds > 0 ? lds : 1);
        if (per_cu > 2048 / threads) per_cu = 2048 / threads;
        if (per_cu < 1) per_cu = 1;
        const int cap = (device_cus() * per_cu) / (splitk > 0 ? splitk : 1);
        if (gx > cap && cap >= 8) gx = cap;
    }
    LAUNCH(kern, dim3(gx, splitk), dim3(threads), lds, s, d);
    return check_launch("igemm");
}
static int igemm_waves_env() {
    static const int v = [] {
        const char* e = getenv("FFN_IGEMM_WAVES");      // 0/unset = heuristic, 4 or 8 waves per workgroup
        return e ? atoi(e) : 0;
    }();
    return v;
}
// ring depth and waves per workgroup for a (tile, split) choice.  Measured on MI355X (tools/bench_kernels.py): occupancy beats
// prefetch depth for these compiler-scheduled loops -- 2 workgroups/CU with a 2-deep ring win over 1 workgroup/CU with a 3-4 deep
// ring by ~35%; on the 128x128 tile 8 waves (4 waves/SIMD) beat 4 waves by 5-30%, and 16 waves win once the K loop is short.
static void igemm_exec_cfg(int dtype, const ffn_igemm_desc& d, int bm, int bn, int splitk, int* ns, int* nw) {
    const int kstage = dtype == FFN_F32 ? 32 : 64;
    const int nk = ((d.K + kstage - 1) / kstage + splitk - 1) / splitk;    // K stages per workgroup
    *ns = igemm_stages_env();
    *nw = igemm_waves_env();
    if (*ns == 0) *ns = 2;
    if (*nw == 0) *nw = (bm == 128 && bn == 128 && !d.conv && nk <= 24) ? 16 : 8;
    if (*ns > 2 && nk < 3) *ns = 2;
    if (bm == 64) *nw = 4;
    if (bm == 128 && bn == 64 && *nw > 8) *nw = 8;
    if (*ns == 1) *nw = 4;
}
template <typename T, int BM, int BN, int AMODE, bool SWAP>
static int launch_igemm(hipStream_t s, const ffn_igemm_desc& d, int splitk) {
    constexpr int stage = (BM + BN) * 128;
    const int ntiles = ((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN);
    int ns, nw;
    igemm_exec_cfg(sizeof(T) == 4 ? FFN_F32 : FFN_BF16, d, BM, BN, splitk, &ns, &nw);
    int rc = FFN_OK;
    bool done = false;
    if constexpr (BM == 128 && BN == 128) {
        if (nw == 8 && ns >= 2) {   // 8 waves (2x4) on the 128x128 tile: 2 waves/SIMD from ONE workgroup, so a deep ring fits the LDS
            if (ns == 2) rc = launch_igemm_kernel(igemm_glds_kernel<T, BM, BN, AMODE, SWAP, 2, 2, 4>, 2 * stage, s, d, ntiles, splitk, 512);
            else if (ns == 3) rc = launch_igemm_kernel(igemm_glds_kernel<T, BM, BN, AMODE, SWAP, 3, 2, 4>, 3 * stage, s, d, ntiles, splitk, 512);
            else rc = launch_igemm_kernel(igemm_glds_kernel<T, BM, BN, AMODE, SWAP, 4, 2, 4>, 4 * stage, s, d, ntiles, splitk, 512);
            done = true;
        } else if (nw == 16) {   // experiment: 16 waves (4x4), 32x32 per wave
            rc = launch_igemm_kernel(igemm_glds_kernel<T, BM, BN, AMODE, SWAP, 2, 4, 4>, 2 * stage, s, d, ntiles, splitk, 1024);
            done = true;
        }
    }
    if constexpr (BM == 128 && BN == 64) {
        if (nw >= 8) {           // 8 waves (4x2), 32x32 per wave
            rc = launch_igemm_kernel(igemm_glds_kernel<T, BM, BN, AMODE, SWAP, 2, 4, 2>, 2 * stage, s, d, ntiles, splitk, 512);
            done = true;
        }
    }
    if (!done) {
        if (ns == 1) rc = launch_igemm_kernel(igemm_kernel<T, BM, BN, AMODE, SWAP>, 2 * stage, s, d, ntiles, splitk);
        else if (ns == 3) rc = launch_igemm_kernel(igemm_glds_kernel<T, BM, BN, AMODE, SWAP, 3>, 3 * stage, s, d, ntiles, splitk);
        else if (ns >= 4) rc = launch_igemm_kernel(igemm_glds_kernel<T, BM, BN, AMODE, SWAP, 4>, 4 * stage, s, d, ntiles, splitk);
        else rc = launch_igemm_kernel(igemm_glds_kernel<T, BM, BN, AMODE, SWAP, 2>, 2 * stage, s, d, ntiles, splitk);
    }
    if (rc || splitk == 1) return rc;
    const long nq = (long)d.M * (d.N / 4);
    LAUNCH(igemm_splitk_reduce_kernel<T>, dim3(grid_for(nq)), dim3(256), 0, s, d, splitk);
    return check_launch("igemm_splitk_reduce");
}
static void igemm_plan_for(int dtype, const ffn_igemm_desc& d, int* bm, int* bn, int* splitk);
template <typename T, int AMODE, bool SWAP>
static int dispatch_igemm_tile(hipStream_t s, const ffn_igemm_desc& d) {
    // pick the largest tile that still gives the chip >= ~1 wave of workgroups (256 CUs, 2 workgroups/CU)
    int bm, bn, sk;
    igemm_plan_for(sizeof(T) == 4 ? FFN_F32 : FFN_BF16, d, &bm, &bn, &sk);
    if (bm == 128 && bn == 128) return launch_igemm<T, 128, 128, AMODE, SWAP>(s, d, sk);
    if (bm == 128) return launch_igemm<T, 128, 64, AMODE, SWAP>(s, d, sk);
    return launch_igemm<T, 64, 64, AMODE, SWAP>(s, d, sk);
}
static bool can_split(const ffn_igemm_desc& d) {
    return d.ws && d.splitk != 1 && !(d.flags & (FFN_IG_GEGLU | FFN_IG_OUT_TRANSPOSED | FFN_IG_OUT_KV64));
}
// tile + number of K slices.  Without split-K small-M problems take small tiles to fill the chip; with it they keep the
// 128x128 tile (operand reuse) and the K loop is cut so that ~2 workgroups per CU exist.
static void igemm_plan_for(int dtype, const ffn_igemm_desc& d, int* bm, int* bn, int* splitk) {
    const long t128 = (long)((d.M + 127) / 128) * ((d.N + 127) / 128);
    const long t12864 = (long)((d.M + 127) / 128) * ((d.N + 63) / 64);
    *splitk = 1;
    static const char* tile_env = getenv("FFN_IGEMM_TILE");   // experiment: force 128x64 / 64x64 for dense problems
    if (tile_env && !d.conv && !(d.flags & FFN_IG_GEGLU)) {
        *bm = atoi(tile_env) >= 128 ? 128 : 64;
        *bn = 64;
    } else if (!d.conv && !(d.flags & FFN_IG_GEGLU) && d.M >= 128 && (!can_split(d) || t12864 >= 128)) {
        *bm = 128; *bn = 64;      // dense Linear layers (short K, memory/latency bound): measured 5-25% faster than 128x128
    } else if (d.N > 64 && (t128 >= 256 || (can_split(d) && d.M > 512 && d.N >= 128))) {
        *bm = 128; *bn = 128;     // measured (tools/bench_kernels.py): from M = 1024 up, 128x128 + split-K beats 128x64 without it
    } else if (t12864 >= 256 || d.M >= 4096 || (can_split(d) && d.M >= 96 && d.N >= 64)) {
        *bm = 128; *bn = 64;      // M <= 512: half the K slices (and slab traffic) of 128x128 for the same number of workgroups
    } else { *bm = 64; *bn = 64; }
    if (can_split(d)) {
        const int kstage = dtype == FFN_F32 ? 32 : 64;
        const int nk = (d.K + kstage - 1) / kstage;
        const long tiles = (long)((d.M + *bm - 1) / *bm) * ((d.N + *bn - 1) / *bn);
        int s = d.splitk > 1 ? d.splitk : (tiles >= 224 ? 1 : (int)((448 + tiles - 1) / tiles));
        if (d.splitk <= 1 && s > nk / 6) s = nk / 6;            // keep >= 6 K stages per slice
        const long per = (long)d.M * d.N * 4;
        if ((long)s * per > d.ws_bytes) s = (int)(d.ws_bytes / per);
        if (s > nk) s = nk;
        if (s < 1) s = 1;
        *splitk = s;
    }
}
static void igemm_tile_for(const ffn_igemm_desc& d, int* bm, int* bn) {
    int s;
    igemm_plan_for(FFN_BF16, d, bm, bn, &s);
}
extern "C" int ffn_igemm_variant(const ffn_igemm_desc* d, int* bm, int* bn) {
    REQUIRE(d && bm && bn, "igemm_variant: null argument");
    igemm_tile_for(*d, bm, bn);
    return FFN_OK;
}
extern "C" int ffn_attn_variant(int dtype, int D, int* dp, int* qf) {
    REQUIRE(dp && qf, "attn_variant: null argument");
    if (dtype == FFN_F32) {
        if (D <= 48) { *dp = 48; *qf = 2; } else if (D <= 64) { *dp = 64; *qf = 2; } else if (D <= 80) { *dp = 80; *qf = 2; } else { *dp = 160; *qf = 1; }
    } else {
        if (D <= 64) { *dp = 64; *qf = 2; } else if (D <= 96) { *dp = 96; *qf = 2; } else { *dp = 160; *qf = 1; }
    }
    return FFN_OK;
}


// ---- bf16 tile configurations and first-use autotuning ------------------------------------------------------------------
// The SD shapes span M = 4 ... 4M rows and N = 4 ... 10240 columns; which (tile, K-split) wins depends on how the tile count
// quantises over 256 CUs as much as on the tile's own efficiency (measured, tools/bench_kernels.py: at M = 98304 the 128x320 tile
// runs the N = 320 convs at 850-1000 TFLOP/s where 128x128 -- 17% of its third column tile wasted -- gives 690-760; 256x256
// reaches 1190 on N = 1280 but loses 20% on N = 640).  So the first time a problem shape is seen outside stream capture, the
// few plausible configurations are timed on the caller's stream with the caller's buffers and the winner is cached.
enum { CFG_64x64, CFG_128x64, CFG_128x128_8, CFG_128x128_16, CFG_256x128, CFG_256x256, CFG_128x320, CFG_128x160, CFG_192x320, CFG_H_128x320, CFG_H_256x128, CFG_H_256x256, CFG_H_128x128, CFG_PP_256x320, CFG_PP_256x256, CFG_PP_192x320, CFG_PP_192x256, CFG_PP_256x128, CFG_COUNT };
struct IgCfgInfo { int bm, bn, nwm, nwn; };
static const IgCfgInfo kCfg[CFG_COUNT] = {{64, 64, 2, 2}, {128, 64, 4, 2}, {128, 128, 2, 4}, {128, 128, 4, 4},
                                          {256, 128, 4, 4}, {256, 256, 4, 4}, {128, 320, 4, 4}, {128, 160, 4, 2}, {192, 320, 3, 4},
                                          {128, 320, 4, 4}, {256, 128, 4, 4}, {256, 256, 4, 4}, {128, 128, 4, 4},    // halo kernel (3x3 stride-1 convs)
                                          {256, 320, 2, 4}, {256, 256, 2, 4}, {192, 320, 2, 4}, {192, 256, 2, 4},      // ping-pong kernel (igemm_p8.h)
                                          {256, 128, 2, 4}};    // its 128-column tile: split-bf16 3x3 convolutions to N % 128 == 0 channels (the VAE's 128-channel layers)
struct IgChoice { int cfg, splitk; };
struct TunedEntry { IgChoice ch; bool validated; };   // imported entries (file / another rank) are checked against the problem at first use

static bool is_halo_cfg(int cfg) { return cfg == CFG_H_128x320 || cfg == CFG_H_256x128 || cfg == CFG_H_256x256 || cfg == CFG_H_128x128; }
static bool is_pp_cfg(int cfg) { return cfg >= CFG_PP_256x320 && cfg <= CFG_PP_256x128; }
// whether the ping-pong kernel (igemm_p8.h) handles this problem on a bm x bn tile: its restrictions are listed in that header
static bool pp_ok(const ffn_igemm_desc& d, int bm, int bn, int splitk = 1) {
    const long lim = (1l << 31) - 4096;
    // K tiles of the kernel: 64 bf16 elements; split-bf16 (d.K = the virtual 3 K): one 128-byte block [hi(32) | lo(32)] = 32 real elements
    const int nkt = d.x3 ? d.K / 96 : d.K / 64;
    // the 128-column tile exists for unsplit split-bf16 3x3 convolutions with a plain / residual epilogue only
    // (and only where neither wider tile divides N: the UNet's N = 640 / 1280 convolutions keep their 320- / 256-column tiles, no extra tuner candidate there)
    if (bn == 128 && (bm != 256 || !d.x3 || d.conv != 1 || splitk > 1 || d.N % 256 == 0 || d.N % 320 == 0 || (d.flags & (FFN_IG_GEGLU | FFN_IG_OUT_PAIR | FFN_IG_OUT_KV64 | FFN_IG_OUT_TRANSPOSED)))) return false;
    if (d.x3) {
        if (d.x3 != 2 || d.K % 96 != 0 || d.a_lo != 32) return false;             // blocked operands only
    } else if (d.K % 64 != 0) return false;
    if (nkt < 2 || d.N % bn != 0 || d.M < bm) return false;
    const int osz = d.x3 ? 4 : 2;                                  // bytes per output / residual element
    if (splitk > 1) {       // split-K: raw fp32 slabs + igemm_splitk_reduce_kernel (which applies every plain epilogue option); slices of whole K tiles
        if (!can_split(d) || nkt % splitk != 0 || nkt / splitk < 2 || (long)splitk * d.M * d.N * 4 > d.ws_bytes || (long)splitk * d.M * d.N * 4 >= lim) return false;
    } else {
        // GELU / RELU: applied by the plain bf16 epilogue, whose accumulators start at the bias -- not beside a residual (it starts there too)
        const int act_ok = (!d.residual && !d.x3 && !d.f8 && !(d.flags & FFN_IG_GEGLU)) ? (FFN_IG_OUT_GELU | FFN_IG_OUT_RELU) : 0;
        if ((d.alpha != 1.0f && !d.f8) || (d.flags & ~(FFN_IG_GEGLU | act_ok | (d.x3 ? (FFN_IG_OUT_F32 | FFN_IG_OUT_PAIR | FFN_IG_OUT_KV64) : 0)))) return false;
        if (d.f8 && (!d.conv || (d.flags & FFN_IG_GEGLU))) return false;
        if (d.x3 && (d.flags & FFN_IG_GEGLU) && !(d.flags & FFN_IG_OUT_PAIR)) return false;      // the split-bf16 GEGLU tile writes the pair form only
        if ((d.flags & FFN_IG_OUT_PAIR) && !(d.flags & FFN_IG_GEGLU) && (d.N % 32 != 0 || (d.ldo / 2) % 32 != 0 || (d.flags & FFN_IG_OUT_TRANSPOSED))) return false;      // plain tile: blocked pair rows only
        if ((d.flags & FFN_IG_GEGLU) && bn != 256) return false;
        if (d.rowbias && d.rows_per_batch < 128) return false;      // a wave's rows (bm / 2) may straddle two images, not three
    }
    long a_bytes;
    if (d.conv) {
        const int pix = d.x3 ? d.lda : d.Cin;                      // elements per input pixel
        const int cpt = d.x3 ? d.Cin / 32 : d.Cin / 64;            // K tiles per tap
        if (d.Cin % (d.x3 ? 32 : 64) != 0 || d.K != (d.conv == 2 ? 4 : 9) * (d.x3 ? 3 : 1) * d.Cin) return false;
        if (d.x3 && d.lda != 2 * d.Cin) return false;
        if (d.conv == 2 && (d.f8 || d.stride != 1 || d.upsample)) return false;
        if ((long)cpt * 9 * cpt >= 65536) return false;            // exactness range of the tap reciprocal
        a_bytes = (long)(d.M / (d.Hout * d.Wout)) * d.Hin * d.Win * pix * 2;
        if (2 * d.Hin + 2 >= 32768 || 2 * d.Win + 2 >= 32768) return false;
        if (a_bytes + 256l * pix * 2 >= lim) return false;
    } else {
        a_bytes = (long)d.M * d.lda * 2;
        if (a_bytes + 256l * d.lda * 2 >= lim) return false;
    }
    if ((long)d.N * d.Kpad * 2 >= lim || (long)(d.M + 256) * d.ldo * ((d.flags & FFN_IG_OUT_PAIR) ? 2 : osz) >= lim) return false;
    if (d.residual && (long)(d.M + 256) * d.ldr * osz >= lim) return false;
    return true;
}
// LDS bytes of ONE halo buffer of the halo conv kernel for tile height bm, or 0 if the problem does not fit the kernel: 3x3,
// stride 1, pad 1, no upsample, whole 64-channel chunks, and a tile = whole image rows (W <= bm) or a piece of one row
static int halo_bytes_for(const ffn_igemm_desc& d, int bm) {
    if (!d.conv || d.stride != 1 || d.upsample || d.pad != 1 || d.Hin != d.Hout || d.Win != d.Wout) return 0;
    if (d.Cin % 64 != 0 || d.Cin > 30000 || d.K != 9 * d.Cin || d.M % bm != 0) return 0;
    const int H = d.Hin, W = d.Win;
    int tw, tr;
    if (W <= bm) {
        if (bm % W != 0 || (H * W) % bm != 0) return 0;
        tw = W; tr = bm / W;
    } else {
        if (W % bm != 0) return 0;
        tw = bm; tr = 1;
    }
    const int nq = ((tr + 2) * (tw + 2) + 7) / 8;
    if (nq > 4 * 16) return 0;                      // 4 halo wave-instructions per wave, 16 waves
    return nq * 8 * 128;
}
// X3 (split-bf16, FFN_BF16X3) problems run the generic 64x64 / 128x64 / 128x128 tiles (recomputing loader) and the ping-pong tiles
static bool x3_cfg(int cfg) { return cfg == CFG_64x64 || cfg == CFG_128x64 || cfg == CFG_128x128_8 || is_pp_cfg(cfg); }
template <int AMODE, bool X3 = false, bool F8 = false>
static int launch_bf16_cfg(hipStream_t s, const ffn_igemm_desc& d, IgChoice ch) {
    const IgCfgInfo& c = kCfg[ch.cfg];
    if ((X3 || F8) && !x3_cfg(ch.cfg)) return fail(FFN_EINVAL, "igemm: configuration %d is not built for split-bf16 / fp8 problems", ch.cfg);
    const int ntiles = ((d.M + c.bm - 1) / c.bm) * ((d.N + c.bn - 1) / c.bn);
    const int lds = 2 * (c.bm + c.bn) * 128, threads = 64 * c.nwm * c.nwn;
    int rc = FFN_OK;
    // FASTK kernels (streaming loader) need every 128-byte K stage inside K / inside one conv tap
    // the streaming loader's zero-page pointers (N-tail columns, rows past M, conv padding) walk 128 B per K stage over the WHOLE K
    // range of the launch: K * 2 bytes must stay inside the 64 KiB zero page
    const bool fastk = (d.conv ? d.Cin % 64 == 0 : d.K % 64 == 0) && (long)d.K * 2 + 256 <= (long)sizeof(g_zero_page);
#define FFN_CFG_CASE(ID, BM_, BN_, WM_, WN_)                                                                                               \
    case ID:                                                                                                                        \
        if constexpr (X3) {                                                                                                         \
            if constexpr (ID == CFG_64x64 || ID == CFG_128x64 || ID == CFG_128x128_8)                                               \
                rc = launch_igemm_kernel(igemm_glds_kernel<bf16, BM_, BN_, AMODE, true, 2, WM_, WN_, false, true>, lds, s, d, ntiles, ch.splitk, threads, true); \
        } else if constexpr (F8) {                                                                                                  \
            if constexpr (AMODE == AMODE_CONV3 && (ID == CFG_64x64 || ID == CFG_128x64 || ID == CFG_128x128_8))                     \
                rc = launch_igemm_kernel(igemm_glds_kernel<bf16, BM_, BN_, AMODE, true, 2, WM_, WN_, false, false, true>, lds, s, d, ntiles, ch.splitk, threads, true); \
        } else {                                                                                                                    \
            rc = fastk ? launch_igemm_kernel(igemm_glds_kernel<bf16, BM_, BN_, AMODE, true, 2, WM_, WN_, true>, lds, s, d, ntiles, ch.splitk, threads, true)  \
                       : launch_igemm_kernel(igemm_glds_kernel<bf16, BM_, BN_, AMODE, true, 2, WM_, WN_, false>, lds, s, d, ntiles, ch.splitk, threads, true); \
        }                                                                                                                           \
        break;
    switch (ch.cfg) {
        FFN_CFG_CASE(CFG_64x64, 64, 64, 2, 2)
        FFN_CFG_CASE(CFG_128x64, 128, 64, 4, 2)
        FFN_CFG_CASE(CFG_128x128_8, 128, 128, 2, 4)
        FFN_CFG_CASE(CFG_128x128_16, 128, 128, 4, 4)
        FFN_CFG_CASE(CFG_256x128, 256, 128, 4, 4)
        FFN_CFG_CASE(CFG_256x256, 256, 256, 4, 4)
        FFN_CFG_CASE(CFG_128x320, 128, 320, 4, 4)
        FFN_CFG_CASE(CFG_128x160, 128, 160, 4, 2)
        FFN_CFG_CASE(CFG_192x320, 192, 320, 3, 4)      // 12 waves: 168 registers per wave (spills at 16 waves x 128)
        case CFG_PP_256x320:
        case CFG_PP_256x256:
        case CFG_PP_192x320:
        case CFG_PP_192x256:
        case CFG_PP_256x128: {
            if (!pp_ok(d, c.bm, c.bn, ch.splitk)) return fail(FFN_EINVAL, "igemm: ping-pong kernel not applicable");
            const int pplds = 2 * (c.bm + c.bn) * 128 + 12288;
            const int nt = ((d.M + c.bm - 1) / c.bm) * (d.N / c.bn) * ch.splitk;
            const int grid = nt < device_cus() ? nt : device_cus();
            (void)hipGetLastError();
#define FFN_PP_LAUNCH(BM_, BN_, RES_, GEGLU_)                                              \
    do {                                                                                   \
        auto kern = igemm_pp_kernel<BM_, BN_, AMODE, RES_, (GEGLU_) && !F8, false, false, X3, F8>;      \
        if ((rc = set_lds(kern, pplds))) return rc;                                        \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), pplds, s, d, 1);                   \
    } while (0)
#define FFN_PP_TILE(BM_)                                                                   \
    do {                                                                                   \
        if (c.bn == 320) {                                                                 \
            if (d.residual) FFN_PP_LAUNCH(BM_, 320, true, false);                          \
            else FFN_PP_LAUNCH(BM_, 320, false, false);                                    \
        } else {                                                                           \
            if constexpr (AMODE == AMODE_DENSE) {                                          \
                if (d.flags & FFN_IG_GEGLU) FFN_PP_LAUNCH(BM_, 256, false, true);          \
                else if (d.residual) FFN_PP_LAUNCH(BM_, 256, true, false);                 \
                else FFN_PP_LAUNCH(BM_, 256, false, false);                                \
            } else {                                                                       \
                if (d.residual) FFN_PP_LAUNCH(BM_, 256, true, false);                      \
                else FFN_PP_LAUNCH(BM_, 256, false, false);                                \
            }                                                                              \
        }                                                                                  \
    } while (0)
#define FFN_PP_SPLIT(BM_, BN_)                                                             \
    do {                                                                                   \
        auto kern = igemm_pp_kernel<BM_, BN_, AMODE, false, false, true, false, X3, F8>;   \
        if ((rc = set_lds(kern, pplds))) return rc;                                        \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), pplds, s, d, ch.splitk);           \
    } while (0)
            if (c.bn == 128) {                 // (pp_ok: split-bf16 3x3 convolution, unsplit)
                if constexpr (X3 && AMODE == AMODE_CONV3) {
                    if (d.residual) FFN_PP_LAUNCH(256, 128, true, false);
                    else FFN_PP_LAUNCH(256, 128, false, false);
                    return check_launch("igemm(ping-pong, 256 x 128)");
                } else {
                    return fail(FFN_EINVAL, "igemm: the 256 x 128 ping-pong tile is built for split-bf16 3x3 convolutions");
                }
            }
            if (ch.splitk > 1) {
                if (c.bm == 256 && c.bn == 320) FFN_PP_SPLIT(256, 320);
                else if (c.bm == 256) FFN_PP_SPLIT(256, 256);
                else if (c.bn == 320) FFN_PP_SPLIT(192, 320);
                else FFN_PP_SPLIT(192, 256);
                if ((rc = check_launch("igemm(ping-pong, split-K)"))) return rc;
                const long nq = (long)d.M * (d.N / 4);
                if constexpr (X3) LAUNCH(igemm_splitk_reduce_kernel<float>, dim3(grid_for(nq)), dim3(256), 0, s, d, ch.splitk);
                else LAUNCH(igemm_splitk_reduce_kernel<bf16>, dim3(grid_for(nq)), dim3(256), 0, s, d, ch.splitk);
                return check_launch("igemm_splitk_reduce");
            }
            if (c.bm == 256) FFN_PP_TILE(256);
            else FFN_PP_TILE(192);
#undef FFN_PP_SPLIT
#undef FFN_PP_TILE
#undef FFN_PP_LAUNCH
            return check_launch("igemm(ping-pong)");
        }
        case CFG_H_128x320:
        case CFG_H_256x128:
        case CFG_H_256x256:
        case CFG_H_128x128:
            if constexpr (AMODE == AMODE_CONV3 && !X3 && !F8) {
                const int hb = halo_bytes_for(d, c.bm);
                if (hb <= 0 || ch.splitk != 1) return fail(FFN_EINVAL, "igemm: halo kernel not applicable");
                const int hlds = 2 * hb + 2 * c.bn * 128;
                const int nt = (d.M / c.bm) * ((d.N + c.bn - 1) / c.bn);
                (void)hipGetLastError();
                if (ch.cfg == CFG_H_128x320) {
                    auto kern = igemm_halo_kernel<bf16, 128, 320, 4, 4>;
                    if ((rc = set_lds(kern, hlds))) return rc;
                    hipLaunchKernelGGL(kern, dim3(nt), dim3(threads), hlds, s, d, hb);
                } else if (ch.cfg == CFG_H_256x128) {
                    auto kern = igemm_halo_kernel<bf16, 256, 128, 4, 4>;
                    if ((rc = set_lds(kern, hlds))) return rc;
                    hipLaunchKernelGGL(kern, dim3(nt), dim3(threads), hlds, s, d, hb);
                } else if (ch.cfg == CFG_H_256x256) {
                    auto kern = igemm_halo_kernel<bf16, 256, 256, 4, 4>;
                    if ((rc = set_lds(kern, hlds))) return rc;
                    hipLaunchKernelGGL(kern, dim3(nt), dim3(threads), hlds, s, d, hb);
                } else {
                    auto kern = igemm_halo_kernel<bf16, 128, 128, 4, 4>;
                    if ((rc = set_lds(kern, hlds))) return rc;
                    hipLaunchKernelGGL(kern, dim3(nt), dim3(threads), hlds, s, d, hb);
                }
                return check_launch("igemm(halo)");
            } else {
                return fail(FFN_EINVAL, "igemm: halo configurations are for 3x3 convolutions");
            }
        default: return fail(FFN_EINVAL, "igemm: bad configuration %d", ch.cfg);
    }
#undef FFN_CFG_CASE
    if (rc || ch.splitk == 1) return rc;
    const long nq = (long)d.M * (d.N / 4);
    if constexpr (X3) LAUNCH(igemm_splitk_reduce_kernel<float>, dim3(grid_for(nq)), dim3(256), 0, s, d, ch.splitk);
    else LAUNCH(igemm_splitk_reduce_kernel<bf16>, dim3(grid_for(nq)), dim3(256), 0, s, d, ch.splitk);
    return check_launch("igemm_splitk_reduce");
}

struct TuneKey {
    int M, N, K, conv, Cin, Hin, Win, stride, upsample, flags, lda, splitk, ptrs, wslabs, rpb, alpha1, ldo, pad, Hout, Wout, Kpad, ldr, dtype;
    bool operator==(const TuneKey& o) const { return memcmp(this, &o, sizeof(TuneKey)) == 0; }
};
struct TuneKeyHash {
    size_t operator()(const TuneKey& k) const {
        const int* p = reinterpret_cast<const int*>(&k);
        size_t h = 1469598103934665603ull;
        for (size_t i = 0; i < sizeof(TuneKey) / sizeof(int); ++i) h = (h ^ (size_t)(unsigned)p[i]) * 1099511628211ull;
        return h;
    }
};
static TuneKey tune_key(const ffn_igemm_desc& d) {
    TuneKey k;
    memset(&k, 0, sizeof(k));
    k.M = d.M; k.N = d.N; k.K = d.K; k.conv = d.conv; k.flags = d.flags; k.lda = d.lda; k.splitk = d.splitk;
    if (d.conv) { k.Cin = d.Cin; k.Hin = d.Hin; k.Win = d.Win; k.stride = d.stride; k.upsample = d.upsample; k.pad = d.pad; k.Hout = d.Hout; k.Wout = d.Wout; }
    k.Kpad = d.Kpad;
    k.ldr = d.residual ? d.ldr : 0;
    k.dtype = d.x3 ? (FFN_BF16X3 | (d.x3 << 8)) : (d.f8 ? FFN_FP8 : FFN_BF16);
    k.ptrs = (d.bias ? 1 : 0) | (d.rowbias ? 2 : 0) | (d.residual ? 4 : 0) | (d.ws && d.ws_bytes > 0 ? 8 : 0);
    const long per = (long)d.M * d.N * 4;
    k.wslabs = d.ws ? (int)(d.ws_bytes / per > 1024 ? 1024 : d.ws_bytes / per) : 0;      // how many split-K slabs the workspace holds
    k.rpb = d.rows_per_batch;
    k.alpha1 = d.alpha == 1.0f;
    k.ldo = d.ldo;
    return k;
}
static std::mutex g_tune_mu;
static std::unordered_map<TuneKey, TunedEntry, TuneKeyHash> g_tuned;
static int g_tune_runtime = 1;      // ffn_igemm_tune_enable(0): no more timing-based tuning in this process (after a table sync)
static bool tune_enabled() {
    static const bool on = [] { const char* e = getenv("FFN_IGEMM_TUNE"); return !(e && atoi(e) == 0); }();
    return on && g_tune_runtime;
}
// the deterministic rule-based choice (also what f32 parity mode uses): igemm_plan_for + igemm_exec_cfg mapped to a configuration
static IgChoice heuristic_choice(const ffn_igemm_desc& d) {
    if (d.conv == 2) {      // 2x2 sub-pixel convolutions exist in the ping-pong kernel only (ffn_igemm has checked that one of its tiles applies)
        for (int bn : {320, 256})
            for (int h : {256, 192})
                if (d.N % bn == 0 && pp_ok(d, h, bn)) return IgChoice{bn == 320 ? (h == 256 ? CFG_PP_256x320 : CFG_PP_192x320) : (h == 256 ? CFG_PP_256x256 : CFG_PP_192x256), 1};
    }
    // the ping-pong tile first, where its unsplit form applies and its tiles fill at least 3/4 of the chip: tile height = the one whose
    // tile count wastes the least of the last round of workgroups (the rule pp_trans_tile uses).  Without this, every launch that
    // cannot be tuned (FFN_IGEMM_TUNE=0, stream capture before a shape was seen, out aliasing residual) fell back to the 2-stage kernels
    if (d.splitk <= 1) {
        for (int bn : {320, 256}) {
            if (d.N % bn != 0) continue;
            long best = -1;
            int bh = 0;
            for (int h : {256, 192}) {
                if (!pp_ok(d, h, bn)) continue;
                const long tiles = (long)((d.M + h - 1) / h) * (d.N / bn);
                if (tiles * 4 < (long)device_cus() * 3) continue;
                const long cost = ((tiles + device_cus() - 1) / device_cus()) * h;
                if (best < 0 || cost < best) { best = cost; bh = h; }
            }
            if (best >= 0) return IgChoice{bn == 320 ? (bh == 256 ? CFG_PP_256x320 : CFG_PP_192x320) : (bh == 256 ? CFG_PP_256x256 : CFG_PP_192x256), 1};
        }
        if (d.N % 128 == 0 && pp_ok(d, 256, 128) && (long)((d.M + 255) / 256) * (d.N / 128) * 4 >= (long)device_cus() * 3) return IgChoice{CFG_PP_256x128, 1};
    }
    int bm, bn, sk, ns, nw;
    igemm_plan_for(FFN_BF16, d, &bm, &bn, &sk);
    igemm_exec_cfg(FFN_BF16, d, bm, bn, sk, &ns, &nw);
    int cfg = CFG_64x64;
    if (bm == 128 && bn == 64) cfg = CFG_128x64;
    if (bm == 128 && bn == 128) cfg = (nw == 16 && !d.x3 && !d.f8) ? CFG_128x128_16 : CFG_128x128_8;
    return IgChoice{cfg, sk};
}
static int candidates_for(const ffn_igemm_desc& d, IgChoice* out, int cap) {
    int n = 0;
    const IgChoice h = heuristic_choice(d);
    out[n++] = h;
    const int nk = (d.K + 63) / 64;
    const long per = (long)d.M * d.N * 4;
    for (int cfg = 0; cfg < CFG_COUNT; ++cfg) {
        const IgCfgInfo& c = kCfg[cfg];
        if ((d.x3 || d.f8) && !x3_cfg(cfg)) continue;
        if (d.conv == 2 && !is_pp_cfg(cfg)) continue;
        if ((cfg == CFG_128x320 || cfg == CFG_128x160 || cfg == CFG_192x320) && (d.flags & FFN_IG_GEGLU)) continue;   // odd number of column blocks per wave
        if (is_halo_cfg(cfg)) {
            const int hb = halo_bytes_for(d, c.bm);
            if (hb <= 0 || d.splitk > 1 || 2 * hb + 2 * c.bn * 128 > 160 * 1024) continue;
            if ((cfg == CFG_H_128x320) && (d.flags & FFN_IG_GEGLU)) continue;
        }
        const bool halo = is_halo_cfg(cfg) || is_pp_cfg(cfg);                      // no generic split-K variants
        if (is_pp_cfg(cfg)) {
            // ping-pong: unsplit where its epilogue applies; split-K (a divisor of the K-tile count, >= 2 K tiles per slice) where the
            // output tiles alone leave most of the chip idle
            const long pt = (long)((d.M + c.bm - 1) / c.bm) * (d.N / (c.bn > 0 ? c.bn : 1));
            if (d.splitk <= 1 && pp_ok(d, c.bm, c.bn) && n < cap) out[n++] = IgChoice{cfg, 1};
            static const int split_below = [] { const char* e = getenv("FFN_PP_SPLIT_BELOW"); return e ? atoi(e) : 160; }();      // unsplit tile counts below this get split-K candidates
            if (cfg != CFG_PP_256x128 && d.splitk != 1 && pt > 0 && pt < split_below && d.N % c.bn == 0) {
                const int nkt = d.x3 ? d.K / 96 : d.K / 64;
                int added = 0;
                for (int sgo = (int)((384 + pt - 1) / pt); sgo >= 2 && added < 2; --sgo) {
                    if (d.splitk > 1 && sgo != d.splitk) continue;
                    if (nkt % sgo != 0 || nkt / sgo < 2 || !pp_ok(d, c.bm, c.bn, sgo)) continue;
                    if (n < cap) out[n++] = IgChoice{cfg, sgo};
                    ++added;
                }
            }
            continue;
        }
        if (c.bm > 64 && c.bm >= 2 * d.M) continue;                                // tile mostly empty
        if (c.bn > 64 && c.bn >= 2 * d.N) continue;
        if (cfg == CFG_64x64 && (long)d.M * d.N > (1l << 22)) continue;
        const long tiles = (long)((d.M + c.bm - 1) / c.bm) * ((d.N + c.bn - 1) / c.bn);
        int splits[3] = {1, 0, 0};
        if (d.splitk > 1) {              // caller-forced split: only where a split launch is legal and the workspace holds the slabs
            int sf = (can_split(d) && !halo) ? d.splitk : 1;
            if ((long)sf * per > d.ws_bytes) sf = (int)(d.ws_bytes / per);
            if (sf > nk) sf = nk;
            splits[0] = sf >= 2 ? sf : 1;
        } else if (can_split(d) && !halo) {
            for (int t = 0; t < 2; ++t) {
                int sgo = (int)(((t ? 512 : 256) + tiles / 2) / tiles);
                if (sgo > nk / 4) sgo = nk / 4;
                if ((long)sgo * per > d.ws_bytes) sgo = (int)(d.ws_bytes / per);
                if (sgo >= 2) splits[1 + t] = sgo;
            }
            if (splits[2] == splits[1]) splits[2] = 0;
        }
        for (int t = 0; t < 3 && n < cap; ++t) {
            if (!splits[t]) continue;
            bool dup = false;
            for (int j = 0; j < n; ++j) dup |= out[j].cfg == cfg && out[j].splitk == splits[t];
            if (!dup) out[n++] = IgChoice{cfg, splits[t]};
        }
    }
    return n;
}
static int g_force_cfg = -1;     // testing hook (ffn_igemm_force_config): run every bf16 problem on this configuration where it is valid
extern "C" int ffn_igemm_num_configs(void) { return CFG_COUNT; }
extern "C" int ffn_igemm_force_config(int cfg) {
    const int prev = g_force_cfg;
    g_force_cfg = (cfg >= 0 && cfg < CFG_COUNT) ? cfg : -1;
    return prev;
}
template <int AMODE, bool X3 = false, bool F8 = false>
static int tuned_bf16(hipStream_t s, const ffn_igemm_desc& d) {
    if (g_force_cfg >= 0) {
        IgChoice cand[40];
        const int nc = candidates_for(d, cand, 40);
        for (int i = 0; i < nc; ++i)
            if (cand[i].cfg == g_force_cfg) return launch_bf16_cfg<AMODE, X3, F8>(s, d, cand[i]);
        return launch_bf16_cfg<AMODE, X3, F8>(s, d, heuristic_choice(d));     // not valid for this problem
    }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap);
    const bool aliased = d.residual == d.out;        // repeated launches would accumulate: never time such a call
    const TuneKey key = tune_key(d);
    {
        std::lock_guard<std::mutex> lk(g_tune_mu);
        auto it = g_tuned.find(key);
        if (it != g_tuned.end()) {
            if (!it->second.validated) {
                // an entry that came in as data (a tune file, another rank's table): launch it only if this build would have
                // offered exactly that (configuration, K split) for THIS problem -- workspace capacity, split legality, tile
                // applicability are all decided in candidates_for; anything else is dropped and the problem is tuned afresh
                IgChoice cand[40];
                const int nc = candidates_for(d, cand, 40);
                bool ok = false;
                for (int i = 0; i < nc; ++i) ok |= cand[i].cfg == it->second.ch.cfg && cand[i].splitk == it->second.ch.splitk;
                if (ok) it->second.validated = true;
                else g_tuned.erase(it), it = g_tuned.end();
            }
            if (it != g_tuned.end()) return launch_bf16_cfg<AMODE, X3, F8>(s, d, it->second.ch);
        }
    }
    if (!tune_enabled() || cap != hipStreamCaptureStatusNone || aliased) return launch_bf16_cfg<AMODE, X3, F8>(s, d, heuristic_choice(d));
    std::lock_guard<std::mutex> lk(g_tune_mu);       // one tuning at a time
    IgChoice cand[40];
    const int nc = candidates_for(d, cand, 40);
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return launch_bf16_cfg<AMODE, X3, F8>(s, d, cand[0]);
    IgChoice best = cand[0];
    float best_ms = 1e30f;
    const int reps = 3;
    for (int i = 0; i < nc; ++i) {
        int rc = launch_bf16_cfg<AMODE, X3, F8>(s, d, cand[i]);      // warm (LDS opt-in, code load)
        if (rc) continue;
        float ms = 1e30f;
        for (int round = 0; round < 2 && !rc; ++round) {       // min of two timed groups: one noisy group must not pick the configuration
            (void)hipEventRecord(e0, s);
            for (int r = 0; r < reps && !rc; ++r) rc = launch_bf16_cfg<AMODE, X3, F8>(s, d, cand[i]);
            (void)hipEventRecord(e1, s);
            if (rc || hipEventSynchronize(e1) != hipSuccess) { rc = rc ? rc : FFN_EHIP; break; }
            float t = 0.f;
            (void)hipEventElapsedTime(&t, e0, e1);
            if (t < ms) ms = t;
        }
        if (rc) continue;
        if (ms < best_ms) { best_ms = ms; best = cand[i]; }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    g_tuned[key] = TunedEntry{best, true};
    static const bool verbose = getenv("FFN_IGEMM_TUNE_VERBOSE") != nullptr;
    if (verbose)
        fprintf(stderr, "[ffn tune] %s M=%d N=%d K=%d flags=%d -> %dx%d split %d (%.1f us, %d candidates)\n", d.conv ? "conv" : "dense", d.M, d.N,
                d.K, d.flags, kCfg[best.cfg].bm, kCfg[best.cfg].bn, best.splitk, best_ms * 1e3f / reps, nc);
    return launch_bf16_cfg<AMODE, X3, F8>(s, d, best);       // the output now holds the winner's result
}
// ---- the tuned table as data: export / import (persist it across processes, broadcast rank 0's table so that every rank of a
// sharded run launches the same configurations -- bf16 results then are bit-identical across ranks)
// entry = [stamp | TuneKey | cfg | splitk]; the stamp names the layout of this build's table (key size, configuration list, arch):
// entries written by a different build are ignored on import
static constexpr int kTuneEntryInts = (int)(sizeof(TuneKey) / sizeof(int)) + 3;
static constexpr int kTuneStamp = 0x67780000 ^ (950 << 4) ^ ((int)sizeof(TuneKey) << 8) ^ CFG_COUNT ^ (4 << 24);      // gfx950, table layout 4 (round 5: blocked split-bf16 operands)
extern "C" int ffn_igemm_tune_entry_ints(void) { return kTuneEntryInts; }
extern "C" int ffn_igemm_tune_stamp(void) { return kTuneStamp; }
extern "C" int ffn_igemm_tune_clear(void) {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    const int n = (int)g_tuned.size();
    g_tuned.clear();
    return n;
}
extern "C" int ffn_igemm_tune_enable(int on) {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    const int prev = g_tune_runtime;
    g_tune_runtime = on ? 1 : 0;
    return prev;
}
extern "C" int ffn_igemm_tune_export(int* buf, int max_entries) {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    int n = 0;
    for (const auto& kv : g_tuned) {
        if (buf && n < max_entries) {
            int* e = buf + (long)n * kTuneEntryInts;
            e[0] = kTuneStamp;
            memcpy(e + 1, &kv.first, sizeof(TuneKey));
            e[kTuneEntryInts - 2] = kv.second.ch.cfg;
            e[kTuneEntryInts - 1] = kv.second.ch.splitk;
        }
        ++n;
    }
    return n;       // number of entries in the table (may exceed max_entries: call again with a larger buffer)
}
extern "C" int ffn_igemm_tune_import(const int* buf, int n_entries) {
    REQUIRE(buf || n_entries == 0, "igemm_tune_import: null buffer");
    std::lock_guard<std::mutex> lk(g_tune_mu);
    int n = 0;
    for (int i = 0; i < n_entries; ++i) {
        const int* e = buf + (long)i * kTuneEntryInts;
        if (e[0] != kTuneStamp) continue;                                       // a table from another build / layout / arch
        TuneKey k;
        memcpy(&k, e + 1, sizeof(TuneKey));
        const IgChoice ch{e[kTuneEntryInts - 2], e[kTuneEntryInts - 1]};
        if (ch.cfg < 0 || ch.cfg >= CFG_COUNT || ch.splitk < 1) continue;
        g_tuned[k] = TunedEntry{ch, false};      // validated against the actual problem (candidates_for) at its first lookup
        ++n;
    }
    return n;
}
static bool tuned_lookup(const ffn_igemm_desc& d, IgChoice* ch) {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    auto it = g_tuned.find(tune_key(d));
    if (it == g_tuned.end()) return false;
    *ch = it->second.ch;
    return true;
}

static bool pp_trans_tile(const ffn_igemm_desc& d, int* bm, int* bn);
static ffn_igemm_desc x3_view(const ffn_igemm_desc& d);
static ffn_igemm_desc f8_view(const ffn_igemm_desc& d);
extern "C" int ffn_igemm_kernel_name(int dtype, const ffn_igemm_desc* d0, char* buf, int len) {
    REQUIRE(d0 && buf && len > 0, "igemm_kernel_name: null argument");
    int bm, bn, sk, ns, nw;
    IgChoice ch;
    const bool x3 = dtype == FFN_BF16X3, f8 = dtype == FFN_FP8;
    const ffn_igemm_desc dd = x3 ? x3_view(*d0) : (f8 ? f8_view(*d0) : *d0);
    const ffn_igemm_desc* d = &dd;
    if (x3 && (d->flags & FFN_IG_OUT_TRANSPOSED)) {
        if (!d->conv && pp_trans_tile(*d, &bm, &bn)) {
            snprintf(buf, len, "void igemm_pp_kernel<%d, %d, 0, false, false, false, true, true, false>(ffn_igemm_desc, int)", bm, bn);
            return FFN_OK;
        }
        igemm_plan_for(FFN_BF16, *d, &bm, &bn, &sk);
        snprintf(buf, len, "void igemm_glds_kernel<bf16, %d, %d, 0, false, 2, 2, 2, false, true, false>(ffn_igemm_desc)", bm, bn);
        return FFN_OK;
    }
    if ((dtype == FFN_BF16 || x3 || f8) && !(d->flags & FFN_IG_OUT_TRANSPOSED)) {      // the tuned (or, untuned, rule-based) bf16 configuration
        if (!tuned_lookup(*d, &ch)) ch = heuristic_choice(*d);
        const IgCfgInfo& c = kCfg[ch.cfg];
        if (is_halo_cfg(ch.cfg)) {
            snprintf(buf, len, "void igemm_halo_kernel<bf16, %d, %d, %d, %d>(ffn_igemm_desc, int)", c.bm, c.bn, c.nwm, c.nwn);
            return FFN_OK;
        }
        if (is_pp_cfg(ch.cfg)) {
            const bool split = ch.splitk > 1;
            snprintf(buf, len, "void igemm_pp_kernel<%d, %d, %d, %s, %s, %s, false, %s, %s>(ffn_igemm_desc, int)", c.bm, c.bn, d->conv ? 1 : 0,
                     (!split && d->residual) ? "true" : "false", (!split && (d->flags & FFN_IG_GEGLU)) ? "true" : "false", split ? "true" : "false",
                     x3 ? "true" : "false", f8 ? "true" : "false");
            return FFN_OK;
        }
        const bool fastk = !x3 && !f8 && (d->conv ? d->Cin % 64 == 0 : d->K % 64 == 0) && (long)d->K * 2 + 256 <= (long)sizeof(g_zero_page);
        snprintf(buf, len, "void igemm_glds_kernel<bf16, %d, %d, %d, true, 2, %d, %d, %s, %s, %s>(ffn_igemm_desc)", c.bm, c.bn, d->conv ? 1 : 0, c.nwm, c.nwn,
                 fastk ? "true" : "false", x3 ? "true" : "false", f8 ? "true" : "false");
        return FFN_OK;
    }
    if (dtype == FFN_BF16 && (d->flags & FFN_IG_OUT_TRANSPOSED) && !d->conv && pp_trans_tile(*d, &bm, &bn)) {
        snprintf(buf, len, "void igemm_pp_kernel<%d, %d, 0, false, false, false, true, false, false>(ffn_igemm_desc, int)", bm, bn);
        return FFN_OK;
    }
    igemm_plan_for(dtype, *d, &bm, &bn, &sk);
    igemm_exec_cfg(dtype, *d, bm, bn, sk, &ns, &nw);
    const char* t = dtype == FFN_F32 ? "float" : "bf16";
    const char* swap = (d->flags & FFN_IG_OUT_TRANSPOSED) ? "false" : "true";
    if (ns == 1) snprintf(buf, len, "void igemm_kernel<%s, %d, %d, %d, %s>(ffn_igemm_desc)", t, bm, bn, d->conv ? 1 : 0, swap);
    else {
        const int nwm = nw == 16 ? 4 : (nw == 8 ? (bn == 64 ? 4 : 2) : 2), nwn = nw / nwm;
        snprintf(buf, len, "void igemm_glds_kernel<%s, %d, %d, %d, %s, %d, %d, %d, false, false, false>(ffn_igemm_desc)", t, bm, bn, d->conv ? 1 : 0, swap, ns, nwm, nwn);
    }
    return FFN_OK;
}
// transposed-output (V^T) launches on the ping-pong kernel: deterministic tile choice (no tuning): the tile height whose tile count
// wastes the least of the last round of 256 workgroups
static bool pp_trans_tile(const ffn_igemm_desc& d, int* bm, int* bn) {
    static const bool on = [] { const char* e = getenv("FFN_IGEMM_PP_TRANS"); return !(e && atoi(e) == 0); }();
    if (!on) return false;
    const long lim = (1l << 31) - 4096;
    *bn = d.N % 320 == 0 ? 320 : (d.N % 256 == 0 ? 256 : 0);
    if (!*bn || d.alpha != 1.0f || d.rows_per_batch % 16 != 0 || d.M % 4 != 0) return false;
    if (d.x3) {         // split-bf16: blocked operands, K tiles of 32 real elements (d.K = the virtual 3 K), fp32 V^T
        if (d.x3 != 2 || d.K % 96 != 0 || d.K < 192 || d.a_lo != 32) return false;
    } else if (d.K % 64 != 0 || d.K < 128) return false;
    if ((long)(d.M + 256) * d.lda * 2 >= lim || (long)d.N * d.Kpad * 2 >= lim) return false;
    if ((long)((d.M + d.rows_per_batch - 1) / d.rows_per_batch) * d.N * d.ldo * (d.x3 ? 4 : 2) >= lim) return false;
    long best = -1;
    for (int h : {256, 192}) {
        if (d.M < h) continue;
        const long tiles = (long)((d.M + h - 1) / h) * (d.N / *bn);
        const long cost = ((tiles + device_cus() - 1) / device_cus()) * h;
        if (best < 0 || cost < best) { best = cost; *bm = h; }
    }
    return best >= 0;
}
template <bool X3 = false>
static int launch_pp_trans(hipStream_t s, const ffn_igemm_desc& d, int bm, int bn) {
    const int pplds = 2 * (bm + bn) * 128 + 12288;
    const int nt = ((d.M + bm - 1) / bm) * (d.N / bn);
    const int grid = nt < device_cus() ? nt : device_cus();
    int rc = FFN_OK;
    (void)hipGetLastError();
#define FFN_PP_TR(BM_, BN_)                                                                         \
    do {                                                                                            \
        auto kern = igemm_pp_kernel<BM_, BN_, AMODE_DENSE, false, false, false, true, X3>;          \
        if ((rc = set_lds(kern, pplds))) return rc;                                                 \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), pplds, s, d, 1);                            \
    } while (0)
    if (bm == 256 && bn == 320) FFN_PP_TR(256, 320);
    else if (bm == 256) FFN_PP_TR(256, 256);
    else if (bn == 320) FFN_PP_TR(192, 320);
    else FFN_PP_TR(192, 256);
#undef FFN_PP_TR
    return check_launch("igemm(ping-pong, transposed)");
}
template <typename T>
static int dispatch_igemm(hipStream_t s, const ffn_igemm_desc& d) {
    const bool tr = d.flags & FFN_IG_OUT_TRANSPOSED;
    if (d.conv) {
        if (tr) return fail(FFN_EINVAL, "igemm: transposed output is only supported for dense A");
        if constexpr (sizeof(T) == 2) return tuned_bf16<AMODE_CONV3>(s, d);
        return dispatch_igemm_tile<T, AMODE_CONV3, true>(s, d);
    }
    if (tr) {
        if constexpr (sizeof(T) == 2) {
            int bm, bn;
            if (pp_trans_tile(d, &bm, &bn)) return launch_pp_trans(s, d, bm, bn);
        }
        return dispatch_igemm_tile<T, AMODE_DENSE, false>(s, d);
    }
    if constexpr (sizeof(T) == 2) return tuned_bf16<AMODE_DENSE>(s, d);
    return dispatch_igemm_tile<T, AMODE_DENSE, true>(s, d);
}
// the library's private view of a split-bf16 problem: the kernels and the tile / split-K logic see the VIRTUAL contraction 3K
static ffn_igemm_desc x3_view(const ffn_igemm_desc& d) {
    ffn_igemm_desc v = d;
    v.x3 = d.x3 == 2 ? 2 : 1;          // operand layout: 1 = planes, 2 = 128-byte blocks [hi(32) | lo(32)] (include/freefine_hip.h)
    v.f8 = 0;
    v.K = 3 * d.K;
    v.flags |= FFN_IG_OUT_F32;
    return v;
}
template <int BM, int BN>
static int launch_x3_trans(hipStream_t s, const ffn_igemm_desc& d) {
    const int ntiles = ((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN);
    return launch_igemm_kernel(igemm_glds_kernel<bf16, BM, BN, AMODE_DENSE, false, 2, 2, 2, false, true>, 2 * (BM + BN) * 128, s, d, ntiles, 1, 256, true);
}
static int dispatch_igemm_x3(hipStream_t s, const ffn_igemm_desc& d) {
    const bool tr = d.flags & FFN_IG_OUT_TRANSPOSED;
    if (d.conv) {
        if (tr) return fail(FFN_EINVAL, "igemm: transposed output is only supported for dense A");
        return tuned_bf16<AMODE_CONV3, true>(s, d);
    }
    if (tr) {            // V^T for the attention kernels, fp32 transposed stores: the ping-pong tile where it applies, else a generic tile
        int bm, bn, sk;
        if (pp_trans_tile(d, &bm, &bn)) return launch_pp_trans<true>(s, d, bm, bn);
        igemm_plan_for(FFN_BF16, d, &bm, &bn, &sk);
        if (bm == 128 && bn == 128) return launch_x3_trans<128, 128>(s, d);
        if (bm == 128) return launch_x3_trans<128, 64>(s, d);
        return launch_x3_trans<64, 64>(s, d);
    }
    return tuned_bf16<AMODE_DENSE, true>(s, d);
}
// fp8 problems: the kernels and every tile / split-K decision see a bf16-SHAPED view (two e4m3 bytes = one "element")
static ffn_igemm_desc f8_view(const ffn_igemm_desc& d) {
    ffn_igemm_desc v = d;
    v.f8 = 1;
    v.x3 = 0;
    v.K = d.K / 2;
    v.Cin = d.Cin / 2;
    v.Kpad = d.Kpad / 2;
    v.lda = d.lda / 2;
    return v;
}
extern "C" int ffn_split_pair(void* stream, const float* src, void* dst, long rows, int C, int ld_src) {
    REQUIRE(src && dst && rows > 0 && C > 0 && C % 4 == 0 && ld_src >= C && ld_src % 4 == 0, "split_pair: bad arguments (C=%d, ld_src=%d)", C, ld_src);
    REQUIRE(aligned16(src) && aligned16(dst), "split_pair: pointers must be 16-byte aligned");
    static const int wide = [] { const char* e = getenv("FFN_PAIR8"); return e ? atoi(e) : 1; }();
    if (wide && C % 8 == 0 && ld_src % 4 == 0) LAUNCH(split_pair8_kernel, dim3(grid_for(rows * (C / 8))), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, (bf16*)dst, rows, C, ld_src);
    else LAUNCH(split_pair_kernel, dim3(grid_for(rows * (C / 4))), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, (bf16*)dst, rows, C, ld_src);
    return check_launch("split_pair");
}
extern "C" int ffn_conv3x3_n4(void* stream, int dtype, const void* x, const float* w, const float* bias, float* out, int B, int H, int W, int Cin) {
    REQUIRE(dtype == FFN_F32 || dtype == FFN_BF16, "conv3x3_n4: fp32 or bf16 activations");
    REQUIRE(x && w && out && aligned16(x) && aligned16(w) && aligned16(out) && (!bias || aligned16(bias)), "conv3x3_n4: null / unaligned pointer");
    REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cin % 16 == 0 && Cin <= 448, "conv3x3_n4: B=%d H=%d W=%d Cin=%d (Cin %% 16 == 0, <= 448)", B, H, W, Cin);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long npix = (long)B * H * W;
    const int lds = 36 * Cin * 4;
    dim3 grid((unsigned)((npix + 63) / 64));
    int rc;
    if (dtype == FFN_F32) {
        if ((rc = set_lds(conv3x3_n4_kernel<float>, lds))) return rc;
        LAUNCH(conv3x3_n4_kernel<float>, grid, dim3(256), lds, s, (const float*)x, w, bias, out, B, H, W, Cin);
    } else {
        if ((rc = set_lds(conv3x3_n4_kernel<bf16>, lds))) return rc;
        LAUNCH(conv3x3_n4_kernel<bf16>, grid, dim3(256), lds, s, (const bf16*)x, w, bias, out, B, H, W, Cin);
    }
    return check_launch("conv3x3_n4");
}
extern "C" int ffn_igemm(void* stream, int dtype, const ffn_igemm_desc* d) {
    REQUIRE(d, "igemm: null descriptor");
    REQUIRE(dtype == FFN_F32 || dtype == FFN_BF16 || dtype == FFN_BF16X3 || dtype == FFN_FP8, "igemm: bad dtype %d", dtype);
    if (dtype == FFN_FP8) {
        REQUIRE(d->A && d->W && d->out && aligned16(d->A) && aligned16(d->W) && aligned16(d->out), "igemm(fp8): A/W/out must be non-null and 16-byte aligned");
        REQUIRE(d->conv && d->Cin > 0 && d->Cin % 16 == 0 && d->K == 9 * d->Cin && d->Kpad >= d->K && d->Kpad % 16 == 0 && d->lda == d->Cin,
                "igemm(fp8): 3x3 convolutions with Cin %% 16 == 0 only (Cin=%d, K=%d, Kpad=%d, lda=%d)", d->Cin, d->K, d->Kpad, d->lda);
        REQUIRE(d->M > 0 && d->N > 0 && d->N % 4 == 0 && d->ldo % 4 == 0 && d->rows_per_batch > 0 && d->M % (d->Hout * d->Wout) == 0, "igemm(fp8): bad shape");
        REQUIRE(!(d->flags & (FFN_IG_GEGLU | FFN_IG_OUT_TRANSPOSED | FFN_IG_OUT_PAIR | FFN_IG_OUT_F32)), "igemm(fp8): plain / SiLU / residual epilogues only");
        REQUIRE(d->stride == 1 || d->stride == 2, "igemm(fp8): stride %d", d->stride);
        int ex = 0;
        REQUIRE(d->alpha > 0.f && frexpf(d->alpha, &ex) == 0.5f, "igemm(fp8): alpha must be a power of two (the un-scaling of two power-of-two operand scales)");
        if (d->residual) REQUIRE(d->ldr % 4 == 0, "igemm(fp8): ldr=%d must be a multiple of 4", d->ldr);
        if (d->ws) REQUIRE(aligned16(d->ws) && d->ws_bytes >= 0, "igemm(fp8): workspace must be 16-byte aligned");
        return tuned_bf16<AMODE_CONV3, false, true>(reinterpret_cast<hipStream_t>(stream), f8_view(*d));
    }
    const int epc = dtype == FFN_F32 ? 4 : 8, kstage = 8 * epc;
    const int kmul = dtype == FFN_BF16X3 ? (d->x3 == 2 ? 2 : 3) : 1;       // W row: [hi | lo | hi] planes (3 K) or [hi(32) | lo(32)] blocks (2 K)
    REQUIRE(d->A && d->W && d->out, "igemm: null A/W/out");
    REQUIRE(aligned16(d->A) && aligned16(d->W) && aligned16(d->out), "igemm: A/W/out must be 16-byte aligned");
    REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "igemm: empty problem M=%d N=%d K=%d", d->M, d->N, d->K);
    REQUIRE(d->Kpad >= kmul * d->K && d->Kpad % epc == 0, "igemm: Kpad=%d (row stride of W) must be >= %d x K=%d and a multiple of %d", d->Kpad, kmul, d->K, epc);
    if (dtype == FFN_BF16X3) {
        const int plane = d->conv ? d->Cin : d->K;
        if (d->x3 == 2) {
            REQUIRE(plane % 32 == 0 && d->a_lo == 32 && d->lda >= 2 * plane, "igemm: blocked split-bf16 operands need K (conv: Cin) %% 32 == 0, a_lo = 32, lda >= 2 x that (%d, a_lo=%d, lda=%d)", plane, d->a_lo, d->lda);
        } else {
            REQUIRE(d->a_lo % 8 == 0 && d->a_lo >= plane && d->a_lo + plane <= d->lda, "igemm: split-bf16 A needs planes of %d elements: a_lo=%d, lda=%d", plane, d->a_lo, d->lda);
        }
        REQUIRE(d->lda % 8 == 0, "igemm: lda=%d must be a multiple of 8", d->lda);
        REQUIRE(d->alpha == 1.0f, "igemm: split-bf16 problems take alpha = 1");
        if (d->residual) REQUIRE(aligned16(d->residual), "igemm: fp32 residual must be 16-byte aligned");
    }
    (void)kstage;
    REQUIRE(d->K % epc == 0, "igemm: K=%d must be a multiple of %d", d->K, epc);
    REQUIRE(d->rows_per_batch > 0, "igemm: rows_per_batch must be > 0");
    if (d->conv) {
        REQUIRE(d->Cin % epc == 0, "igemm: Cin=%d must be a multiple of %d", d->Cin, epc);
        if (d->conv == 2) {
            REQUIRE((dtype == FFN_BF16 || dtype == FFN_BF16X3) && d->K == 4 * d->Cin && d->stride == 1 && d->upsample == 0 && d->pad >= 0 && d->pad <= 3 && d->splitk <= 1,
                    "igemm: 2x2 convolution needs FFN_BF16 / FFN_BF16X3, K = 4*Cin, stride 1, no upsample, pad in 0..3, no forced split");
            const ffn_igemm_desc v = dtype == FFN_BF16X3 ? x3_view(*d) : *d;
            bool any = false;
            for (int bn : {320, 256})
                for (int h : {256, 192}) any |= d->N % bn == 0 && pp_ok(v, h, bn);
            REQUIRE(any, "igemm: 2x2 convolution M=%d N=%d Cin=%d fits no ping-pong tile (Cin %% 64 (split-bf16: 32), N %% 256 / 320, M >= 192)", d->M, d->N, d->Cin);
        } else {
            REQUIRE(d->conv == 1, "igemm: conv=%d", d->conv);
            REQUIRE(d->K == 9 * d->Cin, "igemm: conv K=%d != 9*Cin=%d", d->K, 9 * d->Cin);
        }
        REQUIRE(d->stride == 1 || d->stride == 2, "igemm: stride %d", d->stride);
        REQUIRE(d->upsample == 0 || d->upsample == 1, "igemm: upsample %d", d->upsample);
        REQUIRE(d->M % (d->Hout * d->Wout) == 0, "igemm: M=%d not a multiple of Hout*Wout", d->M);
    } else {
        REQUIRE(d->lda % epc == 0, "igemm: lda=%d must be a multiple of %d", d->lda, epc);
    }
    if (d->flags & FFN_IG_OUT_KV64) {
        REQUIRE(dtype == FFN_BF16X3 && !d->conv && !d->residual && !d->rowbias && d->splitk <= 1 &&
                    !(d->flags & (FFN_IG_GEGLU | FFN_IG_OUT_PAIR | FFN_IG_OUT_SILU | FFN_IG_OUT_GELU | FFN_IG_OUT_RELU)),
                "igemm: FFN_IG_OUT_KV64 needs FFN_BF16X3, dense A, the plain epilogue and no forced split-K");
        if (d->flags & FFN_IG_OUT_TRANSPOSED) REQUIRE(d->rows_per_batch % 64 == 0 && d->M % d->rows_per_batch == 0 && d->ldo >= d->rows_per_batch && d->ldo % 64 == 0,
                                                      "igemm: KV64 transposed output needs rows_per_batch %% 64 == 0, whole batches, ldo %% 64 == 0 (rows_per_batch=%d, ldo=%d)", d->rows_per_batch, d->ldo);
        else REQUIRE(d->kv64_from >= 0 && d->kv64_from < d->N && d->kv64_from % 64 == 0 && d->N % 64 == 0 && d->ldo % 4 == 0,
                     "igemm: KV64 output needs kv64_from and N %% 64 == 0 (kv64_from=%d, N=%d)", d->kv64_from, d->N);
    }
    if (d->flags & FFN_IG_OUT_TRANSPOSED) {
        REQUIRE(d->ldo % 4 == 0, "igemm: transposed ldo=%d must be a multiple of 4", d->ldo);
        REQUIRE(!(d->flags & (FFN_IG_GEGLU | FFN_IG_OUT_F32 | FFN_IG_OUT_SILU | FFN_IG_OUT_GELU | FFN_IG_OUT_RELU)) && !d->residual && !d->rowbias,
                "igemm: transposed output supports bias only");
    } else {
        REQUIRE(d->N % 4 == 0 && d->ldo % 4 == 0, "igemm: N=%d and ldo=%d must be multiples of 4", d->N, d->ldo);
        if (d->residual) REQUIRE(d->ldr % 4 == 0, "igemm: ldr=%d must be a multiple of 4", d->ldr);
        if (d->flags & FFN_IG_OUT_PAIR) {
            const int nout = (d->flags & FFN_IG_GEGLU) ? d->N / 2 : d->N;
            REQUIRE(dtype == FFN_BF16X3 && d->ldo % 16 == 0 && d->ldo / 2 >= nout, "igemm: pair output needs FFN_BF16X3, ldo %% 16 == 0, ldo/2 >= columns");
            REQUIRE(d->residual != d->out, "igemm: pair output cannot overwrite its fp32 residual");
            REQUIRE((d->ldo / 2) % 32 != 0 || nout % 32 == 0 || nout == d->ldo / 2, "igemm: blocked pair output (ldo/2 %% 32 == 0) needs whole 32-column blocks");
            // every producer derives the layout (blocked / planes) from the ROW WIDTH ldo / 2; the ping-pong GEGLU epilogue always writes blocked rows
            REQUIRE(nout == d->ldo / 2 || (d->ldo / 2) % 32 == 0, "igemm: pair output into a wider row needs ldo/2 %% 32 == 0 (ldo=%d, columns=%d)", d->ldo, nout);
        }
        if (d->flags & FFN_IG_GEGLU) {
            REQUIRE(d->N % 64 == 0, "igemm: GEGLU needs N %% 64 == 0 (N=%d)", d->N);
            REQUIRE(!d->residual && !d->rowbias && !(d->flags & (FFN_IG_OUT_F32 | FFN_IG_OUT_SILU | FFN_IG_OUT_GELU | FFN_IG_OUT_RELU)), "igemm: GEGLU epilogue is exclusive");
        }
    }
    {
        const int act = d->flags & (FFN_IG_OUT_SILU | FFN_IG_OUT_GELU | FFN_IG_OUT_RELU);
        REQUIRE((act & (act - 1)) == 0, "igemm: SILU / GELU / RELU are mutually exclusive");
    }
    if (d->ws) REQUIRE(aligned16(d->ws) && d->ws_bytes >= 0, "igemm: workspace must be 16-byte aligned");
    REQUIRE(d->splitk >= 0, "igemm: splitk must be >= 0");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == FFN_BF16X3) return dispatch_igemm_x3(s, x3_view(*d));
    ffn_igemm_desc plain = *d;
    plain.x3 = plain.f8 = 0;
    return dtype == FFN_F32 ? dispatch_igemm<float>(s, plain) : dispatch_igemm<bf16>(s, plain);
}

extern "C" int ffn_igemm_tune(void* stream, int dtype, const ffn_igemm_desc* d) { return ffn_igemm(stream, dtype, d); }

// ---- attention ---------------------------------------------------------------------------------------------------
template <typename T, int DP, int QF, int KT = 64, int OCC = 1, bool MASKS = true>
static int launch_attn(hipStream_t s, const ffn_attn_desc& d) {
    constexpr int SZ = sizeof(T);
    // double-buffered K and V^T tiles + the per-wave multi-pass accumulator
    constexpr int krow = (DP * SZ == 128) ? 128 : DP * SZ + 16;
    constexpr int vrow = (DP * SZ == 128 && KT * SZ == 128) ? 128 : KT * SZ + 16;
    constexpr int lds = 2 * (KT * krow + DP * vrow) + 4 * (DP / 16) * QF * 64 * 16 + 2 * KT;   // + key-mask bytes of the two staged tiles
    static_assert(lds <= 160 * 1024, "attention tile does not fit the 160 KiB LDS");
    auto kern = attn_kernel<T, DP, QF, KT, OCC, MASKS>;
    if (int rc = set_lds(kern, lds)) return rc;      // memoised under a mutex: safe from several host threads
    // 1-D grid, decoded in the kernel through xcd_remap: the query blocks of one (row, head) run on ONE XCD, whose 4 MiB L2 then
    // serves that head's K / V^T (1 MiB at S = 4096) to all of them (measured before: 721 MB fetched for 126 MB of operands)
    dim3 grid(((d.S + 64 * QF - 1) / (64 * QF)) * d.heads * d.Bo);
    LAUNCH(kern, grid, dim3(256), lds, s, d);
    return check_launch("attn");
}
// bf16 launches: whether some active entry carries a key mask (-> the kernel variant with the mask-on-MFMA tile) and whether the
// launch runs on the ping-pong schedule (attention_pp.h): d = 64, whole 64-key tiles, at least half a 256-query workgroup of queries,
// no degenerate uniform-softmax entries (those need the generic tile)
static void attn_bf16_choice(const ffn_attn_desc& d, bool* masks, bool* pp) {
    bool m = false, uniform = false;
    for (int pi = 0; pi < d.npass; ++pi)
        for (int b = 0; b < d.Bo; ++b) {
            const ffn_attn_entry& e = d.e[pi * FFN_ATT_MAXB + b];
            if (e.w_const == 0.f && e.w_slope == 0.f) continue;
            m |= e.kmask != nullptr;
            uniform |= e.kmask != nullptr && (e.flags & (FFN_ATT_UNIFORM_SEL1 | FFN_ATT_UNIFORM_SEL0));
        }
    static const bool pp_on = [] { const char* e = getenv("FFN_ATTN_PP"); return !(e && atoi(e) == 0); }();
    *masks = m;
    *pp = pp_on && d.D == 64 && d.Sk % 64 == 0 && d.S >= 128 && !uniform;
}
// cross attention against a short key sequence (attention_x.h): bf16, d = 64, Sk <= 96, ONE pass whose entries are all active and carry
// no mask / selector / per-query weight.  Returns the key-fragment count of the instantiation (0: not this kernel).
// *multi = 1: the launch has several passes, skipped entries or per-query weights -> xattn_mp_kernel (K / V^T fragment images in LDS).
// esz = bytes per operand element: 2 (bf16 kernels) or 4 (xattn_x3_kernel, attention_xx3.h: fp32 operands in split-bf16 arithmetic)
static int xattn_nkf(const ffn_attn_desc& d, int* multi = nullptr, int esz = 2) {
    static const bool on = [] { const char* e = getenv("FFN_ATTN_X"); return !(e && atoi(e) == 0); }();
    if (!on || d.D != 64 || d.Sk > 96 || (esz == 2 && d.ldo % 8 != 0)) return 0;
    int maxq = 0, maxkv = 0, mp = d.npass != 1;
    for (int pi = 0; pi < d.npass; ++pi)
        for (int b = 0; b < d.Bo; ++b) {
            const ffn_attn_entry& e = d.e[pi * FFN_ATT_MAXB + b];
            if (e.w_const == 0.f && e.w_slope == 0.f) { mp = 1; continue; }
            if (e.kmask || e.qsel) return 0;                    // (flags only qualify a key mask)
            if (e.w_slope != 0.f && !d.w_dev) return 0;
            if (e.wq) mp = 1;
            maxq = e.q_row > maxq ? e.q_row : maxq;
            maxkv = e.kv_row > maxkv ? e.kv_row : maxkv;
        }
    if (multi) *multi = mp;
    const long lim = (1l << 31) - 65536;              // 32-bit byte offsets into every operand
    if ((long)(maxq + 1) * d.S * d.ldq * esz >= lim || (long)d.Bo * d.S * d.ldo * esz >= lim || (long)(maxkv + 1) * d.Sk * d.ldk * esz >= lim ||
        (long)(maxkv + 1) * d.heads * 64 * d.ldvt * esz >= lim)
        return 0;
    const int need = (d.Sk + 15) / 16;
    return need <= 2 ? 2 : (need <= 5 ? 5 : 6);
}
static int launch_xattn(hipStream_t s, const ffn_attn_desc& d, int nkf, int multi) {
    const int pairs = d.Bo * d.heads, nblk = (d.S + 31) / 32;
    if (multi) {                                      // one workgroup of 4 waves per (row, head, chunk): fragment images of all passes in LDS
        int wpp = (8 * device_cus()) / pairs / 4;     // workgroups per (row, head)
        if (wpp < 1) wpp = 1;
        if (wpp > (nblk + 3) / 4) wpp = (nblk + 3) / 4;
        const int bpw = (nblk + 4 * wpp - 1) / (4 * wpp);
        wpp = (nblk + 4 * bpw - 1) / (4 * bpw);
        const int nfr = nkf * 2 + 4 * ((nkf + 1) / 2);
        const int lds = d.npass * nfr * 1024;
        dim3 grid(pairs * wpp);
        int rc;
        if (nkf == 2) { if ((rc = set_lds(xattn_mp_kernel<2>, lds))) return rc; LAUNCH(xattn_mp_kernel<2>, grid, dim3(256), lds, s, d, wpp, bpw); }
        else if (nkf == 5) { if ((rc = set_lds(xattn_mp_kernel<5>, lds))) return rc; LAUNCH(xattn_mp_kernel<5>, grid, dim3(256), lds, s, d, wpp, bpw); }
        else { if ((rc = set_lds(xattn_mp_kernel<6>, lds))) return rc; LAUNCH(xattn_mp_kernel<6>, grid, dim3(256), lds, s, d, wpp, bpw); }
        return check_launch("attn(cross, multi-pass)");
    }
    int wpp = (8 * device_cus()) / pairs;             // waves per (row, head): fill the chip's 8 waves per CU once
    if (wpp < 1) wpp = 1;
    if (wpp > nblk) wpp = nblk;
    const int bpw = (nblk + wpp - 1) / wpp;
    wpp = (nblk + bpw - 1) / bpw;
    dim3 grid((pairs * wpp + 3) / 4);
    if (nkf == 2) LAUNCH(xattn_kernel<2>, grid, dim3(256), 0, s, d, wpp, bpw);
    else if (nkf == 5) LAUNCH(xattn_kernel<5>, grid, dim3(256), 0, s, d, wpp, bpw);
    else LAUNCH(xattn_kernel<6>, grid, dim3(256), 0, s, d, wpp, bpw);
    return check_launch("attn(cross)");
}
// the split-bf16 form: always the LDS-fragment-image structure (hi and lo images of every active pass: 2 x nfr KiB per pass)
template <int NW>
static int launch_xattn_x3_nw(hipStream_t s, const ffn_attn_desc& d, int nkf, int waves_per_cu) {
    const int pairs = d.Bo * d.heads, nblk = (d.S + 31) / 32;
    int wpp = (waves_per_cu * device_cus()) / pairs / NW;         // workgroups per (row, head)
    if (wpp < 1) wpp = 1;
    if (wpp > (nblk + NW - 1) / NW) wpp = (nblk + NW - 1) / NW;
    const int bpw = (nblk + NW * wpp - 1) / (NW * wpp);
    wpp = (nblk + NW * bpw - 1) / (NW * bpw);
    const int nfr = nkf * 2 + 4 * ((nkf + 1) / 2);
    const int lds = d.npass * 2 * nfr * 1024;
    if (lds > 160 * 1024) return -1;                  // (more passes than fit: the caller takes the generic kernel)
    dim3 grid(pairs * wpp);
    int rc;
    if (nkf == 2) { if ((rc = set_lds(xattn_x3_kernel<2, NW>, lds))) return rc; LAUNCH((xattn_x3_kernel<2, NW>), grid, dim3(64 * NW), lds, s, d, wpp, bpw); }
    else if (nkf == 5) { if ((rc = set_lds(xattn_x3_kernel<5, NW>, lds))) return rc; LAUNCH((xattn_x3_kernel<5, NW>), grid, dim3(64 * NW), lds, s, d, wpp, bpw); }
    else { if ((rc = set_lds(xattn_x3_kernel<6, NW>, lds))) return rc; LAUNCH((xattn_x3_kernel<6, NW>), grid, dim3(64 * NW), lds, s, d, wpp, bpw); }
    return check_launch("attn(cross, split-bf16)");
}
static int launch_xattn_x3(hipStream_t s, const ffn_attn_desc& d, int nkf) {
    // 8 waves per workgroup where the fragment images are large against a workgroup's share of the queries (two passes: 88 KiB, one workgroup per CU
    // either way) or the launch is long; 4 (two workgroups per CU) for the short single-pass launches (profiles/r5_xattn_x3_waves_and_stores.txt)
    static const int nw_env = [] { const char* e = getenv("FFN_XATT_NW"); return e ? atoi(e) : 0; }();
    const int nw = nw_env ? nw_env : ((d.npass >= 2 || d.S >= 4096) ? 8 : 4);
    return nw == 8 ? launch_xattn_x3_nw<8>(s, d, nkf, 8) : launch_xattn_x3_nw<4>(s, d, nkf, 8);
}
static bool xattn_x3_fits(const ffn_attn_desc& d, int nkf) { return d.npass * 2 * (nkf * 2 + 4 * ((nkf + 1) / 2)) * 1024 <= 160 * 1024; }
// attn_x3w_kernel lives in its own translation unit (attn_x3w.hip: different code generation flags)
extern "C" __attribute__((visibility("hidden"))) int fx3w_launch(hipStream_t s, const ffn_attn_desc* d, int masks);
static bool attn_x3w_on() {       // FFN_ATTN_X3W=0: attn_x3p_kernel<., PAIRKV> (round 5) instead; read per call so that tests can compare the two
    const char* e = getenv("FFN_ATTN_X3W");
    return !e || atoi(e) != 0;
}
static bool attn_has_masks(const ffn_attn_desc& d) {
    for (int pi = 0; pi < d.npass; ++pi)
        for (int b = 0; b < d.Bo; ++b) {
            const ffn_attn_entry& e = d.e[pi * FFN_ATT_MAXB + b];
            if ((e.w_const != 0.f || e.w_slope != 0.f) && e.kmask) return true;
        }
    return false;
}
extern "C" int ffn_attn_presplit(void* stream, const float* k, const float* vt, void* k_pair, void* vt_pair, int rows, int Sk, int heads, int ldk, int ldvt) {
    REQUIRE(k && vt && k_pair && vt_pair && aligned16(k) && aligned16(vt) && aligned16(k_pair) && aligned16(vt_pair), "attn_presplit: null / unaligned pointer");
    REQUIRE(rows > 0 && heads > 0 && Sk > 0 && Sk % 64 == 0 && ldk >= heads * 64 && ldk % 4 == 0 && ldvt >= Sk && ldvt % 4 == 0,
            "attn_presplit: rows=%d Sk=%d heads=%d ldk=%d ldvt=%d (head dim 64, Sk %% 64 == 0)", rows, Sk, heads, ldk, ldvt);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long nk = (long)rows * Sk * heads * 8, nv = (long)rows * heads * 64 * (Sk / 64) * 8;
    LAUNCH(attn_presplit_k_kernel, dim3(grid_for(nk)), dim3(256), 0, s, k, (bf16*)k_pair, nk, heads, ldk);
    LAUNCH(attn_presplit_vt_kernel, dim3(grid_for(nv)), dim3(256), 0, s, vt, (bf16*)vt_pair, nv, Sk / 64, ldvt);
    return check_launch("attn_presplit");
}
extern "C" int ffn_attn_kernel_name(int dtype, const ffn_attn_desc* d, char* buf, int len) {
    REQUIRE(d && buf && len > 0, "attn_kernel_name: null argument");
    int dp = 0, qf = 0;
    if (dtype == FFN_BF16X3 && d->D <= 64) {
        if (const int nkf = xattn_nkf(*d, nullptr, 4)) {
            if (xattn_x3_fits(*d, nkf)) {
                snprintf(buf, len, "void xattn_x3_kernel<%d, %d>(ffn_attn_desc, int, int)", nkf, (d->npass >= 2 || d->S >= 4096) ? 8 : 4);
                return FFN_OK;
            }
        }
        bool masks, pp;
        attn_bf16_choice(*d, &masks, &pp);
        if (pp && d->kv_pair && attn_x3w_on()) snprintf(buf, len, "void attn_x3w_kernel<%s>(ffn_attn_desc)", attn_has_masks(*d) ? "true" : "false");
        else if (pp) snprintf(buf, len, "void attn_x3p_kernel<%s, %s>(ffn_attn_desc)", attn_has_masks(*d) ? "true" : "false", d->kv_pair ? "true" : "false");
        else snprintf(buf, len, "void attn_x3_kernel<%s>(ffn_attn_desc)", attn_has_masks(*d) ? "true" : "false");
        return FFN_OK;
    }
    if (dtype == FFN_BF16X3) dtype = FFN_F32;
    ffn_attn_variant(dtype, d->D, &dp, &qf);
    if (dtype == FFN_F32) {
        snprintf(buf, len, "void attn_kernel<float, %d, %d, %d, 1, true>(ffn_attn_desc)", dp, qf, dp == 160 ? 32 : 64);
        return FFN_OK;
    }
    int multi = 0;
    if (const int nkf = xattn_nkf(*d, &multi)) {
        snprintf(buf, len, "void %s<%d>(ffn_attn_desc, int, int)", multi ? "xattn_mp_kernel" : "xattn_kernel", nkf);
        return FFN_OK;
    }
    bool masks, pp;
    attn_bf16_choice(*d, &masks, &pp);
    if (pp) snprintf(buf, len, "void attn_pp_kernel<%s>(ffn_attn_desc)", masks ? "true" : "false");
    else snprintf(buf, len, "void attn_kernel<bf16, %d, %d, 64, %d, %s>(ffn_attn_desc)", dp, qf, dp == 64 ? 2 : 1, masks ? "true" : "false");
    return FFN_OK;
}
extern "C" int ffn_attn(void* stream, int dtype, const ffn_attn_desc* d) {
    REQUIRE(d, "attn: null descriptor");
    REQUIRE(dtype == FFN_F32 || dtype == FFN_BF16 || dtype == FFN_BF16X3, "attn: bad dtype %d", dtype);
    const int epc = dtype == FFN_BF16 ? 8 : 4;
    REQUIRE(d->q && d->k && d->vt && d->out, "attn: null q/k/vt/out");
    REQUIRE(aligned16(d->q) && aligned16(d->k) && aligned16(d->vt) && aligned16(d->out), "attn: pointers must be 16-byte aligned");
    REQUIRE(d->Bo > 0 && d->Bo <= FFN_ATT_MAXB, "attn: Bo=%d out of range (max %d)", d->Bo, FFN_ATT_MAXB);
    REQUIRE(d->npass > 0 && d->npass <= FFN_ATT_MAXP, "attn: npass=%d out of range (max %d)", d->npass, FFN_ATT_MAXP);
    REQUIRE(d->S > 0 && d->Sk > 0 && d->heads > 0 && d->D > 0, "attn: empty problem");
    REQUIRE(d->D % epc == 0, "attn: D=%d must be a multiple of %d", d->D, epc);
    REQUIRE(d->ldq % epc == 0 && d->ldk % epc == 0 && d->ldvt % epc == 0 && d->ldo % 4 == 0, "attn: leading dims must be chunk aligned");
    REQUIRE(d->ldvt >= d->Sk, "attn: ldvt=%d < Sk=%d", d->ldvt, d->Sk);
    if (d->out_pair) {
        REQUIRE(dtype == FFN_BF16X3 && d->D <= 64 && d->ldo % 16 == 0 && d->ldo / 2 >= d->heads * d->D, "attn: pair output needs FFN_BF16X3, D <= 64, ldo %% 16 == 0");
        // the cross-attention kernel places pair columns by heads * D, the others by ldo / 2: the two agree in these cases only
        REQUIRE(d->ldo / 2 == d->heads * d->D || ((d->ldo / 2) % 32 == 0 && (d->heads * d->D) % 32 == 0), "attn: pair output into a wider row needs 32-column blocks (ldo=%d, heads*D=%d)", d->ldo, d->heads * d->D);
    }
    if (d->kv_pair) {
        bool m_, pp_;
        attn_bf16_choice(*d, &m_, &pp_);
        REQUIRE(dtype == FFN_BF16X3 && pp_ && !xattn_nkf(*d, nullptr, 4), "attn: kv_pair is for launches that run attn_x3p_kernel (FFN_BF16X3, D = 64, Sk %% 64 == 0, S >= 128, no uniform-softmax entry)");
        long maxkv = 0;
        for (int pi = 0; pi < d->npass; ++pi)
            for (int b = 0; b < d->Bo; ++b) maxkv = d->e[pi * FFN_ATT_MAXB + b].kv_row > maxkv ? d->e[pi * FFN_ATT_MAXB + b].kv_row : maxkv;
        REQUIRE(d->ldk >= d->heads * 64 && d->ldk % 64 == 0 && d->ldvt >= d->Sk && d->ldvt % 64 == 0,
                "attn: pre-split images need ldk >= heads * 64, ldvt >= Sk, both %% 64 == 0 (ldk=%d, ldvt=%d)", d->ldk, d->ldvt);
        REQUIRE((maxkv + 1) * d->Sk * (long)d->ldk * 4 < (1l << 31) - 65536 && (maxkv + 1) * d->heads * 64 * (long)d->ldvt * 4 < (1l << 31) - 65536,
                "attn: pre-split K / V^T images beyond 2 GiB (32-bit byte offsets)");
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int D = d->D;
    if (dtype == FFN_BF16X3 && D <= 64) {      // split-bf16 arithmetic on fp32 operands (attention_x3.h); other head sizes: the exact fp32 kernel
        if (const int nkf = xattn_nkf(*d, nullptr, 4)) {       // short unmasked key sequences (the text cross attention): attention_xx3.h
            if (xattn_x3_fits(*d, nkf)) return launch_xattn_x3(s, *d, nkf);
        }
        constexpr int lds = 2 * (4 * 8192) + 8 * 4 * 2 * 64 * 16;
        dim3 grid(((d->S + 255) / 256) * d->heads * d->Bo);
        int rc;
        bool masks_pp, pp;
        attn_bf16_choice(*d, &masks_pp, &pp);      // the ping-pong schedule has the same preconditions as attn_pp_kernel's (d = 64, Sk % 64 == 0, S >= 128, ...)
        if (pp) {
            constexpr int lds = 5 * (2 * 8192) + 8 * 4 * 2 * 64 * 16;      // K ring of 2 + V^T ring of 3 [hi | lo] images, multi-pass sums
            if (d->kv_pair && attn_x3w_on()) {      // round 6: one wave per SIMD on 32x32x16 MFMAs (attention_x3w.h), same pre-split images
                const hipError_t e = (hipError_t)fx3w_launch(s, d, attn_has_masks(*d) ? 1 : 0);
                if (e != hipSuccess) return fail(FFN_EHIP, "attn(split-bf16, one wave per SIMD): %s", hipGetErrorString(e));
                return FFN_OK;
            }
            if (d->kv_pair) {                       // pre-split K / V^T images by LDS-DMA (+ 512 B of key-mask bytes)
                constexpr int ldsp = lds + 512;
                if (attn_has_masks(*d)) {
                    if ((rc = set_lds(attn_x3p_kernel<true, true>, ldsp))) return rc;
                    LAUNCH((attn_x3p_kernel<true, true>), grid, dim3(512), ldsp, s, *d);
                } else {
                    if ((rc = set_lds(attn_x3p_kernel<false, true>, ldsp))) return rc;
                    LAUNCH((attn_x3p_kernel<false, true>), grid, dim3(512), ldsp, s, *d);
                }
                return check_launch("attn(split-bf16, ping-pong, pre-split K/V)");
            }
            if (attn_has_masks(*d)) {
                if ((rc = set_lds(attn_x3p_kernel<true>, lds))) return rc;
                LAUNCH(attn_x3p_kernel<true>, grid, dim3(512), lds, s, *d);
            } else {
                if ((rc = set_lds(attn_x3p_kernel<false>, lds))) return rc;
                LAUNCH(attn_x3p_kernel<false>, grid, dim3(512), lds, s, *d);
            }
            return check_launch("attn(split-bf16, ping-pong)");
        }
        if (attn_has_masks(*d)) {
            if ((rc = set_lds(attn_x3_kernel<true>, lds))) return rc;
            LAUNCH(attn_x3_kernel<true>, grid, dim3(512), lds, s, *d);
        } else {
            if ((rc = set_lds(attn_x3_kernel<false>, lds))) return rc;
            LAUNCH(attn_x3_kernel<false>, grid, dim3(512), lds, s, *d);
        }
        return check_launch("attn(split-bf16)");
    }
    if (dtype == FFN_BF16X3) dtype = FFN_F32;
    if (dtype == FFN_F32) {
        if (D <= 48) return launch_attn<float, 48, 2>(s, *d);
        if (D <= 64) return launch_attn<float, 64, 2>(s, *d);
        if (D <= 80) return launch_attn<float, 80, 2>(s, *d);
        if (D <= 160) return launch_attn<float, 160, 1, 32>(s, *d);
    } else {
        int multi = 0;
        if (const int nkf = xattn_nkf(*d, &multi)) return launch_xattn(s, *d, nkf, multi);
        bool masks, pp;
        attn_bf16_choice(*d, &masks, &pp);
        if (pp) {
            constexpr int lds = 4 * 8192 + 4 * 8192 + 4 * 256 + 8 * 4 * 2 * 64 * 16;
            dim3 grid(((d->S + 255) / 256) * d->heads * d->Bo);
            int rc;
            if (masks) {
                if ((rc = set_lds(attn_pp_kernel<true>, lds))) return rc;
                LAUNCH(attn_pp_kernel<true>, grid, dim3(512), lds, s, *d);
            } else {
                if ((rc = set_lds(attn_pp_kernel<false>, lds))) return rc;
                LAUNCH(attn_pp_kernel<false>, grid, dim3(512), lds, s, *d);
            }
            return check_launch("attn(ping-pong)");
        }
        if (D <= 64) return masks ? launch_attn<bf16, 64, 2, 64, 2, true>(s, *d) : launch_attn<bf16, 64, 2, 64, 2, false>(s, *d);
        if (D <= 96) return masks ? launch_attn<bf16, 96, 2, 64, 1, true>(s, *d) : launch_attn<bf16, 96, 2, 64, 1, false>(s, *d);
        if (D <= 160) return masks ? launch_attn<bf16, 160, 1, 64, 1, true>(s, *d) : launch_attn<bf16, 160, 1, 64, 1, false>(s, *d);
    }
    return fail(FFN_ENOSYS, "attn: head dim %d not supported (max 160; use the GEMM path)", D);
}

// ---- elementwise / resampling helpers of the depth front end ----------------------------------------------------------
extern "C" int ffn_eltwise(void* stream, int dtype, int op, const void* a, const void* b, void* y, long n) {
    REQUIRE(dtype == FFN_F32 || dtype == FFN_BF16, "eltwise: bad dtype %d", dtype);
    REQUIRE(op == FFN_ELT_RELU || op == FFN_ELT_ADD, "eltwise: bad op %d", op);
    REQUIRE(a && y && (op != FFN_ELT_ADD || b), "eltwise: null operand");
    REQUIRE(n > 0 && n % 4 == 0, "eltwise: n=%ld must be a positive multiple of 4", n);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long n4 = n / 4;
    const int grid = grid_for(n4);
    if (dtype == FFN_F32) {
        if (op == FFN_ELT_RELU) LAUNCH((eltwise_kernel<float, FFN_ELT_RELU>), dim3(grid), dim3(256), 0, s, (const float*)a, (const float*)b, (float*)y, n4);
        else LAUNCH((eltwise_kernel<float, FFN_ELT_ADD>), dim3(grid), dim3(256), 0, s, (const float*)a, (const float*)b, (float*)y, n4);
    } else {
        if (op == FFN_ELT_RELU) LAUNCH((eltwise_kernel<bf16, FFN_ELT_RELU>), dim3(grid), dim3(256), 0, s, (const bf16*)a, (const bf16*)b, (bf16*)y, n4);
        else LAUNCH((eltwise_kernel<bf16, FFN_ELT_ADD>), dim3(grid), dim3(256), 0, s, (const bf16*)a, (const bf16*)b, (bf16*)y, n4);
    }
    return check_launch("eltwise");
}
extern "C" int ffn_resize_bilinear(void* stream, int dtype, const void* x, void* y, int B, int Hin, int Win, int Hout, int Wout, int C, int relu) {
    REQUIRE(dtype == FFN_F32 || dtype == FFN_BF16, "resize_bilinear: bad dtype %d", dtype);
    REQUIRE(x && y, "resize_bilinear: null operand");
    REQUIRE(B > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0 && C > 0 && C % 4 == 0, "resize_bilinear: bad shape (C=%d must be a multiple of 4)", C);
    REQUIRE((long)B * Hout * Wout * C < (1l << 40) && (long)Hin * Win < (1l << 31), "resize_bilinear: problem too large");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long n4 = (long)B * Hout * Wout * (C / 4);
    const int grid = grid_for(n4);
    if (dtype == FFN_F32) {
        if (relu) LAUNCH((resize_bilinear_kernel<float, true>), dim3(grid), dim3(256), 0, s, (const float*)x, (float*)y, n4, Hin, Win, Hout, Wout, C);
        else LAUNCH((resize_bilinear_kernel<float, false>), dim3(grid), dim3(256), 0, s, (const float*)x, (float*)y, n4, Hin, Win, Hout, Wout, C);
    } else {
        if (relu) LAUNCH((resize_bilinear_kernel<bf16, true>), dim3(grid), dim3(256), 0, s, (const bf16*)x, (bf16*)y, n4, Hin, Win, Hout, Wout, C);
        else LAUNCH((resize_bilinear_kernel<bf16, false>), dim3(grid), dim3(256), 0, s, (const bf16*)x, (bf16*)y, n4, Hin, Win, Hout, Wout, C);
    }
    return check_launch("resize_bilinear");
}

// ---- norms -------------------------------------------------------------------------------------------------------
extern "C" int ffn_gn_nchunk(int HW) {
    int n = HW / 128;
    if (n < 1) n = 1;
    if (n > 256) n = 256;
    return n;
}
// one fused launch (statistics + normalise + SiLU by the workgroup that owns a (row, group) slice) or stats + finalize + apply?
// Measured (tools/bench_kernels.py --only norm): the fused kernel reads 20-120 byte per-pixel group segments, so it only wins while
// the tensor is small enough for launch latency to dominate -- up to 8x8 positions at any batch, 16x16 up to 32 rows (at 48 rows,
// the image-batched guided pass, the three-launch form is 7-18 % faster there: 34.9 vs 37.5 us at C = 1280, 56.6 vs 66.1 at 2560),
// 32x32 below ~3M elements.
extern "C" int ffn_gn_fused(int B, int HW, int C, int G) {
    const long slice = (long)HW * (C / (G > 0 ? G : 1));
    if (slice > 131072) return 0;
    if (HW <= 64) return 1;
    if (HW <= 256) return B <= 32 ? 1 : 0;
    return (HW <= 1024 && (long)B * HW * C <= 3000000l) ? 1 : 0;
}
extern "C" int ffn_groupnorm(void* stream, int dtype, const void* x, void* y, const float* gamma, const float* beta, int B, int HW, int C,
                             int G, float eps, int silu, float* partial_ws, float* scale, float* shift) {
    REQUIRE(dtype == FFN_F32 || dtype == FFN_BF16, "groupnorm: bad dtype");
    REQUIRE(x && y && gamma && beta && C % G == 0 && (C / G) % 2 == 0, "groupnorm: bad arguments (C=%d, G=%d)", C, G);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long slice = (long)HW * (C / G);
    (void)slice;
    const bool fused = ffn_gn_fused(B, HW, C, G) != 0;
    const bool pair = silu & FFN_NORM_OUT_PAIR;
    REQUIRE(!pair || (dtype == FFN_F32 && C % 4 == 0), "groupnorm: pair output needs fp32 input and C %% 4 == 0");
    silu &= FFN_NORM_SILU;
    if (fused && pair) {
        dim3 grid(G, B);
        if (silu) LAUNCH((gn_fused_kernel<float, true, true>), grid, dim3(1024), 0, s, (const float*)x, (float*)y, gamma, beta, HW, C, G, eps);
        else LAUNCH((gn_fused_kernel<float, false, true>), grid, dim3(1024), 0, s, (const float*)x, (float*)y, gamma, beta, HW, C, G, eps);
        return check_launch("gn_fused(pair)");
    }
    if (fused) {   // one workgroup owns a whole (batch, group) slice: statistics + normalise + SiLU in ONE launch
                                           // (measured: wins up to 32x32 latents; at 64x64 the 20-60 byte per-pixel group segments coalesce badly)
        dim3 grid(G, B);
        if (dtype == FFN_F32) {
            if (silu) LAUNCH((gn_fused_kernel<float, true>), grid, dim3(1024), 0, s, (const float*)x, (float*)y, gamma, beta, HW, C, G, eps);
            else LAUNCH((gn_fused_kernel<float, false>), grid, dim3(1024), 0, s, (const float*)x, (float*)y, gamma, beta, HW, C, G, eps);
        } else {
            if (silu) LAUNCH((gn_fused_kernel<bf16, true>), grid, dim3(1024), 0, s, (const bf16*)x, (bf16*)y, gamma, beta, HW, C, G, eps);
            else LAUNCH((gn_fused_kernel<bf16, false>), grid, dim3(1024), 0, s, (const bf16*)x, (bf16*)y, gamma, beta, HW, C, G, eps);
        }
        return check_launch("gn_fused");
    }
    REQUIRE(partial_ws && scale && shift, "groupnorm: workspace required beyond 32x32 positions");
    int rc = ffn_gn_stats(stream, dtype, x, gamma, beta, B, HW, C, G, eps, partial_ws, scale, shift);
    if (rc) return rc;
    return ffn_gn_apply(stream, dtype, x, y, scale, shift, B, HW, C, silu | (pair ? FFN_NORM_OUT_PAIR : 0));
}
extern "C" int ffn_groupnorm_pair_raw(void* stream, const void* x, void* y, void* yraw, const float* gamma, const float* beta, int B, int HW, int C, int G,
                                      float eps, int silu, float* partial_ws, float* scale, float* shift) {
    REQUIRE(x && y && yraw && gamma && beta && partial_ws && scale && shift, "groupnorm_pair_raw: null pointer (the workspace is always needed)");
    REQUIRE(C % 8 == 0 && C % G == 0 && (C / G) % 2 == 0 && aligned16(x) && aligned16(y) && aligned16(yraw), "groupnorm_pair_raw: bad arguments (C=%d, G=%d)", C, G);
    REQUIRE(y != yraw && yraw != x && y != x, "groupnorm_pair_raw: x, y and yraw must be distinct buffers");
    int rc = ffn_gn_stats(stream, FFN_F32, x, gamma, beta, B, HW, C, G, eps, partial_ws, scale, shift);
    if (rc) return rc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long n8 = (long)B * HW * (C / 8);
    if (silu & FFN_NORM_SILU) LAUNCH((gn_apply_pair8_kernel<true, true>), dim3(grid_for(n8)), dim3(256), 0, s, (const float*)x, (bf16*)y, scale, shift, n8, HW, C, (bf16*)yraw);
    else LAUNCH((gn_apply_pair8_kernel<false, true>), dim3(grid_for(n8)), dim3(256), 0, s, (const float*)x, (bf16*)y, scale, shift, n8, HW, C, (bf16*)yraw);
    return check_launch("gn_apply(pair + raw pair)");
}
extern "C" int ffn_groupnorm_f8(void* stream, const void* x, void* y, const float* gamma, const float* beta, int B, int HW, int C, int Cp, int G,
                                float eps, int silu, float qscale, float* partial_ws, float* scale, float* shift) {
    REQUIRE(x && y && gamma && beta && partial_ws && scale && shift, "groupnorm_f8: null pointer (the workspace is always needed)");
    REQUIRE(C % 16 == 0 && C % G == 0 && Cp >= C && Cp % 16 == 0 && aligned16(x) && aligned16(y) && qscale > 0.f, "groupnorm_f8: bad arguments (C=%d, Cp=%d)", C, Cp);
    int rc = ffn_gn_stats(stream, FFN_BF16, x, gamma, beta, B, HW, C, G, eps, partial_ws, scale, shift);
    if (rc) return rc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long nch = (long)B * HW * (Cp / 16);
    if (silu & FFN_NORM_SILU) LAUNCH(gn_apply_f8_kernel<true>, dim3(grid_for(nch)), dim3(256), 0, s, (const bf16*)x, (uint8_t*)y, scale, shift, nch, HW, C, Cp, qscale);
    else LAUNCH(gn_apply_f8_kernel<false>, dim3(grid_for(nch)), dim3(256), 0, s, (const bf16*)x, (uint8_t*)y, scale, shift, nch, HW, C, Cp, qscale);
    return check_launch("gn_apply_f8");
}
extern "C" int ffn_gn_stats(void* stream, int dtype, const void* x, const float* gamma, const float* beta, int B, int HW, int C,
                            int G, float eps, float* partial_ws, float* scale, float* shift) {
    REQUIRE(dtype == FFN_F32 || dtype == FFN_BF16, "gn_stats: bad dtype");
    const int epc = dtype == FFN_F32 ? 4 : 8;
    REQUIRE(x && gamma && beta && partial_ws && scale && shift, "gn_stats: null pointer");
    REQUIRE(C % epc == 0 && C % G == 0 && aligned16(x), "gn_stats: C=%d must be a multiple of %d and of G=%d", C, epc, G);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int nchunk = ffn_gn_nchunk(HW);
    const int ppc = (HW + nchunk - 1) / nchunk;
    const int epc_ = dtype == FFN_F32 ? 4 : 8;
    const int cols_ = C / epc_ < 256 ? C / epc_ : 256;
    const int lds = (256 / cols_ > 0 ? 256 / cols_ : 1) * cols_ * 2 * epc_ * (int)sizeof(float);      // [pixel rows of threads][columns][2 x EPC]
    if (dtype == FFN_F32)
        LAUNCH(gn_partial_kernel<float>, dim3(nchunk, B), dim3(256), lds, s, (const float*)x, partial_ws, HW, C, ppc);
    else
        LAUNCH(gn_partial_kernel<bf16>, dim3(nchunk, B), dim3(256), lds, s, (const bf16*)x, partial_ws, HW, C, ppc);
    if (int rc = check_launch("gn_partial")) return rc;      // the next LAUNCH clears the error state: check before it
    LAUNCH(gn_finalize_kernel, dim3(G, B), dim3(64), 0, s, partial_ws, gamma, beta, scale, shift, HW, C, G, nchunk, eps);
    return check_launch("gn_stats");
}
extern "C" int ffn_gn_apply(void* stream, int dtype, const void* x, void* y, const float* scale, const float* shift, int B, int HW,
                            int C, int silu) {
    REQUIRE(dtype == FFN_F32 || dtype == FFN_BF16, "gn_apply: bad dtype");
    const int epc = dtype == FFN_F32 ? 4 : 8;
    REQUIRE(x && y && scale && shift && C % epc == 0 && aligned16(x) && aligned16(y), "gn_apply: bad arguments");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long nch = (long)B * HW * (C / epc);
    const int grid = grid_for(nch);
    if (silu & FFN_NORM_OUT_PAIR) {
        REQUIRE(dtype == FFN_F32, "gn_apply: pair output needs fp32 input");
        static const int wide = [] { const char* e = getenv("FFN_PAIR8"); return e ? atoi(e) : 1; }();      // 0: the round-5 kernels (8-byte stores, libm SiLU)
        if (wide && C % 8 == 0) {
            const long n8 = (long)B * HW * (C / 8);
            if (silu & FFN_NORM_SILU) LAUNCH(gn_apply_pair8_kernel<true>, dim3(grid_for(n8)), dim3(256), 0, s, (const float*)x, (bf16*)y, scale, shift, n8, HW, C);
            else LAUNCH(gn_apply_pair8_kernel<false>, dim3(grid_for(n8)), dim3(256), 0, s, (const float*)x, (bf16*)y, scale, shift, n8, HW, C);
            return check_launch("gn_apply(pair, 8 wide)");
        }
        if (silu & FFN_NORM_SILU) LAUNCH((gn_apply_kernel<float, true, true>), dim3(grid), dim3(256), 0, s, (const float*)x, (float*)y, scale, shift, nch, HW, C);
        else LAUNCH((gn_apply_kernel<float, false, true>), dim3(grid), dim3(256), 0, s, (const float*)x, (float*)y, scale, shift, nch, HW, C);
        return check_launch("gn_apply(pair)");
    }
    if (dtype == FFN_F32) {
        if (silu) LAUNCH((gn_apply_kernel<float, true>), dim3(grid), dim3(256), 0, s, (const float*)x, (float*)y, scale, shift, nch, HW, C);
        else LAUNCH((gn_apply_kernel<float, false>), dim3(grid), dim3(256), 0, s, (const float*)x, (float*)y, scale, shift, nch, HW, C);
    } else {
        if (silu) LAUNCH((gn_apply_kernel<bf16, true>), dim3(grid), dim3(256), 0, s, (const bf16*)x, (bf16*)y, scale, shift, nch, HW, C);
        else LAUNCH((gn_apply_kernel<bf16, false>), dim3(grid), dim3(256), 0, s, (const bf16*)x, (bf16*)y, scale, shift, nch, HW, C);
    }
    return check_launch("gn_apply");
}
extern "C" int ffn_layernorm(void* stream, int dtype, const void* x, void* y, const float* gamma, const float* beta, int M, int C,
                             float eps) {
    REQUIRE(dtype == FFN_F32 || dtype == FFN_BF16, "layernorm: bad dtype");
    const int epc = dtype == FFN_F32 ? 4 : 8;
    REQUIRE(x && y && gamma && beta && C % epc == 0 && aligned16(x) && aligned16(y), "layernorm: bad arguments");
    const int cch = C / epc;
    REQUIRE(cch <= 64 * 6, "layernorm: C=%d too large", C);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int grid = (M + 3) / 4;
    if (dtype == FFN_F32) {
        if (cch <= 128) LAUNCH((layernorm_kernel<float, 2>), dim3(grid), dim3(256), 0, s, (const float*)x, (float*)y, gamma, beta, M, C, eps);
        else LAUNCH((layernorm_kernel<float, 6>), dim3(grid), dim3(256), 0, s, (const float*)x, (float*)y, gamma, beta, M, C, eps);
    } else {
        if (cch <= 128) LAUNCH((layernorm_kernel<bf16, 2>), dim3(grid), dim3(256), 0, s, (const bf16*)x, (bf16*)y, gamma, beta, M, C, eps);
        else LAUNCH((layernorm_kernel<bf16, 6>), dim3(grid), dim3(256), 0, s, (const bf16*)x, (bf16*)y, gamma, beta, M, C, eps);
    }
    return check_launch("layernorm");
}
extern "C" int ffn_layernorm_pair(void* stream, const float* x, void* y, const float* gamma, const float* beta, int M, int C, float eps) {
    REQUIRE(x && y && gamma && beta && C % 4 == 0 && aligned16(x) && aligned16(y), "layernorm_pair: bad arguments");
    const int cch = C / 4;
    REQUIRE(cch <= 64 * 6, "layernorm_pair: C=%d too large", C);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int grid = (M + 3) / 4;
    if (cch <= 128) LAUNCH((layernorm_kernel<float, 2, true>), dim3(grid), dim3(256), 0, s, x, (float*)y, gamma, beta, M, C, eps);
    else LAUNCH((layernorm_kernel<float, 6, true>), dim3(grid), dim3(256), 0, s, x, (float*)y, gamma, beta, M, C, eps);
    return check_launch("layernorm(pair)");
}
extern "C" int ffn_softmax_rows(void* stream, int dtype, const void* x, void* y, long M, int N, float scale) {
    REQUIRE(dtype == FFN_F32 || dtype == FFN_BF16, "softmax_rows: bad dtype");
    REQUIRE(x && y && M > 0 && N > 0, "softmax_rows: bad arguments");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == FFN_F32) LAUNCH(softmax_rows_kernel<float>, dim3((unsigned)M), dim3(256), 0, s, (const float*)x, (float*)y, N, scale);
    else LAUNCH(softmax_rows_kernel<bf16>, dim3((unsigned)M), dim3(256), 0, s, (const bf16*)x, (bf16*)y, N, scale);
    return check_launch("softmax_rows");
}

// ---- scheduler / guidance ---------------------------------------------------------------------------------------
extern "C" int ffn_cfg_masked(void* stream, const float* eps_u, const float* eps_c, const float* mask, float cfg, float* eps, long n,
                              int HW) {
    REQUIRE(eps_u && eps_c && eps && n > 0 && HW > 0, "cfg_masked: bad arguments");
    LAUNCH(cfg_masked_kernel, dim3(grid_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), eps_u, eps_c, mask, cfg, eps, n, HW);
    return check_launch("cfg_masked");
}
extern "C" int ffn_ddim_inv_step(void* stream, const float* eps, const float* x, float c_bt, float c_at, float c_an, float c_bn,
                                 float* x_next, float* pred_x0, long n) {
    REQUIRE(eps && x && x_next && n > 0, "ddim_inv_step: bad arguments");
    LAUNCH(ddim_inv_step_kernel, dim3(grid_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), eps, x, c_bt, c_at, c_an, c_bn, x_next, pred_x0, n);
    return check_launch("ddim_inv_step");
}
extern "C" int ffn_ddim_ctrl_step(void* stream, const ffn_ctrl_step_desc* d) {
    REQUIRE(d && d->eps && d->x && d->x_prev && d->m && d->om, "ddim_ctrl_step: null pointer");
    REQUIRE(d->rows > 0 && d->rows <= 8 && d->CHW > 0 && d->HW > 0 && d->CHW % d->HW == 0, "ddim_ctrl_step: bad shape");
    LAUNCH(ddim_ctrl_step_kernel, dim3(grid_for((long)d->rows * d->CHW)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), *d);
    return check_launch("ddim_ctrl_step");
}

// ---- layout / misc -----------------------------------------------------------------------------------------------
extern "C" int ffn_pack_nchw(void* stream, int dtype, const ffn_pack_desc* d) {
    REQUIRE(d && d->src && d->dst && d->B > 0 && d->B <= 16 && d->CP >= d->Cl, "pack_nchw: bad arguments");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long n = (long)d->B * d->HW * d->CP;
    if (dtype == FFN_F32) LAUNCH(pack_nchw_kernel<float>, dim3(grid_for(n)), dim3(256), 0, s, *d);
    else if (dtype == FFN_BF16) LAUNCH(pack_nchw_kernel<bf16>, dim3(grid_for(n)), dim3(256), 0, s, *d);
    else return fail(FFN_EINVAL, "pack_nchw: bad dtype");
    return check_launch("pack_nchw");
}
extern "C" int ffn_nhwc_to_nchw_f32(void* stream, const float* src, float* dst, int B, int HW, int C, int ld) {
    REQUIRE(src && dst && B > 0 && HW > 0 && C > 0 && ld >= C, "nhwc_to_nchw_f32: bad arguments");
    LAUNCH(nhwc_to_nchw_f32_kernel, dim3(grid_for((long)B * HW * C)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, dst, B, HW, C, ld);
    return check_launch("nhwc_to_nchw_f32");
}
extern "C" int ffn_concat(void* stream, int dtype, const void* a, const void* b, void* out, long rows, int C1, int C2) {
    REQUIRE(dtype == FFN_F32 || dtype == FFN_BF16, "concat: bad dtype");
    const int epc = dtype == FFN_F32 ? 4 : 8;
    REQUIRE(b && out && C1 % epc == 0 && C2 % epc == 0 && aligned16(a) && aligned16(b) && aligned16(out), "concat: bad arguments");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long n = rows * ((a ? C1 + C2 : C2) / epc);     // a == NULL: out[:, :C1] is already in place, only b moves
    if (dtype == FFN_F32) LAUNCH(concat_kernel<float>, dim3(grid_for(n)), dim3(256), 0, s, (const float*)a, (const float*)b, (float*)out, rows, C1, C2);
    else LAUNCH(concat_kernel<bf16>, dim3(grid_for(n)), dim3(256), 0, s, (const bf16*)a, (const bf16*)b, (bf16*)out, rows, C1, C2);
    return check_launch("concat");
}
extern "C" int ffn_timestep_embed(void* stream, int dtype, const float* t_dev, const float* freq, void* out, int B, int half, int flip) {
    REQUIRE(t_dev && freq && out && B > 0 && half > 0, "timestep_embed: bad arguments");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int n = B * 2 * half;
    if (dtype == FFN_F32) LAUNCH(timestep_embed_kernel<float>, dim3((n + 255) / 256), dim3(256), 0, s, t_dev, freq, (float*)out, B, half, flip);
    else if (dtype == FFN_BF16) LAUNCH(timestep_embed_kernel<bf16>, dim3((n + 255) / 256), dim3(256), 0, s, t_dev, freq, (bf16*)out, B, half, flip);
    else return fail(FFN_EINVAL, "timestep_embed: bad dtype");
    return check_launch("timestep_embed");
}
extern "C" int ffn_transpose(void* stream, int dtype, const void* src, void* dst, int B, int R, int C, int ld_src, int ld_dst) {
    REQUIRE(src && dst && B > 0 && R > 0 && C > 0, "transpose: bad arguments");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((C + 31) / 32, (R + 31) / 32, B);
    if (dtype == FFN_F32) LAUNCH(transpose_kernel<float>, grid, dim3(256), 0, s, (const float*)src, (float*)dst, R, C, ld_src, ld_dst);
    else if (dtype == FFN_BF16) LAUNCH(transpose_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)src, (bf16*)dst, R, C, ld_src, ld_dst);
    else return fail(FFN_EINVAL, "transpose: bad dtype");
    return check_launch("transpose");
}
extern "C" int ffn_cast(void* stream, int src_dtype, int dst_dtype, const void* src, void* dst, long n) {
    REQUIRE(src && dst && n > 0, "cast: bad arguments");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int grid = grid_for(n);
    if (src_dtype == FFN_F32 && dst_dtype == FFN_BF16) LAUNCH((cast_kernel<float, bf16>), dim3(grid), dim3(256), 0, s, (const float*)src, (bf16*)dst, n);
    else if (src_dtype == FFN_BF16 && dst_dtype == FFN_F32) LAUNCH((cast_kernel<bf16, float>), dim3(grid), dim3(256), 0, s, (const bf16*)src, (float*)dst, n);
    else if (src_dtype == FFN_F32 && dst_dtype == FFN_F32) LAUNCH((cast_kernel<float, float>), dim3(grid), dim3(256), 0, s, (const float*)src, (float*)dst, n);
    else if (src_dtype == FFN_BF16 && dst_dtype == FFN_BF16) LAUNCH((cast_kernel<bf16, bf16>), dim3(grid), dim3(256), 0, s, (const bf16*)src, (bf16*)dst, n);
    else return fail(FFN_EINVAL, "cast: bad dtypes %d -> %d", src_dtype, dst_dtype);
    return check_launch("cast");
}
extern "C" int ffn_image_to_nhwc(void* stream, int dtype, const uint8_t* img, void* dst, long npix, int CP) {
    REQUIRE(img && dst && npix > 0 && CP >= 3, "image_to_nhwc: bad arguments");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int grid = grid_for(npix * CP);
    if (dtype == FFN_F32) LAUNCH(image_to_nhwc_kernel<float>, dim3(grid), dim3(256), 0, s, img, (float*)dst, npix, CP);
    else if (dtype == FFN_BF16) LAUNCH(image_to_nhwc_kernel<bf16>, dim3(grid), dim3(256), 0, s, img, (bf16*)dst, npix, CP);
    else return fail(FFN_EINVAL, "image_to_nhwc: bad dtype");
    return check_launch("image_to_nhwc");
}
extern "C" int ffn_nhwc_to_image(void* stream, int dtype, const void* src, float* dst, int B, int HW, int ld) {
    REQUIRE(src && dst && B > 0 && HW > 0 && ld >= 3, "nhwc_to_image: bad arguments");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int grid = grid_for((long)B * 3 * HW);
    if (dtype == FFN_F32) LAUNCH(nhwc_to_image_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)src, dst, B, HW, ld);
    else if (dtype == FFN_BF16) LAUNCH(nhwc_to_image_kernel<bf16>, dim3(grid), dim3(256), 0, s, (const bf16*)src, dst, B, HW, ld);
    else return fail(FFN_EINVAL, "nhwc_to_image: bad dtype");
    return check_launch("nhwc_to_image");
}

// ---- point-cloud warp of the 3-D front end (splat.h) ----------------------------------------------------------------
extern "C" int ffn_splat_lift(void* stream, const float* depth, const int* idx, float* pts, int n, int W, int H, float fx, float fy) {
    REQUIRE(depth && idx && pts, "splat_lift: null operand");
    REQUIRE(n > 0 && W > 0 && H > 0 && (long)W * H < (1l << 31) && fx != 0.f && fy != 0.f, "splat_lift: bad shape / focal length");
    LAUNCH(splat_lift_kernel, dim3(grid_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), depth, idx, pts, n, W, H, fx, fy);
    return check_launch("splat_lift");
}
extern "C" int ffn_splat_project(void* stream, const float* pts, float* proj, int n, const ffn_splat_xform* x) {
    REQUIRE(pts && proj && x, "splat_project: null operand");
    REQUIRE(n > 0, "splat_project: empty cloud");
    SplatXform X;
    for (int i = 0; i < 3; ++i) { X.c[i] = x->center[i]; X.t[i] = x->translate[i]; X.s[i] = x->scale[i]; }
    for (int i = 0; i < 9; ++i) X.R[i] = x->rotate[i];
    REQUIRE(x->tan_half_fov > 0.f, "splat_project: tan_half_fov must be positive");
    X.inv_tan = 1.f / x->tan_half_fov;
    LAUNCH(splat_project_kernel, dim3(grid_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pts, proj, n, X);
    return check_launch("splat_project");
}
extern "C" int ffn_splat_bin(void* stream, int fill, const float* proj, int n, float radius, int W, int H, int* counts, const int* offs, int* list) {
    REQUIRE(proj && counts, "splat_bin: null operand");
    REQUIRE(!fill || (offs && list), "splat_bin: the filling pass needs the tile offsets and the list");
    REQUIRE(n > 0 && W > 0 && H > 0 && radius > 0.f, "splat_bin: bad shape / radius");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (fill) LAUNCH(splat_bin_kernel<true>, dim3(grid_for(n)), dim3(256), 0, s, proj, n, radius, W, H, counts, offs, list);
    else LAUNCH(splat_bin_kernel<false>, dim3(grid_for(n)), dim3(256), 0, s, proj, n, radius, W, H, counts, offs, list);
    return check_launch("splat_bin");
}
extern "C" int ffn_splat_render(void* stream, const float* proj, const float* rgb, const int* offs, const int* list, float radius, int K, int W, int H,
                                float* image, int* idx_sum, uint8_t* covered) {
    REQUIRE(proj && rgb && offs && list && image && idx_sum && covered, "splat_render: null operand");
    REQUIRE(K >= 1 && K <= 32, "splat_render: points per pixel K=%d out of range (1 .. 32)", K);
    REQUIRE(W > 0 && H > 0 && radius > 0.f, "splat_render: bad shape / radius");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int tiles = ((W + 15) / 16) * ((H + 15) / 16);
    if (K <= 8) LAUNCH(splat_render_kernel<8>, dim3(tiles), dim3(256), 0, s, proj, rgb, offs, list, radius, K, W, H, image, idx_sum, covered);
    else if (K <= 16) LAUNCH(splat_render_kernel<16>, dim3(tiles), dim3(256), 0, s, proj, rgb, offs, list, radius, K, W, H, image, idx_sum, covered);
    else LAUNCH(splat_render_kernel<32>, dim3(tiles), dim3(256), 0, s, proj, rgb, offs, list, radius, K, W, H, image, idx_sum, covered);
    return check_launch("splat_render");
}
