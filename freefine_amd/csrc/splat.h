// Point-cloud lift -> rigid transform -> perspective projection -> z-sorted top-K disc splat with alpha compositing: the 3-D coarse edit of
// the GeoBench-3D front end, i.e. the arithmetic of IntegratedP3DTransRasterBlendingFull
// (/root/reference/src/utils/geo_utils.py:427-528), whose renderer is pytorch3d's PointsRasterizer + AlphaCompositor (pytorch3d is not in
// the image: PARITY UNPINNED -- the semantics below restate pytorch3d's published rasterize_points / alpha_composite kernels).
//
//   lift        (geo_utils.py:436-457): masked pixel (i, j) with depth z -> X = -(i - W/2) z / fx, Y = -(j - H/2) z / fy, Z = z
//               (the sign flip is the "open-cv world -> pytorch3d world" line :455)
//   transform   (:459-466, 343-378, 399-425): P' = ((P - c + t) R) * s with c = mean of the lifted points, row-vector convention,
//               R = Rx Ry Rz of the XYZ Euler angles, t the translation made absolute by the cloud's extent
//   camera      (:478-481): FoVPerspectiveCameras(R = I, T = c, fov = 60 deg): view = P' + c, ndc.xy = view.xy / (view.z tan 30deg), depth = view.z
//   rasterize   (:482-491): pixel (row, col) has NDC centre (xf, yf) = (ndc(W-1-col), ndc(H-1-row)) (+X left, +Y up); point p covers the pixel
//               iff view.z >= 0 and (xf-px)^2 + (yf-py)^2 < radius^2; the K covering points of smallest depth are kept, sorted by depth
//               (ties by point index)
//   composite   (:499-507): weights w_k = 1 - d2_k / radius^2, out = sum_k w_k prod_{j<k} (1 - w_j) rgb_k, background 0
//
// Layout: the splat is tile based (16 x 16 pixels per workgroup): a counting pass and a filling pass bin every point into the tiles its
// disc's bounding box touches (two launches around one exclusive scan), then one workgroup per tile streams its list through LDS in chunks
// of 256 points while each lane keeps the top-K of ITS pixel in registers (fully unrolled compare-exchange chain: no dynamic register
// indexing).  HBM-bound in the points (16 B each, read once per touched tile); the compare chain is the issue cost.
#pragma once
#include "common.h"

struct SplatXform {
    float c[3], t[3], R[9], s[3];
    float inv_tan;            // 1 / tan(fov / 2)
};

__device__ __forceinline__ float splat_pix_to_ndc(int i, int S1, int S2) {       // pytorch3d PixToNonSquareNdc
    float range = 2.0f;
    if (S1 > S2) range = ((float)S1 / (float)S2) * range;
    const float offset = range * 0.5f;
    return -offset + (range * (float)i + offset) / (float)S1;
}

// pts[n] = lifted point of masked pixel idx[n] (flat j * W + i)
__global__ __launch_bounds__(256) void splat_lift_kernel(const float* __restrict__ depth, const int* __restrict__ idx, float* __restrict__ pts, int n,
                                                         int W, int H, float fx, float fy) {
    for (int p = blockIdx.x * 256 + threadIdx.x; p < n; p += gridDim.x * 256) {
        const int f = idx[p], j = f / W, i = f - j * W;
        const float z = depth[f];
        const float x = ((float)i - (float)W * 0.5f) * z / fx, y = ((float)j - (float)H * 0.5f) * z / fy;
        *reinterpret_cast<f32x4*>(pts + 4l * p) = f32x4{-x, -y, z, 0.f};
    }
}

// proj[n] = (x_ndc, y_ndc, z_view, 0)
__global__ __launch_bounds__(256) void splat_project_kernel(const float* __restrict__ pts, float* __restrict__ proj, int n, const SplatXform X) {
    for (int p = blockIdx.x * 256 + threadIdx.x; p < n; p += gridDim.x * 256) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(pts + 4l * p);
        const float q0 = a[0] - X.c[0] + X.t[0], q1 = a[1] - X.c[1] + X.t[1], q2 = a[2] - X.c[2] + X.t[2];
        float v0 = (q0 * X.R[0] + q1 * X.R[3] + q2 * X.R[6]) * X.s[0] + X.c[0];
        float v1 = (q0 * X.R[1] + q1 * X.R[4] + q2 * X.R[7]) * X.s[1] + X.c[1];
        float v2 = (q0 * X.R[2] + q1 * X.R[5] + q2 * X.R[8]) * X.s[2] + X.c[2];
        *reinterpret_cast<f32x4*>(proj + 4l * p) = f32x4{v0 * X.inv_tan / v2, v1 * X.inv_tan / v2, v2, 0.f};
    }
}

// bounding box (in tiles of 16 x 16 pixels) of the disc of a projected point; false: nothing to draw
__device__ __forceinline__ bool splat_tile_box(const f32x4 a, float radius, int W, int H, int& tx0, int& tx1, int& ty0, int& ty1) {
    if (!(a[2] >= 0.f) || !(fabsf(a[0]) < 1e30f) || !(fabsf(a[1]) < 1e30f)) return false;       // behind the camera / not finite
    float rx = 2.f, ry = 2.f;
    if (W > H) rx = ((float)W / (float)H) * 2.f;
    if (H > W) ry = ((float)H / (float)W) * 2.f;
    const float ox = rx * 0.5f, oy = ry * 0.5f;
    // xf(u) = -ox + (rx u + ox) / W with u = W - 1 - col  ->  u = ((xf + ox) W - ox) / rx
    const float u_lo = ((a[0] - radius + ox) * (float)W - ox) / rx, u_hi = ((a[0] + radius + ox) * (float)W - ox) / rx;
    const float v_lo = ((a[1] - radius + oy) * (float)H - oy) / ry, v_hi = ((a[1] + radius + oy) * (float)H - oy) / ry;
    if (u_hi < -1.f || u_lo > (float)W || v_hi < -1.f || v_lo > (float)H) return false;
    int c0 = W - 1 - (int)floorf(fminf(u_hi, (float)W)) - 1, c1 = W - 1 - (int)ceilf(fmaxf(u_lo, -1.f)) + 1;
    int r0 = H - 1 - (int)floorf(fminf(v_hi, (float)H)) - 1, r1 = H - 1 - (int)ceilf(fmaxf(v_lo, -1.f)) + 1;
    c0 = c0 < 0 ? 0 : c0; r0 = r0 < 0 ? 0 : r0;
    c1 = c1 > W - 1 ? W - 1 : c1; r1 = r1 > H - 1 ? H - 1 : r1;
    if (c0 > c1 || r0 > r1) return false;
    tx0 = c0 >> 4; tx1 = c1 >> 4; ty0 = r0 >> 4; ty1 = r1 >> 4;
    return true;
}

// FILL = false: counts[tile] += 1 per touched tile;  FILL = true: list[offs[tile] + cursor[tile]++] = p
template <bool FILL>
__global__ __launch_bounds__(256) void splat_bin_kernel(const float* __restrict__ proj, int n, float radius, int W, int H, int* __restrict__ counts,
                                                        const int* __restrict__ offs, int* __restrict__ list) {
    const int TW = (W + 15) >> 4;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < n; p += gridDim.x * 256) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(proj + 4l * p);
        int tx0, tx1, ty0, ty1;
        if (!splat_tile_box(a, radius, W, H, tx0, tx1, ty0, ty1)) continue;
        for (int ty = ty0; ty <= ty1; ++ty)
            for (int tx = tx0; tx <= tx1; ++tx) {
                const int t = ty * TW + tx;
                const int pos = atomicAdd(counts + t, 1);
                if (FILL) list[offs[t] + pos] = p;
            }
    }
}

// one workgroup per tile, one lane per pixel
template <int KMAX>
__global__ __launch_bounds__(256) void splat_render_kernel(const float* __restrict__ proj, const float* __restrict__ rgb, const int* __restrict__ offs,
                                                           const int* __restrict__ list, float radius, int K, int W, int H,
                                                           float* __restrict__ image, int* __restrict__ idx_sum, uint8_t* __restrict__ covered) {
    __shared__ f32x4 s_pt[256];
    const int TW = (W + 15) >> 4;
    const int tile = blockIdx.x, ty = tile / TW, tx = tile - ty * TW;
    const int row = ty * 16 + (threadIdx.x >> 4), col = tx * 16 + (threadIdx.x & 15);
    const bool live = row < H && col < W;
    const float xf = splat_pix_to_ndc(W - 1 - col, W, H), yf = splat_pix_to_ndc(H - 1 - row, H, W);
    const float r2 = radius * radius;
    float qz[KMAX], qd[KMAX];
    int qi[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        qz[k] = __builtin_inff();
        qd[k] = 0.f;
        qi[k] = 0x7fffffff;
    }
    const int beg = offs[tile], end = offs[tile + 1];
    for (int base = beg; base < end; base += 256) {
        const int m = end - base < 256 ? end - base : 256;
        __syncthreads();
        if ((int)threadIdx.x < m) {
            const int p = list[base + threadIdx.x];
            f32x4 a = *reinterpret_cast<const f32x4*>(proj + 4l * p);
            a[3] = __int_as_float(p);
            s_pt[threadIdx.x] = a;
        }
        __syncthreads();
        if (!live) continue;
        for (int e = 0; e < m; ++e) {
            const f32x4 a = s_pt[e];
            const float dx = xf - a[0], dy = yf - a[1], d2 = dx * dx + dy * dy;
            if (!(d2 < r2)) continue;
            float cz = a[2], cd = d2;
            int ci = __float_as_int(a[3]);
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                if (k < K) {
                    const bool lt = cz < qz[k] || (cz == qz[k] && ci < qi[k]);
                    const float tz = lt ? qz[k] : cz, td = lt ? qd[k] : cd;
                    const int ti = lt ? qi[k] : ci;
                    qz[k] = lt ? cz : qz[k];
                    qd[k] = lt ? cd : qd[k];
                    qi[k] = lt ? ci : qi[k];
                    cz = tz; cd = td; ci = ti;
                }
            }
        }
    }
    if (!live) return;
    float cum = 1.f, o0 = 0.f, o1 = 0.f, o2 = 0.f;
    int sum = 0;
    bool any = false;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
            if (qi[k] == 0x7fffffff) {
                sum -= 1;
            } else {
                const float w = 1.f - qd[k] / r2;
                const float* f = rgb + 3l * qi[k];
                o0 += cum * w * f[0];
                o1 += cum * w * f[1];
                o2 += cum * w * f[2];
                cum *= 1.f - w;
                sum += qi[k];
                any = true;
            }
        }
    }
    const long o = (long)row * W + col;
    image[3 * o] = o0;
    image[3 * o + 1] = o1;
    image[3 * o + 2] = o2;
    idx_sum[o] = sum;
    covered[o] = any ? 1 : 0;
}
