// ops.hip -- the bandwidth-bound glue of the MADNet hot path for gfx950: feature warping,
// TF1-legacy bilinear resize (+scale/relu/crop fusions), reflect padding, the photometric
// SSIM+L1 reprojection loss with its disparity gradient, validation metrics, momentum update.
// Every kernel is a coalesced NHWC stream with 16-byte accesses where the layout allows;
// reductions are two-stage (wave shuffles -> per-block partial -> one finishing block) so the
// results are deterministic.
#include "mh_common.h"

namespace {

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// ------------------------------------------------------------------------------------------
// MadNet._linear_warping (Nets/MadNet.py:400-436) with coords from _build_indeces (:378-397)
// ------------------------------------------------------------------------------------------
struct WarpArgs {
    const float* img; const float* u; const float* g; float* out; float* dimg; float* du;
    int img_ld, out_ld, g_ld, dimg_ld, acc_u;
    int B, H, W, C;
    int64_t total;
};

__global__ __launch_bounds__(256) void warp_fwd_kernel(WarpArgs p) {
    const int C4 = p.C >> 2;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < p.total; q += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(q % C4);
        const int64_t pix = q / C4;
        const int x = (int)(pix % p.W);
        const int64_t rowbase = pix - x;
        const float cx = (float)x + p.u[pix];
        const float x0 = floorf(cx), x1 = x0 + 1.0f;
        const float xmax = (float)(p.W - 1);
        const float x0s = clampf(x0, 0.f, xmax), x1s = clampf(x1, 0.f, xmax);
        const float w0 = (x1 - cx) * (x0 == x0s ? 1.f : 0.f);
        const float w1 = (cx - x0) * (x1 == x1s ? 1.f : 0.f);
        const float4 a = *reinterpret_cast<const float4*>(p.img + (rowbase + (int)x0s) * p.img_ld + c4 * 4);
        const float4 b = *reinterpret_cast<const float4*>(p.img + (rowbase + (int)x1s) * p.img_ld + c4 * 4);
        float4 o;
        o.x = w0 * a.x + w1 * b.x; o.y = w0 * a.y + w1 * b.y;
        o.z = w0 * a.z + w1 * b.z; o.w = w0 * a.w + w1 * b.w;
        *reinterpret_cast<float4*>(p.out + pix * p.out_ld + c4 * 4) = o;
    }
}

// one wave = 64 lanes = (64/LPP) pixels x LPP channel lanes; du reduced with shuffles.
template <int LPP>
__global__ __launch_bounds__(256) void warp_bwd_kernel(WarpArgs p) {
    const int C4 = p.C >> 2;
    constexpr int PPB = 256 / LPP;
    const int sub = threadIdx.x % LPP;
    const int64_t npix = (int64_t)p.B * p.H * p.W;
    const int64_t nit = (npix + PPB - 1) / PPB;
    for (int64_t it = blockIdx.x; it < nit; it += gridDim.x) {
        const int64_t pix = it * PPB + threadIdx.x / LPP;
        const bool live = pix < npix;
        const int64_t pp = live ? pix : 0;
        const int x = (int)(pp % p.W);
        const int64_t rowbase = pp - x;
        const float cx = (float)x + p.u[pp];
        const float x0 = floorf(cx), x1 = x0 + 1.0f;
        const float xmax = (float)(p.W - 1);
        const float x0s = clampf(x0, 0.f, xmax), x1s = clampf(x1, 0.f, xmax);
        const float m0 = (x0 == x0s) ? 1.f : 0.f, m1 = (x1 == x1s) ? 1.f : 0.f;
        const float w0 = (x1 - cx) * m0, w1 = (cx - x0) * m1;
        const int i0 = (int)x0s, i1 = (int)x1s;
        float dcx = 0.f;
        for (int c4 = sub; c4 < C4; c4 += LPP) {
            if (!live) continue;
            const float4 gv = *reinterpret_cast<const float4*>(p.g + pp * p.g_ld + c4 * 4);
            if (p.dimg) {
                float* d0 = p.dimg + (rowbase + i0) * p.dimg_ld + c4 * 4;
                float* d1 = p.dimg + (rowbase + i1) * p.dimg_ld + c4 * 4;
                if (w0 != 0.f) { mh_atomic_add(d0 + 0, w0 * gv.x); mh_atomic_add(d0 + 1, w0 * gv.y); mh_atomic_add(d0 + 2, w0 * gv.z); mh_atomic_add(d0 + 3, w0 * gv.w); }
                if (w1 != 0.f) { mh_atomic_add(d1 + 0, w1 * gv.x); mh_atomic_add(d1 + 1, w1 * gv.y); mh_atomic_add(d1 + 2, w1 * gv.z); mh_atomic_add(d1 + 3, w1 * gv.w); }
            }
            if (p.du) {
                const float4 a = *reinterpret_cast<const float4*>(p.img + (rowbase + i0) * p.img_ld + c4 * 4);
                const float4 b = *reinterpret_cast<const float4*>(p.img + (rowbase + i1) * p.img_ld + c4 * 4);
                dcx += gv.x * (m1 * b.x - m0 * a.x) + gv.y * (m1 * b.y - m0 * a.y) +
                       gv.z * (m1 * b.z - m0 * a.z) + gv.w * (m1 * b.w - m0 * a.w);
            }
        }
        if (p.du) {
#pragma unroll
            for (int o = LPP >> 1; o > 0; o >>= 1) dcx += __shfl_xor(dcx, o);
            if (live && sub == 0) p.du[pp] = p.acc_u ? p.du[pp] + dcx : dcx;
        }
    }
}

// ------------------------------------------------------------------------------------------
// TF1 legacy bilinear resize + scale / relu / crop fusions
// ------------------------------------------------------------------------------------------
struct ResizeArgs {
    const float* in; const float* g; float* out; float* din;
    int B, Hi, Wi, Hr, Wr, cy, cx, Ho, Wo, mode, accumulate;
    float mul, sy, sx;   // sy = (float)Hi/(float)Hr
};

__device__ __forceinline__ void interp1(int i, float scale, int n, int& lo, int& hi, float& t) {
    const float src = (float)i * scale;
    lo = (int)src;
    hi = min(lo + 1, n - 1);
    t = src - (float)lo;
}

// (no mul+add contraction in the interpolation arithmetic: the resize kernels and the fused level front end of corr.hip then
//  produce bit-identical values, and the CPU emulator build -- no fma on plain x86-64 -- matches the GPU)
__device__ __forceinline__ float bilerp(const float* img, int Wi, int y0, int y1, float ty, int x0, int x1, float tx,
                                        float mul, bool relu_in) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    float tl = img[(int64_t)y0 * Wi + x0], tr = img[(int64_t)y0 * Wi + x1];
    float bl = img[(int64_t)y1 * Wi + x0], br = img[(int64_t)y1 * Wi + x1];
    if (relu_in) {
        tl = fmaxf(tl * mul, 0.f); tr = fmaxf(tr * mul, 0.f);
        bl = fmaxf(bl * mul, 0.f); br = fmaxf(br * mul, 0.f);
    }
    const float top = tl + (tr - tl) * tx;
    const float bot = bl + (br - bl) * tx;
    return top + (bot - top) * ty;
}

__global__ __launch_bounds__(256) void resize_fwd_kernel(ResizeArgs p) {
    const int64_t total = (int64_t)p.B * p.Ho * p.Wo;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
        const int x = (int)(q % p.Wo);
        const int64_t t2 = q / p.Wo;
        const int y = (int)(t2 % p.Ho);
        const int b = (int)(t2 / p.Ho);
        int y0, y1, x0, x1; float ty, tx;
        interp1(y + p.cy, p.sy, p.Hi, y0, y1, ty);
        interp1(x + p.cx, p.sx, p.Wi, x0, x1, tx);
        const float* img = p.in + (int64_t)b * p.Hi * p.Wi;
        float v = bilerp(img, p.Wi, y0, y1, ty, x0, x1, tx, p.mul, p.mode == 1);
        if (p.mode == 0) v *= p.mul;
        else if (p.mode == 2) v = fmaxf(v * p.mul, 0.f);
        p.out[q] = v;
    }
}

// gather form of ResizeBilinearGrad: one lane per INPUT pixel, loops the output pixels whose
// lower/upper source index hits it (exactly the forward's float index arithmetic).
// LPP lanes share one input pixel and split the candidate output ROWS (an up-scaling by 4 gives an input pixel ~8 x 8 candidate
// output pixels, each needing the forward's bilinear taps again for the relu mask of mode 2: serial per lane that was 21 us for the
// 96x320 -> 375x1242 head at the start of the backward pass); partial sums meet in a shuffle butterfly.
template <int LPP>
__global__ __launch_bounds__(256) void resize_bwd_kernel(ResizeArgs p) {
    const int64_t total = (int64_t)p.B * p.Hi * p.Wi;
    const float isy = 1.0f / p.sy, isx = 1.0f / p.sx;
    const int sub = LPP > 1 ? (int)(threadIdx.x % LPP) : 0;
    const int64_t nit = (total * LPP + 255) / 256;              // every lane of a wave runs the same trip count (shuffles below)
    for (int64_t it = blockIdx.x; it < nit; it += gridDim.x) {
        const int64_t q0 = (it * 256 + threadIdx.x) / LPP;
        const bool live = q0 < total;
        const int64_t q = live ? q0 : 0;
        const int sx = (int)(q % p.Wi);
        const int64_t t2 = q / p.Wi;
        const int sy = (int)(t2 % p.Hi);
        const int b = (int)(t2 / p.Hi);
        const float* img = p.in + (int64_t)b * p.Hi * p.Wi;
        const float* gimg = p.g + (int64_t)b * p.Ho * p.Wo;
        const int ya = max(p.cy, (int)floorf((float)(sy - 1) * isy) - 1);
        const int yb = min(p.cy + p.Ho - 1, (int)ceilf((float)(sy + 1) * isy) + 1);
        const int xa = max(p.cx, (int)floorf((float)(sx - 1) * isx) - 1);
        const int xb = min(p.cx + p.Wo - 1, (int)ceilf((float)(sx + 1) * isx) + 1);
        float acc = 0.f;
        for (int Y = ya + sub; Y <= yb && live; Y += LPP) {
            int y0, y1; float ty;
            interp1(Y, p.sy, p.Hi, y0, y1, ty);
            const float wy = (y0 == sy ? 1.0f - ty : 0.f) + (y1 == sy ? ty : 0.f);
            if (wy == 0.f) continue;
            for (int X = xa; X <= xb; ++X) {
                int x0, x1; float tx;
                interp1(X, p.sx, p.Wi, x0, x1, tx);
                const float wx = (x0 == sx ? 1.0f - tx : 0.f) + (x1 == sx ? tx : 0.f);
                if (wx == 0.f) continue;
                float gv = gimg[(int64_t)(Y - p.cy) * p.Wo + (X - p.cx)];
                if (p.mode == 2) {
                    const float z = bilerp(img, p.Wi, y0, y1, ty, x0, x1, tx, p.mul, false) * p.mul;
                    if (!(z > 0.f)) gv = 0.f;
                }
                acc += gv * wy * wx;
            }
        }
#pragma unroll
        for (int o = LPP >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (live && sub == 0) {
            acc *= p.mul;
            if (p.mode == 1 && !(img[(int64_t)sy * p.Wi + sx] * p.mul > 0.f)) acc = 0.f;
            p.din[q] = p.accumulate ? p.din[q] + acc : acc;
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward front end of a disparity head in ONE launch (mh_head_bwd): the head's output gradient dV is assembled from what feeds it --
//   kind 0: the coordinate gradient du of the next finer level through the gradient of the x2 legacy resize (u = resize(V) * 20 / 2^k,
//           Nets/MadNet.py:274): exactly resize_bwd_kernel<1>'s gather, same summation order;
//   kind 1: two per-pixel addends with their own pixel strides (level 2: dfinal and the disparity channel of the context input's gradient)
// -- written out (fp32 + the bf16 shadow the streamed filter gradient of the head reads), and pushed through the head's 3x3 Cin -> 1 conv
// backwards (conv_k1_dgrad_kernel's arithmetic) with the leaky mask of the layer below, again with the shadow.  Two launches (three at
// level 2) of ~7 us each on the critical chain before.  A workgroup owns 4 x 32 head pixels; dV of the tile + a one-pixel halo goes through LDS.
// ------------------------------------------------------------------------------------------
struct HeadBwdArgs {
    ResizeArgs rz;                   // kind 0: rz.g = du (fine), Hi x Wi = the head's size
    const float* a1; const float* a2; int a1_ld, a2_ld;      // kind 1
    float* dV; unsigned short* dV_sh; int dV_sh_ld;
    const float* w; float* dx; const float* mask_ref; unsigned short* dx_sh;
    int N, dx_ld, mask_ld, dx_sh_ld, acc_dx, kind;
    float mask_alpha;
    int tiles_x, tiles_y;
};
#define HB_TH 4
#define HB_TW 32
__device__ __forceinline__ float head_dv_from_du(const ResizeArgs& p, int b, int sy, int sx, bool exact2) {
    const float* gimg = p.g + (int64_t)b * p.Ho * p.Wo;
    if (exact2) {
        // Hr = 2 Hi, no crop: fine row Y = 2 sy sits on the coarse row (weight 1), its neighbours half way (0.5; the last fine row clamps
        // onto the last coarse row: 1).  Same candidates, weights and order as the general walk below -- bit-identical, without its index search.
        // (the nine candidates are loaded unconditionally from clamped addresses and the out-of-range ones dropped by a select: loads inside `continue`
        //  branches were nine memory round trips one after the other -- round 4)
        float v[9], wgt[9];
        bool ok[9];
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int Y = 2 * sy + dy;
            const bool oky = (unsigned)Y < (unsigned)p.Ho;
            const int Yc = oky ? Y : 2 * sy;
            const float wy = dy == 0 ? 1.0f : ((dy == 1 && sy == p.Hi - 1) ? 1.0f : 0.5f);
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int X = 2 * sx + dx;
                const bool okx = (unsigned)X < (unsigned)p.Wo;
                const int Xc = okx ? X : 2 * sx;
                const float wx = dx == 0 ? 1.0f : ((dx == 1 && sx == p.Wi - 1) ? 1.0f : 0.5f);
                const int k = (dy + 1) * 3 + dx + 1;
                v[k] = gimg[(int64_t)Yc * p.Wo + Xc] * wy;
                wgt[k] = wx; ok[k] = oky && okx;
            }
        }
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) acc += ok[k] ? v[k] * wgt[k] : 0.f;
        return acc * p.mul;
    }
    const float isy = 1.0f / p.sy, isx = 1.0f / p.sx;
    const int ya = max(p.cy, (int)floorf((float)(sy - 1) * isy) - 1);
    const int yb = min(p.cy + p.Ho - 1, (int)ceilf((float)(sy + 1) * isy) + 1);
    const int xa = max(p.cx, (int)floorf((float)(sx - 1) * isx) - 1);
    const int xb = min(p.cx + p.Wo - 1, (int)ceilf((float)(sx + 1) * isx) + 1);
    float acc = 0.f;
    for (int Y = ya; Y <= yb; ++Y) {
        int y0, y1; float ty;
        interp1(Y, p.sy, p.Hi, y0, y1, ty);
        const float wy = (y0 == sy ? 1.0f - ty : 0.f) + (y1 == sy ? ty : 0.f);
        if (wy == 0.f) continue;
        for (int X = xa; X <= xb; ++X) {
            int x0, x1; float tx;
            interp1(X, p.sx, p.Wi, x0, x1, tx);
            const float wx = (x0 == sx ? 1.0f - tx : 0.f) + (x1 == sx ? tx : 0.f);
            if (wx == 0.f) continue;
            acc += gimg[(int64_t)(Y - p.cy) * p.Wo + (X - p.cx)] * wy * wx;
        }
    }
    return acc * p.mul;
}
__global__ __launch_bounds__(256) void head_bwd_kernel(HeadBwdArgs p) {
    __shared__ float sV[(HB_TH + 2) * (HB_TW + 2)];
    __shared__ __attribute__((aligned(16))) float sW[9 * 64];
    const int tid = threadIdx.x;
    const int H = p.rz.Hi, W = p.rz.Wi;
    const int tx = blockIdx.x % p.tiles_x;
    const int t2 = blockIdx.x / p.tiles_x;
    const int ty = t2 % p.tiles_y, b = t2 / p.tiles_y;
    const int y0 = ty * HB_TH, x0 = tx * HB_TW;
    const bool exact2 = p.kind == 0 && p.rz.Hr == 2 * H && p.rz.Wr == 2 * W && p.rz.cy == 0 && p.rz.cx == 0 && p.rz.Ho == p.rz.Hr && p.rz.Wo == p.rz.Wr;
    for (int i = tid; i < 9 * p.N; i += 256) sW[i] = p.w[i];
    if (tid < (HB_TH + 2) * (HB_TW + 2)) {
        const int ly = tid / (HB_TW + 2), lx = tid - ly * (HB_TW + 2);
        const int y = y0 - 1 + ly, x = x0 - 1 + lx;
        float v = 0.f;
        if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
            const int64_t pix = ((int64_t)b * H + y) * W + x;
            if (p.kind == 0) v = head_dv_from_du(p.rz, b, y, x, exact2);
            else v = (p.a1 ? p.a1[pix * p.a1_ld] : 0.f) + (p.a2 ? p.a2[pix * p.a2_ld] : 0.f);
            if (ly >= 1 && ly <= HB_TH && lx >= 1 && lx <= HB_TW) {
                p.dV[pix] = v;
                if (p.dV_sh) p.dV_sh[pix * p.dV_sh_ld] = (unsigned short)mh_pack_bf16(v, 0.f);
            }
        }
        sV[tid] = v;
    }
    __syncthreads();
    // dx[y][x][n] = mask(sum_taps dV[y + 1 - ky][x + 1 - kx] * w[ky][kx][n]) (+ old)
    const int G4 = p.N >> 2;
    if (G4 <= 16 && (p.acc_dx || p.mask_ref)) {
        // up to 8 items per thread: the old map / the mask of every item requested before the first is consumed (a load per loop iteration behind a store
        // to the same buffer cannot be hoisted by the compiler: four dependent round trips for a 32-channel head)
        constexpr int MAXIT = HB_TH * HB_TW * 16 / 256;
        float4 oldv[MAXIT], mkv[MAXIT];
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int i = tid + it * 256;
            const int g = i % G4, px = i / G4;
            const int iy = px / HB_TW, ix = px - iy * HB_TW;
            const int y = y0 + iy, x = x0 + ix;
            const bool live = i < HB_TH * HB_TW * G4 && y < H && x < W;
            const int64_t m = ((int64_t)b * H + (live ? y : y0)) * W + (live ? x : x0);
            oldv[it] = (live && p.acc_dx) ? *reinterpret_cast<const float4*>(p.dx + m * p.dx_ld + g * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            mkv[it] = (live && p.mask_ref) ? *reinterpret_cast<const float4*>(p.mask_ref + m * p.mask_ld + g * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
        }
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int i = tid + it * 256;
            if (i >= HB_TH * HB_TW * G4) break;
            const int g = i % G4, px = i / G4;
            const int iy = px / HB_TW, ix = px - iy * HB_TW;
            const int y = y0 + iy, x = x0 + ix;
            if (y >= H || x >= W) continue;
            const int n = g * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ky = t / 3, kx = t - ky * 3;
                const float z = sV[(iy + 2 - ky) * (HB_TW + 2) + ix + 2 - kx];
                const float4 w = *reinterpret_cast<const float4*>(sW + t * p.N + n);
                v.x += z * w.x; v.y += z * w.y; v.z += z * w.z; v.w += z * w.w;
            }
            const int64_t m = ((int64_t)b * H + y) * W + x;
            float* dst = p.dx + m * p.dx_ld + n;
            if (p.acc_dx) { const float4 o = oldv[it]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            if (p.mask_ref) {
                const float4 mk = mkv[it];
                v.x *= mk.x > 0.f ? 1.0f : p.mask_alpha; v.y *= mk.y > 0.f ? 1.0f : p.mask_alpha;
                v.z *= mk.z > 0.f ? 1.0f : p.mask_alpha; v.w *= mk.w > 0.f ? 1.0f : p.mask_alpha;
            }
            *reinterpret_cast<float4*>(dst) = v;
            if (p.dx_sh) *reinterpret_cast<uint2*>(p.dx_sh + m * p.dx_sh_ld + n) = make_uint2(mh_pack_bf16(v.x, v.y), mh_pack_bf16(v.z, v.w));
        }
        return;
    }
    for (int i = tid; i < HB_TH * HB_TW * G4; i += 256) {
        const int g = i % G4, px = i / G4;
        const int iy = px / HB_TW, ix = px - iy * HB_TW;
        const int y = y0 + iy, x = x0 + ix;
        if (y >= H || x >= W) continue;
        const int n = g * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ky = t / 3, kx = t - ky * 3;
            const float z = sV[(iy + 2 - ky) * (HB_TW + 2) + ix + 2 - kx];
            const float4 w = *reinterpret_cast<const float4*>(sW + t * p.N + n);
            v.x += z * w.x; v.y += z * w.y; v.z += z * w.z; v.w += z * w.w;
        }
        const int64_t m = ((int64_t)b * H + y) * W + x;
        float* dst = p.dx + m * p.dx_ld + n;
        if (p.acc_dx) { const float4 o = *reinterpret_cast<const float4*>(dst); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        if (p.mask_ref) {
            const float4 mk = *reinterpret_cast<const float4*>(p.mask_ref + m * p.mask_ld + n);
            v.x *= mk.x > 0.f ? 1.0f : p.mask_alpha; v.y *= mk.y > 0.f ? 1.0f : p.mask_alpha;
            v.z *= mk.z > 0.f ? 1.0f : p.mask_alpha; v.w *= mk.w > 0.f ? 1.0f : p.mask_alpha;
        }
        *reinterpret_cast<float4*>(dst) = v;
        if (p.dx_sh) *reinterpret_cast<uint2*>(p.dx_sh + m * p.dx_sh_ld + n) = make_uint2(mh_pack_bf16(v.x, v.y), mh_pack_bf16(v.z, v.w));
    }
}

// ------------------------------------------------------------------------------------------
// multi-channel TF1-legacy bilinear resize of NHWC images (Stereo_Online_Adaptation.scale_tensor :22-23 ->
// preprocessing.rescale_image :269-273 on the 3-channel frames when --reprojectionScale != 1) and its gradient
// ------------------------------------------------------------------------------------------
struct ResizeImgArgs { const float* in; const float* g; float* out; float* din; int B, Hi, Wi, C, Ho, Wo; float sy, sx; };

__global__ __launch_bounds__(256) void resize_image_fwd_kernel(ResizeImgArgs p) {
    const int64_t total = (int64_t)p.B * p.Ho * p.Wo * p.C;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
        const int c = (int)(q % p.C);
        int64_t t = q / p.C;
        const int x = (int)(t % p.Wo); t /= p.Wo;
        const int y = (int)(t % p.Ho);
        const int b = (int)(t / p.Ho);
        int y0, y1, x0, x1; float ty, tx;
        interp1(y, p.sy, p.Hi, y0, y1, ty);
        interp1(x, p.sx, p.Wi, x0, x1, tx);
        const float* img = p.in + (int64_t)b * p.Hi * p.Wi * p.C + c;
        const float tl = img[((int64_t)y0 * p.Wi + x0) * p.C], tr = img[((int64_t)y0 * p.Wi + x1) * p.C];
        const float bl = img[((int64_t)y1 * p.Wi + x0) * p.C], br = img[((int64_t)y1 * p.Wi + x1) * p.C];
        const float top = tl + (tr - tl) * tx, bot = bl + (br - bl) * tx;
        p.out[q] = top + (bot - top) * ty;
    }
}

// gather form (one lane per INPUT element), same index arithmetic as the forward
__global__ __launch_bounds__(256) void resize_image_bwd_kernel(ResizeImgArgs p) {
    const int64_t total = (int64_t)p.B * p.Hi * p.Wi * p.C;
    const float isy = 1.0f / p.sy, isx = 1.0f / p.sx;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
        const int c = (int)(q % p.C);
        int64_t t = q / p.C;
        const int sx = (int)(t % p.Wi); t /= p.Wi;
        const int sy = (int)(t % p.Hi);
        const int b = (int)(t / p.Hi);
        const float* gimg = p.g + (int64_t)b * p.Ho * p.Wo * p.C + c;
        const int ya = max(0, (int)floorf((float)(sy - 1) * isy) - 1), yb = min(p.Ho - 1, (int)ceilf((float)(sy + 1) * isy) + 1);
        const int xa = max(0, (int)floorf((float)(sx - 1) * isx) - 1), xb = min(p.Wo - 1, (int)ceilf((float)(sx + 1) * isx) + 1);
        float acc = 0.f;
        for (int Y = ya; Y <= yb; ++Y) {
            int y0, y1; float ty;
            interp1(Y, p.sy, p.Hi, y0, y1, ty);
            const float wy = (y0 == sy ? 1.0f - ty : 0.f) + (y1 == sy ? ty : 0.f);
            if (wy == 0.f) continue;
            for (int X = xa; X <= xb; ++X) {
                int x0, x1; float tx;
                interp1(X, p.sx, p.Wi, x0, x1, tx);
                const float wx = (x0 == sx ? 1.0f - tx : 0.f) + (x1 == sx ? tx : 0.f);
                if (wx != 0.f) acc += gimg[((int64_t)Y * p.Wo + X) * p.C] * wy * wx;
            }
        }
        p.din[q] = acc;
    }
}

// ------------------------------------------------------------------------------------------
// preprocessing.bilinear_sampler (Data_utils/preprocessing.py:121-199), general form: out[b,y,x,:] = 4-tap bilinear sample
// of imgs[b] at coords[b,y,x] = (cx, cy).  Indices are CLAMPED to the border and the weights are NOT masked (the code's
// behaviour, not its docstring: SURVEY App. D.7); the flat gather index is computed in float32 like the reference (:170-187)
// and then cast.  Gradients: w.r.t. coords (d out / d cx = (wt_y0*(im10-im00) + wt_y1*(im11-im01)), likewise cy; floor has no
// gradient) and w.r.t. imgs (scatter, fp32 atomics into a pre-zeroed buffer).
// ------------------------------------------------------------------------------------------
struct SamplerArgs { const float* img; const float* coords; const float* g; float* out; float* dcoords; float* dimg;
                     int B, Hs, Ws, C, Ht, Wt; };

__device__ __forceinline__ void sampler_taps(const SamplerArgs& p, int b, float cx, float cy, int64_t (&idx)[4], float (&w)[4],
                                             float& wx0, float& wx1, float& wy0, float& wy1) {
    const float x0 = floorf(cx), x1 = x0 + 1.0f, y0 = floorf(cy), y1 = y0 + 1.0f;
    wx0 = x1 - cx; wx1 = cx - x0; wy0 = y1 - cy; wy1 = cy - y0;
    const float xm = (float)(p.Ws - 1), ym = (float)(p.Hs - 1);
    const float x0s = clampf(x0, 0.f, xm), x1s = clampf(x1, 0.f, xm), y0s = clampf(y0, 0.f, ym), y1s = clampf(y1, 0.f, ym);
    const float dim2 = (float)p.Ws, base = (float)b * (float)(p.Ws * p.Hs);
    const float by0 = base + y0s * dim2, by1 = base + y1s * dim2;
    idx[0] = (int64_t)(int)(x0s + by0); idx[1] = (int64_t)(int)(x0s + by1);        // im00, im01
    idx[2] = (int64_t)(int)(x1s + by0); idx[3] = (int64_t)(int)(x1s + by1);        // im10, im11
    w[0] = wx0 * wy0; w[1] = wx0 * wy1; w[2] = wx1 * wy0; w[3] = wx1 * wy1;
}

__global__ __launch_bounds__(256) void sampler_fwd_kernel(SamplerArgs p) {
    const int64_t total = (int64_t)p.B * p.Ht * p.Wt;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
        const int b = (int)(q / ((int64_t)p.Ht * p.Wt));
        int64_t idx[4]; float w[4], wx0, wx1, wy0, wy1;
        sampler_taps(p, b, p.coords[q * 2], p.coords[q * 2 + 1], idx, w, wx0, wx1, wy0, wy1);
        for (int c = 0; c < p.C; ++c) {
            // tf.add_n([w00*im00, w01*im01, w10*im10, w11*im11]): left-to-right sum
            float v = w[0] * p.img[idx[0] * p.C + c];
            v += w[1] * p.img[idx[1] * p.C + c];
            v += w[2] * p.img[idx[2] * p.C + c];
            v += w[3] * p.img[idx[3] * p.C + c];
            p.out[q * p.C + c] = v;
        }
    }
}

__global__ __launch_bounds__(256) void sampler_bwd_kernel(SamplerArgs p) {
    const int64_t total = (int64_t)p.B * p.Ht * p.Wt;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
        const int b = (int)(q / ((int64_t)p.Ht * p.Wt));
        int64_t idx[4]; float w[4], wx0, wx1, wy0, wy1;
        sampler_taps(p, b, p.coords[q * 2], p.coords[q * 2 + 1], idx, w, wx0, wx1, wy0, wy1);
        float gx = 0.f, gy = 0.f;
        for (int c = 0; c < p.C; ++c) {
            const float gv = p.g[q * p.C + c];
            const float i00 = p.img[idx[0] * p.C + c], i01 = p.img[idx[1] * p.C + c], i10 = p.img[idx[2] * p.C + c], i11 = p.img[idx[3] * p.C + c];
            // wt_x0 = x1 - cx (d/dcx = -1), wt_x1 = cx - x0 (+1); same for y
            gx += gv * (wy0 * (i10 - i00) + wy1 * (i11 - i01));
            gy += gv * (wx0 * (i01 - i00) + wx1 * (i11 - i10));
            if (p.dimg) {
                mh_atomic_add(p.dimg + idx[0] * p.C + c, gv * w[0]); mh_atomic_add(p.dimg + idx[1] * p.C + c, gv * w[1]);
                mh_atomic_add(p.dimg + idx[2] * p.C + c, gv * w[2]); mh_atomic_add(p.dimg + idx[3] * p.C + c, gv * w[3]);
            }
        }
        if (p.dcoords) { p.dcoords[q * 2] = gx; p.dcoords[q * 2 + 1] = gy; }
    }
}

// ------------------------------------------------------------------------------------------
// preprocessing.pad_image (REFLECT) + channel padding
// ------------------------------------------------------------------------------------------
struct PadArgs { const float* in; float* out; int B, H, W, C, Hp, Wp, pt, pl, out_ld; float div, sub; };

__global__ __launch_bounds__(256) void pad_reflect_kernel(PadArgs p) {
    const int64_t total = (int64_t)p.B * p.Hp * p.Wp;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
        const int x = (int)(q % p.Wp);
        const int64_t t2 = q / p.Wp;
        const int y = (int)(t2 % p.Hp);
        const int b = (int)(t2 / p.Hp);
        int sy = y - p.pt, sx = x - p.pl;
        sy = sy < 0 ? -sy : (sy >= p.H ? 2 * (p.H - 1) - sy : sy);
        sx = sx < 0 ? -sx : (sx >= p.W ? 2 * (p.W - 1) - sx : sx);
        const float* src = p.in + (((int64_t)b * p.H + sy) * p.W + sx) * p.C;
        float* dst = p.out + q * p.out_ld;
        for (int c = 0; c < p.out_ld; ++c) dst[c] = c < p.C ? (p.div == 1.0f ? src[c] : src[c] / p.div) - p.sub : 0.f;
    }
}

// ------------------------------------------------------------------------------------------
// reprojection loss (mean_SSIM_L1 of warp_image(right, disp) vs left)
// ------------------------------------------------------------------------------------------
struct LossArgs {
    const float* left; const float* right; const float* disp;
    float* rep; float* drep; float* coef; float* part1; float* part2; float* result; float* ddisp;
    int B, H, W, nblk1, nblk2;
    float grad_scale;
};

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = mh_wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// One launch for the maps and the gradient: a workgroup owns a 16 x 32 pixel tile and works through LDS on the tile + a 2-pixel halo
// (a pixel's gradient needs the SSIM coefficients of the <= 9 windows that contain it, a window needs the warped image at its 9 pixels):
//   A  warp right by the disparity at the 20 x 36 halo pixels (bilinear_sampler: clamped indices, un-masked weights; rows are integral
//      so only the y0 taps carry weight) -> LDS; d rep / d x for the tile's own pixels -> LDS; L1 partial sum over the tile's pixels;
//   B  SSIM over the 18 x 34 3x3 VALID windows that touch the tile -> per-window derivative coefficients in LDS
//        d map / d x_p = alpha + beta*y_p + gamma*x_p   for the 9 pixels p of the window
//      (windows outside the image: zeros); SSIM partial sum over the windows whose top-left corner is a pixel of the tile;
//   C  d loss / d disp[p] = - sum_ch (d loss / d rep[p,ch]) * drep[p,ch].
// (Three launches with the maps in HBM before: 39 us per step on the critical path at 375 x 1242; per-pixel arithmetic and summation order of
// the gradient unchanged.)
#define LT_H 16
#define LT_W 32
#define LT_HH (LT_H + 4)
#define LT_HW (LT_W + 4)
#define LT_WH (LT_H + 2)
#define LT_WW (LT_W + 2)
__global__ __launch_bounds__(256) void loss_tile_kernel(LossArgs p, int tiles_x, int tiles_y, int with_grad) {
    __shared__ float red[4];
    __shared__ float s_rep[LT_HH * LT_HW * 3];
    __shared__ float s_y[LT_HH * LT_HW * 3];
    __shared__ float s_drep[LT_H * LT_W * 3];
    __shared__ float s_coef[LT_WH * LT_WW * 9];
    const int tid = threadIdx.x;
    const int tx = blockIdx.x % tiles_x;
    const int t2 = blockIdx.x / tiles_x;
    const int ty = t2 % tiles_y, b = t2 / tiles_y;
    const int y0t = ty * LT_H, x0t = tx * LT_W;
    const float s = 1.0f / 256.0f;
    const float xmax = (float)(p.W - 1);
    // ---- A -------------------------------------------------------------------------------------
    // Round 5: the three halo pixels of a thread go through the phase TOGETHER -- three disparity loads in flight, then all 27 image loads (range-checked buffer
    // loads: a pixel outside the image = an out-of-range offset = zeros, no branch around a load).  The loop this replaces waited for disp, then for its nine
    // image values, once per pixel: six dependent memory round trips at the head of a launch that sits on the forward chain of every mode.
    float l1 = 0.f;
    {
        constexpr int NI = (LT_HH * LT_HW + 255) / 256;          // 3
        const unsigned npx = (unsigned)p.B * (unsigned)p.H * (unsigned)p.W;
        const __amdgpu_buffer_rsrc_t rs_d = mh_make_rsrc(p.disp, npx * 4u);
        const __amdgpu_buffer_rsrc_t rs_l = mh_make_rsrc(p.left, npx * 12u);
        const __amdgpu_buffer_rsrc_t rs_r = mh_make_rsrc(p.right, npx * 12u);
        int qv[NI], rbv[NI];
        bool inb[NI];
        float dv[NI];
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int i = tid + u * 256;
            const int ly = i / LT_HW, lx = i - ly * LT_HW;
            const int y = y0t - 2 + ly, x = x0t - 2 + lx;
            inb[u] = i < LT_HH * LT_HW && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
            rbv[u] = (b * p.H + y) * p.W;
            qv[u] = rbv[u] + x;
            dv[u] = mh_buf_load1(rs_d, inb[u] ? qv[u] * 4 : MH_OOB);
        }
        float av[NI][3], bv[NI][3], lv[NI][3], w0v[NI], w1v[NI];
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int i = tid + u * 256;
            const int ly = i / LT_HW, lx = i - ly * LT_HW;
            const int x = x0t - 2 + lx;
            const float cx = (float)x - dv[u];
            const float xf0 = floorf(cx), xf1 = xf0 + 1.0f;
            w0v[u] = xf1 - cx; w1v[u] = cx - xf0;
            const int i0 = (int)clampf(xf0, 0.f, xmax), i1 = (int)clampf(xf1, 0.f, xmax);
            int o0 = (rbv[u] + i0) * 12, o1 = (rbv[u] + i1) * 12, ol = qv[u] * 12;
            MH_KEEP_VGPR(o0); MH_KEEP_VGPR(o1); MH_KEEP_VGPR(ol);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                av[u][c] = mh_buf_load1(rs_r, inb[u] ? o0 + 4 * c : MH_OOB);
                bv[u][c] = mh_buf_load1(rs_r, inb[u] ? o1 + 4 * c : MH_OOB);
                lv[u][c] = mh_buf_load1(rs_l, inb[u] ? ol + 4 * c : MH_OOB);
            }
        }
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int i = tid + u * 256;
            if (i >= LT_HH * LT_HW) break;
            const int ly = i / LT_HW, lx = i - ly * LT_HW;
            float r0v = 0.f, r1v = 0.f, r2v = 0.f, y0v = 0.f, y1v = 0.f, y2v = 0.f;
            if (inb[u]) {
                const float a0 = av[u][0] * s, a1 = av[u][1] * s, a2 = av[u][2] * s;
                const float b0 = bv[u][0] * s, b1 = bv[u][1] * s, b2 = bv[u][2] * s;
                const float w0 = w0v[u], w1 = w1v[u];
                r0v = w0 * a0 + w1 * b0; r1v = w0 * a1 + w1 * b1; r2v = w0 * a2 + w1 * b2;
                y0v = lv[u][0] * s; y1v = lv[u][1] * s; y2v = lv[u][2] * s;
                const int iy = ly - 2, ix = lx - 2;
                if ((unsigned)iy < (unsigned)LT_H && (unsigned)ix < (unsigned)LT_W) {
                    float* d = s_drep + (iy * LT_W + ix) * 3;
                    d[0] = b0 - a0; d[1] = b1 - a1; d[2] = b2 - a2;
                    l1 += fabsf(r0v - y0v) + fabsf(r1v - y1v) + fabsf(r2v - y2v);
                }
            }
            s_rep[i * 3 + 0] = r0v; s_rep[i * 3 + 1] = r1v; s_rep[i * 3 + 2] = r2v;
            s_y[i * 3 + 0] = y0v; s_y[i * 3 + 1] = y1v; s_y[i * 3 + 2] = y2v;
        }
    }
    __syncthreads();
    // ---- B -------------------------------------------------------------------------------------
    const int Hw = p.H - 2, Ww = p.W - 2;
    float ssum = 0.f;
    for (int i = tid; i < LT_WH * LT_WW; i += 256) {
        const int ly = i / LT_WW, lx = i - ly * LT_WW;
        const int wy = y0t - 2 + ly, wx = x0t - 2 + lx;              // window (wy, wx) = halo pixels (ly .. ly+2, lx .. lx+2)
        float co[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if ((unsigned)wy < (unsigned)Hw && (unsigned)wx < (unsigned)Ww) {
            const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
            float sx[3] = {0, 0, 0}, sy[3] = {0, 0, 0}, sxx[3] = {0, 0, 0}, syy[3] = {0, 0, 0}, sxy[3] = {0, 0, 0};
            for (int dy = 0; dy < 3; ++dy)
                for (int dx = 0; dx < 3; ++dx) {
                    const int h = ((ly + dy) * LT_HW + lx + dx) * 3;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float xv = s_rep[h + c], yv = s_y[h + c];
                        sx[c] += xv; sy[c] += yv; sxx[c] += xv * xv; syy[c] += yv * yv; sxy[c] += xv * yv;
                    }
                }
            const bool own = ly >= 2 && lx >= 2 && ly < 2 + LT_H && lx < 2 + LT_W;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float inv9 = 1.0f / 9.0f;
                const float mx = sx[c] * inv9, my = sy[c] * inv9;
                const float vx = sxx[c] * inv9 - mx * mx, vy = syy[c] * inv9 - my * my, vxy = sxy[c] * inv9 - mx * my;
                const float n1 = 2.f * mx * my + C1, n2 = 2.f * vxy + C2;
                const float d1 = mx * mx + my * my + C1, d2 = vx + vy + C2;
                const float S = (n1 * n2) / (d1 * d2);
                const float mraw = (1.0f - S) * 0.5f;
                if (own) ssum += clampf(mraw, 0.f, 1.f);
                // tf.clip_by_value passes the gradient iff 0 <= x <= 1 ; d map = -0.5 dS
                const float pass = (mraw >= 0.f && mraw <= 1.f) ? -0.5f * (2.0f / 9.0f) : 0.f;
                const float idd = 1.0f / (d1 * d2);
                const float beta = n1 * idd;
                const float gamma = -S / d2;
                const float alpha = (my * n2 - n1 * my) * idd - S * (mx * d2 - d1 * mx) * idd;
                co[c * 3 + 0] = pass * alpha; co[c * 3 + 1] = pass * beta; co[c * 3 + 2] = pass * gamma;
            }
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) s_coef[i * 9 + k] = co[k];
    }
    const float tot1 = block_sum(l1, red);          // (its barriers also publish s_coef)
    const float tot2 = block_sum(ssum, red);
    if (tid == 0) { p.part1[blockIdx.x] = tot1; p.part2[blockIdx.x] = tot2; }
    if (!with_grad) return;
    // ---- C -------------------------------------------------------------------------------------
    const float k_ssim = 0.85f / ((float)p.B * (float)Hw * (float)Ww * 3.0f);
    const float k_l1 = 0.15f / ((float)p.B * (float)p.H * (float)p.W * 3.0f);
    for (int i = tid; i < LT_H * LT_W; i += 256) {
        const int iy = i / LT_W, ix = i - iy * LT_W;
        const int y = y0t + iy, x = x0t + ix;
        if (y >= p.H || x >= p.W) continue;
        float A[3] = {0, 0, 0}, Bc[3] = {0, 0, 0}, Gc[3] = {0, 0, 0};
        // windows (y-2 .. y, x-2 .. x) = local windows (iy .. iy+2, ix .. ix+2); the ones outside the image hold zeros
        for (int dy = 0; dy < 3; ++dy)
            for (int dx = 0; dx < 3; ++dx) {
                const float* cp = s_coef + ((iy + dy) * LT_WW + ix + dx) * 9;
                A[0] += cp[0]; Bc[0] += cp[1]; Gc[0] += cp[2];
                A[1] += cp[3]; Bc[1] += cp[4]; Gc[1] += cp[5];
                A[2] += cp[6]; Bc[2] += cp[7]; Gc[2] += cp[8];
            }
        const int h = ((iy + 2) * LT_HW + ix + 2) * 3;
        float gd = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float xv = s_rep[h + c], yv = s_y[h + c];
            const float diff = xv - yv;
            const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
            const float grep = k_ssim * (A[c] + Bc[c] * yv + Gc[c] * xv) + k_l1 * sgn;
            gd -= grep * s_drep[i * 3 + c];
        }
        p.ddisp[((int64_t)b * p.H + y) * p.W + x] = gd * p.grad_scale;
    }
}

__global__ __launch_bounds__(256) void loss_final_kernel(LossArgs p) {
    __shared__ double red[256];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < p.nblk1; i += 256) a += (double)p.part1[i];
    for (int i = threadIdx.x; i < p.nblk2; i += 256) b += (double)p.part2[i];
    red[threadIdx.x] = a; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    const double l1 = red[0]; __syncthreads();
    red[threadIdx.x] = b; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) {
        const double n1 = (double)p.B * p.H * p.W * 3.0;
        const double n2 = (double)p.B * (p.H - 2) * (p.W - 2) * 3.0;
        const double ms = red[0] / n2, ml = l1 / n1;
        p.result[0] = (float)(0.85 * ms + 0.15 * ml);
        p.result[1] = (float)ms;
        p.result[2] = (float)ml;
    }
}

// ------------------------------------------------------------------------------------------
// validation metrics (Stereo_Online_Adaptation.py:74-82)
// ------------------------------------------------------------------------------------------
struct MetArgs { const float* disp; const float* gt; float* part; float* result; int64_t total; int nblk; float th; };

__global__ __launch_bounds__(256) void metrics_kernel(MetArgs p) {
    __shared__ float red[4];
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float e = 0.f, bad = 0.f, v = 0.f;
    if (q < p.total) {
        const float gt = p.gt[q];
        v = (gt == 0.f) ? 0.f : 1.f;
        e = fabsf(p.disp[q] - gt) * v;
        bad = e > p.th ? 1.f : 0.f;
    }
    const float se = block_sum(e, red);
    const float sb = block_sum(bad, red);
    const float sv = block_sum(v, red);
    if (threadIdx.x == 0) { p.part[blockIdx.x * 3 + 0] = se; p.part[blockIdx.x * 3 + 1] = sb; p.part[blockIdx.x * 3 + 2] = sv; }
}

__global__ __launch_bounds__(256) void metrics_final_kernel(MetArgs p) {
    __shared__ double red[3][256];
    double a = 0, b = 0, c = 0;
    for (int i = threadIdx.x; i < p.nblk; i += 256) { a += p.part[i * 3]; b += p.part[i * 3 + 1]; c += p.part[i * 3 + 2]; }
    red[0][threadIdx.x] = a; red[1][threadIdx.x] = b; red[2][threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            red[0][threadIdx.x] += red[0][threadIdx.x + o];
            red[1][threadIdx.x] += red[1][threadIdx.x + o];
            red[2][threadIdx.x] += red[2][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        p.result[0] = (float)(red[0][0] / red[2][0]);
        p.result[1] = (float)(red[1][0] / red[2][0]);
        p.result[2] = (float)red[2][0];
    }
}

// ------------------------------------------------------------------------------------------
// proxy-label loss of the continual-adaptation variant: loss_factory.get_proxy_loss('mean_l1') (Losses/loss_factory.py:304-351)
//   valid = !(proxy <= 0 || proxy >= 192) ; loss = weight * sum(valid * |pred - proxy|) / sum(valid)
//   d loss / d pred = weight * valid * sign(pred - proxy) / sum(valid)       (tf.abs gradient: sign, 0 at 0)
// ------------------------------------------------------------------------------------------
// The supervised multi-scale loss of Train.py (loss_factory.get_supervised_loss('mean_l1'), Losses/loss_factory.py:256-302) is the
// same reduction with valid = !(target == 0 || target >= max_disp): `hi` / `zero_only` select the rule.
struct ProxyArgs { const float* pred; const float* proxy; float* part; float* result; float* dpred; int64_t total; int nblk; float weight, gs; float hi; int zero_only; };
__device__ __forceinline__ float proxy_valid(const ProxyArgs& p, float px) {
    const bool bad = (p.zero_only ? px == 0.f : px <= 0.f) || px >= p.hi;
    return bad ? 0.f : 1.f;
}

__global__ __launch_bounds__(256) void proxy_partial_kernel(ProxyArgs p) {
    __shared__ float red[4];
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float e = 0.f, v = 0.f;
    if (q < p.total) {
        const float px = p.proxy[q];
        v = proxy_valid(p, px);
        e = fabsf(p.pred[q] - px) * v;
    }
    const float se = block_sum(e, red);
    const float sv = block_sum(v, red);
    if (threadIdx.x == 0) { p.part[blockIdx.x * 2 + 0] = se; p.part[blockIdx.x * 2 + 1] = sv; }
}

__global__ __launch_bounds__(256) void proxy_final_kernel(ProxyArgs p) {
    __shared__ double red[2][256];
    double a = 0, c = 0;
    for (int i = threadIdx.x; i < p.nblk; i += 256) { a += p.part[i * 2]; c += p.part[i * 2 + 1]; }
    red[0][threadIdx.x] = a; red[1][threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        p.result[0] = (float)((double)p.weight * red[0][0] / red[1][0]);      // 0/0 = NaN like the TF graph
        p.result[1] = (float)red[1][0];
    }
}

__global__ __launch_bounds__(256) void proxy_grad_kernel(ProxyArgs p) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= p.total) return;
    const float px = p.proxy[q];
    const float v = proxy_valid(p, px);
    const float d = p.pred[q] - px;
    const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    p.dpred[q] = p.gs * p.weight * v * sgn / p.result[1];
}

// ------------------------------------------------------------------------------------------
// momentum / glue
// ------------------------------------------------------------------------------------------
// 16-byte form (all three buffers 16-byte aligned: the engines' flat buffers are): two loads in flight per lane, the launch sits behind the join
// at the very end of the step.  Same arithmetic per element.
__global__ __launch_bounds__(256) void momentum4_kernel(float4* var, float4* acc, const float4* g, int64_t n4, float lr, float mom, float gs) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 m = acc[i], gg = g[i];
        float4 v = var[i];
        float4 a;
        a.x = mom * m.x + gs * gg.x; a.y = mom * m.y + gs * gg.y; a.z = mom * m.z + gs * gg.z; a.w = mom * m.w + gs * gg.w;
        acc[i] = a;
        v.x -= lr * a.x; v.y -= lr * a.y; v.z -= lr * a.z; v.w -= lr * a.w;
        var[i] = v;
    }
}
__global__ __launch_bounds__(256) void momentum_kernel(float* var, float* acc, const float* g, int64_t n, float lr, float mom, float gs) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float a = mom * acc[i] + gs * g[i];
        acc[i] = a;
        var[i] -= lr * a;
    }
}

// tf.train.AdamOptimizer(lr, beta1) of Train.py:95 (TF 1.12 training/adam.py, ApplyAdam):
//   lr_t = lr * sqrt(1 - beta2_power) / (1 - beta1_power);  m += (g - m)(1 - b1);  v += (g^2 - v)(1 - b2);
//   var -= (m * lr_t) / (sqrt(v) + eps)         state = {beta1_power, beta2_power}, multiplied by b1 / b2 AFTER the update
__global__ __launch_bounds__(256) void adam_kernel(float* var, float* m, float* v, const float* g, int64_t n, const float* state,
                                                   float lr, float b1, float b2, float eps, float gs) {
    const float lr_t = lr * sqrtf(1.0f - state[1]) / (1.0f - state[0]);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float gi = gs * g[i];
        const float mi = m[i] + (gi - m[i]) * (1.0f - b1);          // the ApplyAdam functor's own form (core/kernels/training_ops.cc)
        const float vi = v[i] + (gi * gi - v[i]) * (1.0f - b2);
        m[i] = mi; v[i] = vi;
        var[i] -= (mi * lr_t) / (sqrtf(vi) + eps);
    }
}
__global__ void adam_advance_kernel(float* state, float b1, float b2) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { state[0] *= b1; state[1] *= b2; }
}

__global__ __launch_bounds__(256) void copy_channels_kernel(const float* src, int src_ld, float* dst, int dst_ld, int64_t npix,
                                                            int nch, float scale, int accumulate) {
    const int64_t total = npix * nch;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
        const int c = (int)(q % nch);
        const int64_t pix = q / nch;
        const float v = scale * src[pix * src_ld + c];
        float* d = dst + pix * dst_ld + c;
        *d = accumulate ? *d + v : v;
    }
}

__global__ __launch_bounds__(256) void leaky_bwd_kernel(float* dy, int dy_ld, const float* y, int y_ld, int64_t npix, int nch, float alpha) {
    const int64_t total = npix * nch;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
        const int c = (int)(q % nch);
        const int64_t pix = q / nch;
        if (!(y[pix * y_ld + c] > 0.f)) dy[pix * dy_ld + c] *= alpha;
    }
}

// column sums, coalesced along the channels: a wave row = 64/nchp pixels x nchp channels (nchp = power of two covering
// min(nch, 64)), blockIdx.y = 64-channel chunk; pixels strided over waves and workgroups; shuffle reduction over the
// pixels of a wave row, LDS over the 4 waves, one atomic per channel per workgroup.
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* dz, int dz_ld, int64_t npix, int nch, float* db, int nchp, float* ws) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int ppw = 64 / nchp;
    const int c = blockIdx.y * 64 + (lane & (nchp - 1));
    const int pl = lane / nchp;
    float v = 0.f;
    if (c < nch) {
        // eight loads in flight per lane (one per iteration left the launch latency bound: 15 - 30 dependent round trips on DispNet's 192x640 maps)
        const int64_t st = (int64_t)gridDim.x * 4 * ppw;
        int64_t q = ((int64_t)blockIdx.x * 4 + w) * ppw + pl;
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (; q + 7 * st < npix; q += 8 * st) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = dz[(q + u * st) * dz_ld + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += t[u];
        }
        for (; q < npix; q += st) a[0] += dz[q * dz_ld + c];
        v = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    for (int o = nchp; o < 64; o <<= 1) v += __shfl_xor(v, o);
    red[w][lane] = v;
    __syncthreads();
    if (w == 0 && pl == 0 && c < nch) {
        const float t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        // partial form (round 6): the workgroup's column sums go to ws[blockIdx.x][nch] with plain stores and mh_wgrad_reduce sums the workgroups in order
        if (ws) ws[(int64_t)blockIdx.x * nch + c] = t; else mh_atomic_add(db + c, t);
    }
}

__global__ __launch_bounds__(256) void fill_kernel(float* p, int64_t n, float v) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = v;
}

inline int grid_for(int64_t n, int cap = 256 * 16) {
    int64_t b = (n + 255) / 256;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" int mh_warp_fwd(const float* img, int32_t img_ld, const float* u, float* out, int32_t out_ld,
                           int32_t B, int32_t H, int32_t W, int32_t C, void* stream) {
    MH_REQUIRE(img && u && out, MH_ERR_ARG, "mh_warp_fwd: null argument");
    MH_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0, MH_ERR_ARG, "mh_warp_fwd: bad dimension");
    MH_REQUIRE(C % 4 == 0 && img_ld % 4 == 0 && out_ld % 4 == 0 && mh_aligned16(img) && mh_aligned16(out), MH_ERR_ALIGN,
               "mh_warp_fwd: C and lds must be multiples of 4, pointers 16-byte aligned");
    WarpArgs a{}; a.img = img; a.u = u; a.out = out; a.img_ld = img_ld; a.out_ld = out_ld;
    a.B = B; a.H = H; a.W = W; a.C = C; a.total = (int64_t)B * H * W * (C / 4);
    hipLaunchKernelGGL(warp_fwd_kernel, dim3(grid_for(a.total)), dim3(256), 0, (hipStream_t)stream, a);
    return mh_check_launch("warp_fwd");
}

extern "C" int mh_warp_bwd(const float* g, int32_t g_ld, const float* img, int32_t img_ld, const float* u,
                           float* dimg, int32_t dimg_ld, float* du, int32_t acc_u,
                           int32_t B, int32_t H, int32_t W, int32_t C, void* stream) {
    MH_REQUIRE(g && img && u && (dimg || du), MH_ERR_ARG, "mh_warp_bwd: null argument");
    MH_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0, MH_ERR_ARG, "mh_warp_bwd: bad dimension");
    MH_REQUIRE(C % 4 == 0 && img_ld % 4 == 0 && g_ld % 4 == 0 && mh_aligned16(img) && mh_aligned16(g), MH_ERR_ALIGN,
               "mh_warp_bwd: C and lds must be multiples of 4, pointers 16-byte aligned");
    WarpArgs a{}; a.g = g; a.img = img; a.u = u; a.dimg = dimg; a.du = du; a.acc_u = acc_u;
    a.g_ld = g_ld; a.img_ld = img_ld; a.dimg_ld = dimg_ld; a.B = B; a.H = H; a.W = W; a.C = C;
    const int C4 = C / 4;
    const int64_t npix = (int64_t)B * H * W;
    hipStream_t s = (hipStream_t)stream;
    if (C4 <= 4) hipLaunchKernelGGL((warp_bwd_kernel<4>), dim3(grid_for(npix * 4)), dim3(256), 0, s, a);
    else if (C4 <= 8) hipLaunchKernelGGL((warp_bwd_kernel<8>), dim3(grid_for(npix * 8)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((warp_bwd_kernel<16>), dim3(grid_for(npix * 16)), dim3(256), 0, s, a);
    return mh_check_launch("warp_bwd");
}

static int resize_args(ResizeArgs& a, int B, int Hi, int Wi, int Hr, int Wr, int cy, int cx, int Ho, int Wo, float mul, int mode) {
    MH_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && Hr > 0 && Wr > 0 && Ho > 0 && Wo > 0, MH_ERR_ARG, "mh_resize: bad dimension");
    MH_REQUIRE(cy >= 0 && cx >= 0 && cy + Ho <= Hr && cx + Wo <= Wr, MH_ERR_ARG, "mh_resize: crop outside the resized image");
    MH_REQUIRE(mode >= 0 && mode <= 2, MH_ERR_ARG, "mh_resize: mode must be 0,1,2");
    a.B = B; a.Hi = Hi; a.Wi = Wi; a.Hr = Hr; a.Wr = Wr; a.cy = cy; a.cx = cx; a.Ho = Ho; a.Wo = Wo; a.mode = mode;
    a.mul = mul; a.sy = (float)Hi / (float)Hr; a.sx = (float)Wi / (float)Wr;
    return 0;
}

extern "C" int mh_resize_fwd(const float* in, float* out, int32_t B, int32_t Hi, int32_t Wi, int32_t Hr, int32_t Wr,
                             int32_t cy, int32_t cx, int32_t Ho, int32_t Wo, float mul, int32_t mode, void* stream) {
    MH_REQUIRE(in && out, MH_ERR_ARG, "mh_resize_fwd: null argument");
    ResizeArgs a{}; a.in = in; a.out = out;
    if (int e = resize_args(a, B, Hi, Wi, Hr, Wr, cy, cx, Ho, Wo, mul, mode)) return e;
    hipLaunchKernelGGL(resize_fwd_kernel, dim3(grid_for((int64_t)B * Ho * Wo)), dim3(256), 0, (hipStream_t)stream, a);
    return mh_check_launch("resize_fwd");
}

// uint8 frames (what the camera / the PNG decoder delivers) -> float32 0..255 (what the graph reads): the cast tf.data does on the
// host (Data_utils/data_reader.py:98 tf.cast(..., tf.float32)) moved behind the PCIe copy, which then carries 1 byte per value
__global__ __launch_bounds__(256) void u8_to_f32_kernel(const unsigned char* __restrict__ in, float* __restrict__ out, int64_t n) {
    const int64_t q4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (q4 + 4 <= n && ((((uintptr_t)in) | ((uintptr_t)out)) & 3u) == 0 && (((uintptr_t)out) & 15u) == 0) {
        const unsigned v = *reinterpret_cast<const unsigned*>(in + q4);
        *reinterpret_cast<float4*>(out + q4) = make_float4((float)(v & 0xffu), (float)((v >> 8) & 0xffu), (float)((v >> 16) & 0xffu), (float)(v >> 24));
    } else {
        for (int64_t q = q4; q < q4 + 4 && q < n; ++q) out[q] = (float)in[q];
    }
}

// ---- the step's frames through a table the HOST rewrites between two replays of a captured step (mh_fetch_inputs) ---------------------------------------------------
// A captured step reads its frames at fixed addresses; a prefetcher delivers frame t in slot t % depth.  Instead of three copy launches in front of every replay, the
// step's first node reads the slot's addresses from a small table (device-visible host memory, or device memory the host updates) and moves / casts the frames into the
// fixed input buffers itself.  Entry k: src[k] = NULL or == dst[k]: nothing to do; u8[k]: the source holds 8-bit values.
struct FetchArgs { const mh_input_table* tab; float* dst[MH_FETCH_MAX]; int64_t n[MH_FETCH_MAX]; int64_t q0[MH_FETCH_MAX + 1]; };
__global__ __launch_bounds__(256) void fetch_inputs_kernel(FetchArgs a) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;          // one quad of floats
    int k = 0;
#pragma unroll
    for (int j = 1; j < MH_FETCH_MAX; ++j) k += q >= a.q0[j] ? 1 : 0;
    if (q >= a.q0[MH_FETCH_MAX]) return;
    const void* src = a.tab->src[k];
    float* dst = a.dst[k];
    if (!src || src == (const void*)dst) return;
    const int64_t e0 = (q - a.q0[k]) * 4, n = a.n[k];
    if (a.tab->u8[k]) {
        const unsigned char* in = (const unsigned char*)src;
        if (e0 + 4 <= n && ((((uintptr_t)in) & 3u) | (((uintptr_t)dst) & 15u)) == 0) {
            const unsigned v = *reinterpret_cast<const unsigned*>(in + e0);
            *reinterpret_cast<float4*>(dst + e0) = make_float4((float)(v & 0xffu), (float)((v >> 8) & 0xffu), (float)((v >> 16) & 0xffu), (float)(v >> 24));
        } else {
            for (int64_t e = e0; e < e0 + 4 && e < n; ++e) dst[e] = (float)in[e];
        }
    } else {
        const float* in = (const float*)src;
        if (e0 + 4 <= n && ((((uintptr_t)in) | ((uintptr_t)dst)) & 15u) == 0) *reinterpret_cast<float4*>(dst + e0) = *reinterpret_cast<const float4*>(in + e0);
        else for (int64_t e = e0; e < e0 + 4 && e < n; ++e) dst[e] = in[e];
    }
}

extern "C" int mh_fetch_inputs(const mh_input_table* table, float* const* dst, const int64_t* n, int32_t count, void* stream) {
    MH_REQUIRE(table && dst && n && count >= 1 && count <= MH_FETCH_MAX, MH_ERR_ARG, "mh_fetch_inputs: bad argument");
    FetchArgs a{};
    a.tab = table;
    int64_t q = 0;
    for (int k = 0; k < MH_FETCH_MAX; ++k) {
        a.q0[k] = q;
        if (k < count) {
            MH_REQUIRE(dst[k] && n[k] >= 0, MH_ERR_ARG, "mh_fetch_inputs: null destination / negative count");
            a.dst[k] = dst[k]; a.n[k] = n[k];
            q += (n[k] + 3) / 4;
        }
    }
    a.q0[MH_FETCH_MAX] = q;
    if (q == 0) return 0;
    MH_REQUIRE((q + 255) / 256 < (1ll << 31), MH_ERR_UNSUPPORTED, "mh_fetch_inputs: too many elements");
    hipLaunchKernelGGL(fetch_inputs_kernel, dim3((unsigned)((q + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return mh_check_launch("fetch_inputs");
}

// the device-side address of a page-locked host allocation (what a kernel dereferences to read a table the host keeps writing)
extern "C" int mh_host_device_pointer(void* host, void** device) {
    MH_REQUIRE(host && device, MH_ERR_ARG, "mh_host_device_pointer: null argument");
    hipError_t e = hipHostGetDevicePointer(device, host, 0);
    if (e != hipSuccess) { mh_set_error("mh_host_device_pointer: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}

extern "C" int mh_u8_to_f32(const uint8_t* in, float* out, int64_t n, void* stream) {
    MH_REQUIRE(in && out && n > 0, MH_ERR_ARG, "mh_u8_to_f32: bad argument");
    MH_REQUIRE((n + 1023) / 1024 < (1ll << 31), MH_ERR_UNSUPPORTED, "mh_u8_to_f32: too many elements");
    hipLaunchKernelGGL(u8_to_f32_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, in, out, n);
    return mh_check_launch("u8_to_f32");
}

extern "C" int mh_resize_image_fwd(const float* in, float* out, int32_t B, int32_t Hi, int32_t Wi, int32_t C, int32_t Ho, int32_t Wo,
                                   void* stream) {
    MH_REQUIRE(in && out && B > 0 && Hi > 0 && Wi > 0 && C > 0 && Ho > 0 && Wo > 0, MH_ERR_ARG, "mh_resize_image_fwd: bad argument");
    ResizeImgArgs a{in, nullptr, out, nullptr, B, Hi, Wi, C, Ho, Wo, (float)Hi / (float)Ho, (float)Wi / (float)Wo};
    hipLaunchKernelGGL(resize_image_fwd_kernel, dim3(grid_for((int64_t)B * Ho * Wo * C)), dim3(256), 0, (hipStream_t)stream, a);
    return mh_check_launch("resize_image_fwd");
}

extern "C" int mh_resize_image_bwd(const float* g, float* din, int32_t B, int32_t Hi, int32_t Wi, int32_t C, int32_t Ho, int32_t Wo,
                                   void* stream) {
    MH_REQUIRE(g && din && B > 0 && Hi > 0 && Wi > 0 && C > 0 && Ho > 0 && Wo > 0, MH_ERR_ARG, "mh_resize_image_bwd: bad argument");
    ResizeImgArgs a{nullptr, g, nullptr, din, B, Hi, Wi, C, Ho, Wo, (float)Hi / (float)Ho, (float)Wi / (float)Wo};
    hipLaunchKernelGGL(resize_image_bwd_kernel, dim3(grid_for((int64_t)B * Hi * Wi * C)), dim3(256), 0, (hipStream_t)stream, a);
    return mh_check_launch("resize_image_bwd");
}

extern "C" int mh_bilinear_sampler_fwd(const float* imgs, const float* coords, float* out, int32_t B, int32_t Hs, int32_t Ws, int32_t C,
                                       int32_t Ht, int32_t Wt, void* stream) {
    MH_REQUIRE(imgs && coords && out && B > 0 && Hs > 0 && Ws > 0 && C > 0 && Ht > 0 && Wt > 0, MH_ERR_ARG, "mh_bilinear_sampler_fwd: bad argument");
    MH_REQUIRE((int64_t)B * Hs * Ws < (1 << 24), MH_ERR_UNSUPPORTED,
               "mh_bilinear_sampler_fwd: B*Hs*Ws must stay below 2^24 (the reference computes the gather index in float32)");
    SamplerArgs a{imgs, coords, nullptr, out, nullptr, nullptr, B, Hs, Ws, C, Ht, Wt};
    hipLaunchKernelGGL(sampler_fwd_kernel, dim3(grid_for((int64_t)B * Ht * Wt)), dim3(256), 0, (hipStream_t)stream, a);
    return mh_check_launch("bilinear_sampler_fwd");
}

extern "C" int mh_bilinear_sampler_bwd(const float* g, const float* imgs, const float* coords, float* dcoords, float* dimgs, int32_t B,
                                       int32_t Hs, int32_t Ws, int32_t C, int32_t Ht, int32_t Wt, void* stream) {
    MH_REQUIRE(g && imgs && coords && (dcoords || dimgs) && B > 0 && Hs > 0 && Ws > 0 && C > 0 && Ht > 0 && Wt > 0, MH_ERR_ARG,
               "mh_bilinear_sampler_bwd: bad argument");
    MH_REQUIRE((int64_t)B * Hs * Ws < (1 << 24), MH_ERR_UNSUPPORTED, "mh_bilinear_sampler_bwd: B*Hs*Ws must stay below 2^24");
    SamplerArgs a{imgs, coords, g, nullptr, dcoords, dimgs, B, Hs, Ws, C, Ht, Wt};
    hipLaunchKernelGGL(sampler_bwd_kernel, dim3(grid_for((int64_t)B * Ht * Wt)), dim3(256), 0, (hipStream_t)stream, a);
    return mh_check_launch("bilinear_sampler_bwd");
}

extern "C" int mh_resize_bwd(const float* g, const float* in, float* din, int32_t accumulate, int32_t B, int32_t Hi, int32_t Wi,
                             int32_t Hr, int32_t Wr, int32_t cy, int32_t cx, int32_t Ho, int32_t Wo, float mul, int32_t mode,
                             void* stream) {
    MH_REQUIRE(g && in && din, MH_ERR_ARG, "mh_resize_bwd: null argument");
    ResizeArgs a{}; a.in = in; a.g = g; a.din = din; a.accumulate = accumulate;
    if (int e = resize_args(a, B, Hi, Wi, Hr, Wr, cy, cx, Ho, Wo, mul, mode)) return e;
    // rows of candidates per input pixel ~ 2 / sy (8 at an up-scaling of 4, 130 at 64: the full-resolution loss heads of the coarse
    // MAD blocks): split them over 4 / 16 / 64 lanes
    const int64_t npx = (int64_t)B * Hi * Wi;
    hipStream_t s = (hipStream_t)stream;
    if (a.sy <= 1.0f / 12.0f) hipLaunchKernelGGL((resize_bwd_kernel<64>), dim3(grid_for(npx * 64)), dim3(256), 0, s, a);
    else if (a.sy <= 1.0f / 6.0f) hipLaunchKernelGGL((resize_bwd_kernel<16>), dim3(grid_for(npx * 16)), dim3(256), 0, s, a);
    else if (a.sy <= 1.0f / 3.0f) hipLaunchKernelGGL((resize_bwd_kernel<4>), dim3(grid_for(npx * 4)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((resize_bwd_kernel<1>), dim3(grid_for(npx)), dim3(256), 0, s, a);
    return mh_check_launch("resize_bwd");
}

extern "C" int mh_head_bwd(const mh_head_bwd_desc* d, const float* src0, const float* src1, float* dV, void* dV_shadow, const float* w,
                           float* dx, const float* mask_ref, void* dx_shadow, void* stream) {
    MH_REQUIRE(d && dV && w && dx && (src0 || src1), MH_ERR_ARG, "mh_head_bwd: null argument");
    MH_REQUIRE(d->kind == 0 || d->kind == 1, MH_ERR_ARG, "mh_head_bwd: kind must be 0 (resize gradient of a finer level's du) or 1 (addends)");
    MH_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->N >= 4 && d->N <= 64 && d->N % 4 == 0, MH_ERR_ARG, "mh_head_bwd: bad dimension (N = 4 .. 64, multiple of 4)");
    MH_REQUIRE(d->dx_ld % 4 == 0 && d->dx_ld >= d->N && mh_aligned16(dx) && mh_aligned16(w) && (!mask_ref || (d->mask_ld % 4 == 0 && mh_aligned16(mask_ref))) &&
               (!dx_shadow || mh_aligned16(dx_shadow)), MH_ERR_ALIGN, "mh_head_bwd: dx / mask / w must be 16-byte aligned with lds that are multiples of 4");
    HeadBwdArgs a{};
    if (d->kind == 0) {
        MH_REQUIRE(src0, MH_ERR_ARG, "mh_head_bwd: kind 0 needs the finer level's coordinate gradient");
        a.rz.g = src0;
        if (int e = resize_args(a.rz, d->B, d->H, d->W, d->Hr, d->Wr, d->cy, d->cx, d->Ho, d->Wo, d->mul, 0)) return e;
    } else {
        a.rz.B = d->B; a.rz.Hi = d->H; a.rz.Wi = d->W;
        a.a1 = src0; a.a2 = src1; a.a1_ld = d->src0_ld; a.a2_ld = d->src1_ld;
        MH_REQUIRE((!src0 || d->src0_ld >= 1) && (!src1 || d->src1_ld >= 1), MH_ERR_ARG, "mh_head_bwd: addend pixel strides must be >= 1");
    }
    a.dV = dV; a.dV_sh = (unsigned short*)dV_shadow; a.dV_sh_ld = 32;
    a.w = w; a.dx = dx; a.mask_ref = mask_ref; a.dx_sh = (unsigned short*)dx_shadow;
    a.N = d->N; a.dx_ld = d->dx_ld; a.mask_ld = d->mask_ld; a.dx_sh_ld = (d->N + 31) / 32 * 32; a.acc_dx = d->accumulate_dx; a.kind = d->kind;
    a.mask_alpha = d->mask_alpha;
    a.tiles_x = (d->W + HB_TW - 1) / HB_TW; a.tiles_y = (d->H + HB_TH - 1) / HB_TH;
    const int64_t grid = (int64_t)d->B * a.tiles_x * a.tiles_y;
    MH_REQUIRE(grid < (1ll << 31), MH_ERR_ARG, "mh_head_bwd: too many tiles");
    hipLaunchKernelGGL(head_bwd_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    mh_note_kernel("head_bwd_kernel kind %d N=%d", d->kind, d->N);
    return mh_check_launch("head_bwd");
}

extern "C" int mh_pad_reflect(const float* in, float* out, int32_t B, int32_t H, int32_t W, int32_t C,
                              int32_t Hp, int32_t Wp, int32_t pad_t, int32_t pad_l, int32_t out_ld,
                              float div, float sub, void* stream) {
    MH_REQUIRE(in && out, MH_ERR_ARG, "mh_pad_reflect: null argument");
    MH_REQUIRE(B > 0 && H > 1 && W > 1 && C > 0 && Hp >= H && Wp >= W && out_ld >= C, MH_ERR_ARG, "mh_pad_reflect: bad dimension");
    MH_REQUIRE(pad_t >= 0 && pad_l >= 0 && pad_t < H && pad_l < W && Hp - H - pad_t < H && Wp - W - pad_l < W &&
               Hp - H - pad_t >= 0 && Wp - W - pad_l >= 0, MH_ERR_ARG, "mh_pad_reflect: REFLECT pad must be smaller than the image");
    MH_REQUIRE(div != 0.f, MH_ERR_ARG, "mh_pad_reflect: div must be non-zero");
    PadArgs a{in, out, B, H, W, C, Hp, Wp, pad_t, pad_l, out_ld, div, sub};
    hipLaunchKernelGGL(pad_reflect_kernel, dim3(grid_for((int64_t)B * Hp * Wp)), dim3(256), 0, (hipStream_t)stream, a);
    return mh_check_launch("pad_reflect");
}

static inline int64_t nblk(int64_t n) { return (n + 255) / 256; }

extern "C" int64_t mh_loss_ws_floats(int32_t B, int32_t H, int32_t W) {
    const int64_t n = (int64_t)B * H * W, nw = (int64_t)B * (H - 2) * (W - 2);
    return 8 * n + 12 * nw + nblk(n) + nblk(nw) + 64;
}

extern "C" int mh_reprojection_loss(const float* left, const float* right, const float* disp, float* ws, float* result,
                                    float* ddisp, float grad_scale, int32_t B, int32_t H, int32_t W, void* stream) {
    return mh_reprojection_loss_phase(left, right, disp, ws, result, ddisp, grad_scale, B, H, W, 0, stream);
}

extern "C" int mh_reprojection_loss_phase(const float* left, const float* right, const float* disp, float* ws, float* result,
                                          float* ddisp, float grad_scale, int32_t B, int32_t H, int32_t W, int32_t phase, void* stream) {
    MH_REQUIRE(phase >= 0 && phase <= 2, MH_ERR_ARG, "mh_reprojection_loss_phase: phase must be 0 (all), 1 (maps + gradient) or 2 (final reduction)");
    MH_REQUIRE(left && right && disp && ws && result, MH_ERR_ARG, "mh_reprojection_loss: null argument");
    MH_REQUIRE(B > 0 && H >= 3 && W >= 3, MH_ERR_ARG, "mh_reprojection_loss: image must be at least 3x3");
    MH_REQUIRE(mh_aligned16(ws), MH_ERR_ALIGN, "mh_reprojection_loss: workspace must be 16-byte aligned");
    const int64_t n = (int64_t)B * H * W, nw = (int64_t)B * (H - 2) * (W - 2);
    MH_REQUIRE(n * 12 < (1ll << 31) - 64, MH_ERR_ARG, "mh_reprojection_loss: too many pixels (32-bit byte offsets into the frames)");
    LossArgs a{};
    a.left = left; a.right = right; a.disp = disp; a.result = result; a.ddisp = ddisp; a.grad_scale = grad_scale;
    a.B = B; a.H = H; a.W = W;
    const int tiles_x = (W + LT_W - 1) / LT_W, tiles_y = (H + LT_H - 1) / LT_H;
    const int64_t ntiles = (int64_t)B * tiles_x * tiles_y;               // <= nblk(n): the workspace keeps its layout
    MH_REQUIRE(ntiles < (1ll << 31), MH_ERR_ARG, "mh_reprojection_loss: too many tiles");
    a.rep = ws; a.drep = ws + 4 * n; a.coef = ws + 8 * n;                  // (maps: LDS only since the tile kernel; the offsets keep the layout)
    a.part1 = ws + 8 * n + 12 * nw; a.nblk1 = (int)ntiles;
    a.part2 = a.part1 + nblk(n); a.nblk2 = (int)ntiles;
    hipStream_t s = (hipStream_t)stream;
    if (phase != 2) hipLaunchKernelGGL(loss_tile_kernel, dim3((unsigned)ntiles), dim3(256), 0, s, a, tiles_x, tiles_y, ddisp ? 1 : 0);
    if (phase != 1) hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, s, a);
    return mh_check_launch("reprojection_loss");
}

extern "C" int64_t mh_metrics_ws_floats(int32_t B, int32_t H, int32_t W) { return 3 * nblk((int64_t)B * H * W) + 16; }

extern "C" int mh_metrics(const float* disp, const float* gt, float* ws, float* result, float pixel_th,
                          int32_t B, int32_t H, int32_t W, void* stream) {
    MH_REQUIRE(disp && gt && ws && result, MH_ERR_ARG, "mh_metrics: null argument");
    MH_REQUIRE(B > 0 && H > 0 && W > 0, MH_ERR_ARG, "mh_metrics: bad dimension");
    MetArgs a{disp, gt, ws, result, (int64_t)B * H * W, (int)nblk((int64_t)B * H * W), pixel_th};
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(metrics_kernel, dim3(a.nblk), dim3(256), 0, s, a);
    hipLaunchKernelGGL(metrics_final_kernel, dim3(1), dim3(256), 0, s, a);
    return mh_check_launch("metrics");
}

extern "C" int64_t mh_proxy_ws_floats(int32_t B, int32_t H, int32_t W) { return 2 * nblk((int64_t)B * H * W); }

static int masked_l1(const char* what, const float* pred, const float* target, float* ws, float* result, float* dpred, float weight,
                     float grad_scale, float hi, int zero_only, int32_t B, int32_t H, int32_t W, void* stream) {
    MH_REQUIRE(pred && target && ws && result, MH_ERR_ARG, "%s: null argument", what);
    MH_REQUIRE(B > 0 && H > 0 && W > 0, MH_ERR_ARG, "%s: bad dimension", what);
    ProxyArgs a{pred, target, ws, result, dpred, (int64_t)B * H * W, (int)nblk((int64_t)B * H * W), weight, grad_scale, hi, zero_only};
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(proxy_partial_kernel, dim3(a.nblk), dim3(256), 0, s, a);
    hipLaunchKernelGGL(proxy_final_kernel, dim3(1), dim3(256), 0, s, a);
    if (dpred) hipLaunchKernelGGL(proxy_grad_kernel, dim3(a.nblk), dim3(256), 0, s, a);
    return mh_check_launch(what);
}

extern "C" int mh_proxy_loss(const float* pred, const float* proxy, float* ws, float* result, float* dpred, float weight,
                             float grad_scale, int32_t B, int32_t H, int32_t W, void* stream) {
    return masked_l1("mh_proxy_loss", pred, proxy, ws, result, dpred, weight, grad_scale, 192.0f, 0, B, H, W, stream);
}

extern "C" int mh_supervised_loss(const float* pred, const float* target, float* ws, float* result, float* dpred, float weight,
                                  float grad_scale, float max_disp, int32_t B, int32_t H, int32_t W, void* stream) {
    MH_REQUIRE(max_disp > 0.f, MH_ERR_ARG, "mh_supervised_loss: max_disp must be positive");
    return masked_l1("mh_supervised_loss", pred, target, ws, result, dpred, weight, grad_scale, max_disp, 1, B, H, W, stream);
}

extern "C" int mh_adam(float* var, float* m, float* v, const float* grad, int64_t n, const float* state, float lr, float beta1,
                       float beta2, float eps, float grad_scale, void* stream) {
    MH_REQUIRE(var && m && v && grad && state && n > 0, MH_ERR_ARG, "mh_adam: bad argument");
    MH_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f, MH_ERR_ARG, "mh_adam: beta out of [0, 1)");
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, var, m, v, grad, n, state, lr, beta1, beta2, eps, grad_scale);
    return mh_check_launch("adam");
}

extern "C" int mh_adam_advance(float* state, float beta1, float beta2, void* stream) {
    MH_REQUIRE(state, MH_ERR_ARG, "mh_adam_advance: null state");
    hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, beta1, beta2);
    return mh_check_launch("adam_advance");
}

extern "C" int mh_momentum(float* var, float* accum, const float* grad, int64_t n, float lr, float momentum,
                           float grad_scale, void* stream) {
    MH_REQUIRE(var && accum && grad && n > 0, MH_ERR_ARG, "mh_momentum: bad argument");
    const int64_t n4 = n >> 2;
    if (n4 > 0 && (((uintptr_t)var | (uintptr_t)accum | (uintptr_t)grad) & 15u) == 0)
        hipLaunchKernelGGL(momentum4_kernel, dim3(grid_for(n4)), dim3(256), 0, (hipStream_t)stream, (float4*)var, (float4*)accum, (const float4*)grad, n4, lr, momentum,
                           grad_scale);
    else if (n4 > 0) { hipLaunchKernelGGL(momentum_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, var, accum, grad, n, lr, momentum, grad_scale); return mh_check_launch("momentum"); }
    if (n & 3)          // (the tail of a range that is not a multiple of 4)
        hipLaunchKernelGGL(momentum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, var + 4 * n4, accum + 4 * n4, grad + 4 * n4, n & 3, lr, momentum, grad_scale);
    return mh_check_launch("momentum");
}

extern "C" int mh_copy_channels(const float* src, int32_t src_ld, float* dst, int32_t dst_ld, int64_t npix,
                                int32_t nch, float scale, int32_t accumulate, void* stream) {
    MH_REQUIRE(src && dst && npix > 0 && nch > 0 && src_ld >= nch && dst_ld >= nch, MH_ERR_ARG, "mh_copy_channels: bad argument");
    hipLaunchKernelGGL(copy_channels_kernel, dim3(grid_for(npix * nch)), dim3(256), 0, (hipStream_t)stream,
                       src, src_ld, dst, dst_ld, npix, nch, scale, accumulate);
    return mh_check_launch("copy_channels");
}

extern "C" int mh_leaky_bwd(float* dy, int32_t dy_ld, const float* y, int32_t y_ld, int64_t npix, int32_t nch,
                            float alpha, void* stream) {
    MH_REQUIRE(dy && y && npix > 0 && nch > 0, MH_ERR_ARG, "mh_leaky_bwd: bad argument");
    hipLaunchKernelGGL(leaky_bwd_kernel, dim3(grid_for(npix * nch)), dim3(256), 0, (hipStream_t)stream, dy, dy_ld, y, y_ld, npix, nch, alpha);
    return mh_check_launch("leaky_bwd");
}

extern "C" int mh_bias_grad(const float* dz, int32_t dz_ld, int64_t npix, int32_t nch, float* db, void* stream) {
    MH_REQUIRE(dz && db && npix > 0 && nch > 0 && dz_ld >= nch, MH_ERR_ARG, "mh_bias_grad: bad argument");
    int nchp = 1;
    while (nchp < nch && nchp < 64) nchp <<= 1;
    const int64_t rows = (npix + (64 / nchp) - 1) / (64 / nchp);          // wave rows of work
    int gx = (int)((rows + 4 * 8 - 1) / (4 * 8));                          // ~8 rows per wave
    if (gx > 1024) gx = 1024;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(bias_grad_kernel, dim3(gx, (nch + 63) / 64), dim3(256), 0, (hipStream_t)stream, dz, dz_ld, npix, nch, db, nchp, (float*)nullptr);
    return mh_check_launch("bias_grad");
}

extern "C" int mh_bias_grad_blocks(int64_t npix, int32_t nch) {
    if (npix <= 0 || nch <= 0) return 0;
    int nchp = 1;
    while (nchp < nch && nchp < 64) nchp <<= 1;
    const int64_t rows = (npix + (64 / nchp) - 1) / (64 / nchp);
    int64_t gx = (rows + 4 * 8 - 1) / (4 * 8);
    return (int)(gx > 1024 ? 1024 : (gx < 1 ? 1 : gx));
}

extern "C" int mh_bias_grad_partial(const float* dz, int32_t dz_ld, int64_t npix, int32_t nch, float* ws, int32_t nblocks, void* stream) {
    MH_REQUIRE(dz && ws && npix > 0 && nch > 0 && dz_ld >= nch, MH_ERR_ARG, "mh_bias_grad_partial: bad argument");
    MH_REQUIRE(nblocks == mh_bias_grad_blocks(npix, nch), MH_ERR_ARG, "mh_bias_grad_partial: nblocks %d must be mh_bias_grad_blocks(npix, nch) = %d (the workspace was sized for it)",
               nblocks, mh_bias_grad_blocks(npix, nch));
    int nchp = 1;
    while (nchp < nch && nchp < 64) nchp <<= 1;
    hipLaunchKernelGGL(bias_grad_kernel, dim3(nblocks, (nch + 63) / 64), dim3(256), 0, (hipStream_t)stream, dz, dz_ld, npix, nch, (float*)nullptr, nchp, ws);
    return mh_check_launch("bias_grad_partial");
}

// device time stamp (diagnostics of the replayed step: where the side lane starts, how long the tail behind the last input gradient is):
// one lane stores the constant-rate wall clock (s_memrealtime, 100 MHz on gfx950) into a slot.  A plan op like any other, so the stamp sits at
// its place in the captured graph and needs no tracer (rocprofv3 shifts the side queue by ~0.3 ms: profiles/r03_experiments.txt #3).
// deterministic mode: float buffer += its fixed-point twin (value * 2^48), twin cleared (mh_common.h: mh_atomic_add)
__global__ __launch_bounds__(256) void det_flush_kernel(float* dst, long long* acc, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const long long a = acc[i];
        if (a) { dst[i] += (float)((double)a * (1.0 / 281474976710656.0)); acc[i] = 0; }
    }
}
extern "C" int mh_det_flush(float* dst, void* acc, int64_t n, void* stream) {
    MH_REQUIRE(dst && acc && n >= 0, MH_ERR_ARG, "mh_det_flush: null buffer");
    if (n == 0) return 0;
    const int64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(det_flush_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream, dst, (long long*)acc, n);
    return mh_check_launch("det_flush");
}

__global__ void stamp_kernel(long long* slot) { *slot = wall_clock64(); }
extern "C" int mh_stamp(void* slot, void* stream) {
    MH_REQUIRE(slot && (((uintptr_t)slot) & 7u) == 0, MH_ERR_ARG, "mh_stamp: 8-byte aligned slot");
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (long long*)slot);
    return mh_check_launch("stamp");
}
extern "C" int64_t mh_stamp_rate_khz(void) {
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) return 0;
    return khz;
}

extern "C" int mh_fill(float* p, int64_t n, float v, void* stream) {
    MH_REQUIRE(p && n > 0, MH_ERR_ARG, "mh_fill: bad argument");
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, n, v);
    return mh_check_launch("fill");
}

// this translation unit's copy of the deterministic-accumulation table (mh_common.h)
extern "C" int mh_det_sync_ops(const void* t) { return mh_det_upload(*reinterpret_cast<const mh_det_table*>(t)); }
extern "C" int mh_det_ovf_ops(void) { return mh_det_overflow_take(); }      // this translation unit's saturation flag of the deterministic twin (mh_common.h)
