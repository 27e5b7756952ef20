// corr.hip -- 1-D correlation / cost volume for gfx950, forward and gradient.
//
// Replaces sharedLayers.correlation (Nets/sharedLayers.py:23-51; the 'TF' formulation is the
// oracle) and the CUDA op behind it: CorrelateData / CorrelateDataBackward0/1 and their
// launchers (Nets/Native/shift_corr.cu.cc:17-289).  Differences by design: inputs are
// UN-padded NHWC (zero padding handled in-kernel), the output is NHWC written straight into
// the estimator's concatenated input [reference | corr | upsampled_disp] (MadNet.py:77-80,
// 370-375) so tf.concat costs nothing, and the backward follows the TF gradient (the CUDA
// backward is defective).
//
// HBM-bound op: algorithmic bytes = B*H*W*(2C + D)*4.  Forward kernels, by shift count D:
//   corr_fwd_direct (D <= 9, the MADNet radius-2 volume; default): a group of LPP lanes owns a pixel and splits the
//     channels, 1 + D independent 16-byte buffer loads per lane (the right-feature window is served from L1/L2), the D
//     partial dot products meet in __shfl_xor butterflies -- 67-75 % of the HBM peak at the SURVEY 8(d) protocol shape;
//   corr_fwd_small (D <= 9, LDS-staged right window, plain pointers): the path of feature maps >= 2 GiB (no buffer descriptor) and the A/B partner of
//     corr_fwd_direct behind mh_tune_corr(0) -- measured slower than direct;
//   corr_fwd_mfma (D > 9, DispNet's 81-shift volume): the band of the row-wise product L * R^T on the fp32 MFMA;
//   corr_fwd_large (generic fallback for the fused-concat forms of large D).
#include <algorithm>
#include <atomic>
#include <type_traits>
#include "mh_common.h"

namespace {

constexpr int MAXD_SMALL = 9;   // register-resident shift count of the small-D kernel

struct CorrArgs {
    const float* L; const float* R; const float* u; float* out;
    int l_ld, r_ld, out_ld, coff;
    int B, H, W, C, md, stride, D;
    int copy_left, zero_tail, segs;
    unsigned l_bytes, r_bytes;
    int remap;       // large-D bf16 kernel: XCD-aware workgroup order (the segments of an image row share one L2)
};

// LPP lanes per pixel (power of two <= 64), TW pixels per workgroup segment.
template <int LPP, int TW>
__global__ __launch_bounds__(256) void corr_fwd_small(CorrArgs p) {
    HIP_DYNAMIC_SHARED(float, smem)   // [(TW + 2md)][C]; no static LDS in this kernel -> base is 16-byte aligned
    const int tid = threadIdx.x;
    const int seg = blockIdx.x % p.segs;
    const int row = blockIdx.x / p.segs;          // b*H + y
    const int x0 = seg * TW;
    const int C4 = p.C >> 2;
    const int win = TW + 2 * p.md;
    const float* Rrow = p.R + (int64_t)row * p.W * p.r_ld;
    // ---- stage the right window (zero outside the row: tf.pad in correlation_tf) ----------
    for (int q = tid; q < win * C4; q += 256) {
        const int px = q / C4, c4 = q - px * C4;
        const int xs = x0 - p.md + px;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (xs >= 0 && xs < p.W) v = *reinterpret_cast<const float4*>(Rrow + (int64_t)xs * p.r_ld + c4 * 4);
        *reinterpret_cast<float4*>(&smem[(px * C4 + c4) * 4]) = v;
    }
    __syncthreads();

    constexpr int PPB = 256 / LPP;               // pixels per pass
    const int sub = tid % LPP;
    const float inv_c = 1.0f / (float)p.C;
    for (int pp = tid / LPP; pp < TW; pp += PPB) {
        const int x = x0 + pp;
        const bool live = x < p.W;
        float accd[MAXD_SMALL];
#pragma unroll
        for (int j = 0; j < MAXD_SMALL; ++j) accd[j] = 0.f;
        const int64_t pix = (int64_t)row * p.W + (live ? x : 0);
        const float* Lp = p.L + pix * p.l_ld;
        float* Op = p.out + pix * p.out_ld;
        for (int c4 = sub; c4 < C4; c4 += LPP) {
            float4 l = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live) {
                l = *reinterpret_cast<const float4*>(Lp + c4 * 4);
                if (p.copy_left) *reinterpret_cast<float4*>(Op + c4 * 4) = l;
            }
#pragma unroll
            for (int j = 0; j < MAXD_SMALL; ++j) {
                if (j < p.D) {
                    const float4 r = *reinterpret_cast<const float4*>(&smem[((pp + j * p.stride) * C4 + c4) * 4]);
                    accd[j] += l.x * r.x + l.y * r.y + l.z * r.z + l.w * r.w;
                }
            }
        }
        // butterfly over the LPP lanes that share this pixel
#pragma unroll
        for (int j = 0; j < MAXD_SMALL; ++j) {
            if (j < p.D) {
#pragma unroll
                for (int o = LPP >> 1; o > 0; o >>= 1) accd[j] += __shfl_xor(accd[j], o);
            }
        }
        if (live && sub == 0) {
            float* dst = Op + p.coff;
#pragma unroll
            for (int j = 0; j < MAXD_SMALL; ++j)
                if (j < p.D) dst[j] = accd[j] * inv_c;
            int tail = p.coff + p.D;
            if (p.u) { Op[tail] = p.u[pix]; ++tail; }
            if (p.zero_tail)
                for (; tail < p.out_ld; ++tail) Op[tail] = 0.f;
        }
    }
}

// Direct variant for small D: no LDS, no barrier.  Every lane issues its left float4 and the D shifted
// right float4 as independent bounds-checked buffer loads (a shift that leaves the row gets an
// out-of-range offset => the hardware returns the zero padding of correlation_tf), so D+1 16-byte loads
// per lane are in flight at once; neighbouring pixels re-read the same right rows from L1 (5x request
// amplification at the TA, still far below its bandwidth), HBM sees each byte once.
template <int LPP, int DT>
__global__ __launch_bounds__(256) void corr_fwd_direct(CorrArgs p) {
    // One group of 256/LPP pixels per workgroup, no persistent loop (the launch covers every pixel: a
    // grid of tens of thousands of small workgroups keeps more bytes in flight than a capped grid
    // looping over 64-bit indices -- measured 74 % vs 45 % of the HBM peak, profiles/r01_corr_experiment.txt),
    // 32-bit index math, DT = compile-time shift count (5 for radius_d 2) so that no dead load is issued.
    constexpr int PPB = 256 / LPP;
    const int tid = threadIdx.x;
    const int sub = tid % LPP;
    const int C4 = p.C >> 2;
    const float inv_c = 1.0f / (float)p.C;
    const int npix = p.B * p.H * p.W;
    const __amdgpu_buffer_rsrc_t rsL = mh_make_rsrc(p.L, p.l_bytes);
    const __amdgpu_buffer_rsrc_t rsR = mh_make_rsrc(p.R, p.r_bytes);
    const int pix = blockIdx.x * PPB + tid / LPP;
    const bool live = pix < npix;
    const int pp = live ? pix : 0;
    const int x = pp % p.W;
    float accd[DT];
#pragma unroll
    for (int j = 0; j < DT; ++j) accd[j] = 0.f;
    float* Op = p.out + (int64_t)pp * p.out_ld;
    for (int c4 = sub; c4 < C4; c4 += LPP) {
        const float4 l = mh_buf_load4(rsL, live ? (pp * p.l_ld + c4 * 4) * 4 : MH_OOB);
        float4 r[DT];
#pragma unroll
        for (int j = 0; j < DT; ++j) {
            const int dx = j * p.stride - p.md;
            const bool ok = live && (j < p.D) && (unsigned)(x + dx) < (unsigned)p.W;
            r[j] = mh_buf_load4(rsR, ok ? ((pp + dx) * p.r_ld + c4 * 4) * 4 : MH_OOB);
        }
        if (live && p.copy_left) *reinterpret_cast<float4*>(Op + c4 * 4) = l;
#pragma unroll
        for (int j = 0; j < DT; ++j) accd[j] += l.x * r[j].x + l.y * r[j].y + l.z * r[j].z + l.w * r[j].w;
    }
#pragma unroll
    for (int j = 0; j < DT; ++j) {
#pragma unroll
        for (int o = LPP >> 1; o > 0; o >>= 1) accd[j] += __shfl_xor(accd[j], o);
    }
    // outputs [coff, coff+D) = costs, then u, then the zero tail: entry e is written by lane e % LPP, so the
    // D costs of a pixel leave as one store instruction of adjacent dwords instead of D one-lane stores
    if (live) {
        const int nout = p.zero_tail ? p.out_ld - p.coff : p.D + (p.u ? 1 : 0);
        for (int e = sub; e < nout; e += LPP) {
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < DT; ++j) v = (e == j) ? accd[j] * inv_c : v;
            if (p.u && e == p.D) v = p.u[pp];
            Op[p.coff + e] = v;
        }
    }
}

// ---- fused front end of one pyramid level (MadNet.py:274-295 per level): the inter-level upsample u = mul * resize(V_coarse)
// (tf.image.resize_images, TF1 legacy bilinear), the horizontal linear warp of the right features by u (_build_indeces +
// _linear_warping, MadNet.py:378-436: taps outside the row get weight ZERO) and the cost volume + concat, in ONE launch instead
// of three (resize_fwd, warp_fwd, corr_fwd): at batch 1 these ops are launch-latency bound (5 us each for < 1 us of traffic).
// The warped right pixel at every shift is rebuilt from two raw right pixels on the fly (1 + 2*D loads per lane instead of 1 + D,
// all L1/L2 hits); the group of a pixel also stores its own warped feature row and u, which the backward pass reads.
struct FrontArgs {
    const float* Vc; const float* L; const float* R; float* out; float* Rw; float* u;
    int Hc, Wc; float mul, sy, sx;
    int l_ld, r_ld, out_ld, rw_ld, coff;
    int B, H, W, C, md, D, zero_tail;
    unsigned l_bytes, r_bytes;
    unsigned short* out_hi; unsigned short* out_lo; int out_pld;      // != null: the estimator input also leaves as bf16 planes (hi [+ lo])
    // HEAD instances (mh_level_front_head_fwd): Vc is an OUTPUT -- the coarser level's disparity head (3x3, K -> 1, linear) runs in this launch
    const float* X; const float* hw; const float* hb; float* Vw; int x_ld, K, segs, cwcap, rows2, rowpairs; unsigned x_bytes;
};

// HEAD (round 5, mh_level_front_head_fwd): the disparity head of the COARSER level -- Vc = conv3x3(X, hw) + hb, 32 -> 1 channels, linear (MadNet.py:118) -- is
// computed here instead of by a launch of its own in front of this one (a 4.5 - 5 us node of the forward chain per level for 0.1 - 2 MFLOP).  A workgroup
// owns PPB pixels of ONE row: they read two coarse rows x <= cwcap columns of Vc; the workgroup stages the 4 x (cw + 2) x K patch of X those need, computes
// the <= 2 * cwcap head values with conv_n1_fwd_kernel's arithmetic (8 lanes per value, same tap order, same butterfly), keeps them in LDS for its own
// interpolation and stores them to Vc: every coarse pixel is read by some fine pixel, so the union of the stores is the whole map (neighbouring workgroups
// write identical values to the columns they share).
template <int LPP, int DT, bool HEAD>
__global__ __launch_bounds__(256) void level_front_kernel(FrontArgs p) {
    constexpr int PPB = 256 / LPP;
    HIP_DYNAMIC_SHARED(float, fsm)                // HEAD: Xs [4][cwcap + 2][K] | Ws [9][K] | Vs [2][cwcap]
    const int tid = threadIdx.x;
    const int sub = tid % LPP;
    const int C4 = p.C >> 2;
    const float inv_c = 1.0f / (float)p.C;
    const int npix = p.B * p.H * p.W;
    const __amdgpu_buffer_rsrc_t rsL = mh_make_rsrc(p.L, p.l_bytes);
    const __amdgpu_buffer_rsrc_t rsR = mh_make_rsrc(p.R, p.r_bytes);
    int pix;
    bool live;
    const int wg = (int)blockIdx.x;              // (an XCD-aware order -- bands of rows per XCD -- measured no better: r05_experiments.txt #17)
    if constexpr (HEAD) {
        const int seg = wg % p.segs, rowi = wg / p.segs;
        if (p.rows2) {
            // H = 2 Hc: the fine rows 2r and 2r + 1 interpolate from the SAME two coarse rows -- a workgroup takes PPB / 2 columns of both, and the head values it
            // has to compute (2 x its coarse columns) drop by a third (r05_experiments.txt #17)
            constexpr int HALF = PPB / 2;
            const int rp = rowi % p.rowpairs, bb = rowi / p.rowpairs;
            const int j = tid / LPP, r = j >= HALF ? 1 : 0;
            const int xx = seg * HALF + (j - r * HALF);
            live = xx < p.W;
            pix = (bb * p.H + 2 * rp + r) * p.W + (live ? xx : 0);
        } else {
            const int xx = seg * PPB + tid / LPP;
            live = xx < p.W;
            pix = rowi * p.W + (live ? xx : 0);
        }
    } else {
        pix = blockIdx.x * PPB + tid / LPP;
        live = pix < npix;
    }
    const int pp = (HEAD || live) ? pix : 0;      // (HEAD: a lane without a pixel still belongs to its workgroup's row -- it stages that row's patch)
    const int x = pp % p.W;
    const int t2 = pp / p.W;
    const int y = t2 % p.H, b = t2 / p.H;
    const int row = pp - x;
    // vertical taps of the legacy resize are shared by the D shifts (same row)
    const float srcy = (float)y * p.sy;
    const int y0 = (int)srcy, y1 = min(y0 + 1, p.Hc - 1);
    const float ty = srcy - (float)y0;
    const float* V0 = p.Vc + ((int64_t)b * p.Hc + y0) * p.Wc;
    const float* V1 = p.Vc + ((int64_t)b * p.Hc + y1) * p.Wc;
    int c_lo = 0;
    if constexpr (HEAD) {
        // (y, b and hence y0 / y1 are workgroup-uniform: one row per workgroup)
        const int G4 = p.K >> 2;
        const int wpx = p.rows2 ? PPB / 2 : PPB;             // columns of this workgroup
        const int xs0 = (wg % p.segs) * wpx;
        const int flo = max(xs0 - p.md, 0), fhi = min(xs0 + wpx - 1 + p.md, p.W - 1);
        c_lo = (int)((float)flo * p.sx);
        const int c_hi = min((int)((float)fhi * p.sx) + 1, p.Wc - 1);
        const int cw = c_hi - c_lo + 1;                     // <= cwcap (the launcher's bound)
        const int pw = cw + 2;
        float* const Xs = fsm;
        float* const Ws = fsm + 4 * (p.cwcap + 2) * p.K;
        float* const Vs = Ws + 9 * p.K;
        const __amdgpu_buffer_rsrc_t rsX = mh_make_rsrc(p.X, p.x_bytes);
        // patch rows y0 - 1 .. y0 + 2, columns c_lo - 1 .. c_hi + 1; all loads in flight before the first LDS store (the launcher guarantees items <= 6 * 256)
        constexpr int U = 6;
        const int items = 4 * pw * G4;
        float4 xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = tid + u * 256;
            const int g = q % G4, t3 = q / G4;
            const int pc = t3 % pw, pr = t3 / pw;
            const int yc = y0 - 1 + pr, xc = c_lo - 1 + pc;
            const bool ok = q < items && (unsigned)yc < (unsigned)p.Hc && (unsigned)xc < (unsigned)p.Wc;
            xv[u] = mh_buf_load4(rsX, ok ? (((b * p.Hc + yc) * p.Wc + xc) * p.x_ld + g * 4) * 4 : MH_OOB);
        }
        float4 wv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tid < 9 * G4) wv = reinterpret_cast<const float4*>(p.hw)[tid];
        const float hbias = p.hb ? p.hb[0] : 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = tid + u * 256;
            if (q < items) *reinterpret_cast<float4*>(Xs + q * 4) = xv[u];          // [pr][pc][g] with pw columns: q = (pr * pw + pc) * G4 + g
        }
        if (tid < 9 * G4) *reinterpret_cast<float4*>(Ws + tid * 4) = wv;
        __syncthreads();
        const int dy1 = y1 - y0;                            // 0 on the last coarse row
        const int nv = 2 * cw;
        const int hs = tid & 7;                              // 8 lanes per head value (conv_n1_fwd_kernel<8>: one channel group per lane)
        for (int v0 = 0; v0 < nv; v0 += 32) {                // (uniform trip count: the butterfly below needs every lane)
            const int v = v0 + (tid >> 3);
            const bool vl = v < nv;
            const int r = vl ? v / cw : 0, c = vl ? v - r * cw : 0;
            const int dy = r ? dy1 : 0;
            float acc = 0.f;
            if (hs < G4) {
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int ky = t / 3, kx = t - ky * 3;
                    const float4 xq = *reinterpret_cast<const float4*>(Xs + (((dy + ky) * pw + c + kx) * G4 + hs) * 4);
                    const float4 w = *reinterpret_cast<const float4*>(Ws + t * p.K + hs * 4);      // (the bank in registers -- 9 loads per lane -- measured SLOWER: #11)
                    acc += (xq.x * w.x + xq.y * w.y) + (xq.z * w.z + xq.w * w.w);
                }
            }
#pragma unroll
            for (int o = 4; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
            if (vl && hs == 0) {
                const float val = acc + hbias;
                Vs[r * p.cwcap + c] = val;
                p.Vw[((int64_t)b * p.Hc + y0 + dy) * p.Wc + c_lo + c] = val;
            }
        }
        __syncthreads();
    }
    float w0[DT], w1[DT], uc = 0.f;
    int o0[DT], o1[DT];
#pragma unroll
    for (int j = 0; j < DT; ++j) {
#if defined(__clang__)
#pragma clang fp contract(off)          // same un-contracted interpolation arithmetic as resize_fwd_kernel (ops.hip): bit-identical u
#endif
        const int xs = x + j - p.md;
        const bool in = live && j < p.D && (unsigned)xs < (unsigned)p.W;
        const int xq = in ? xs : 0;
        const float srcx = (float)xq * p.sx;
        const int x0 = (int)srcx, x1 = min(x0 + 1, p.Wc - 1);
        const float tx = srcx - (float)x0;
        float tl, tr, bl, br;
        if constexpr (HEAD) {
            const float* const Vs = fsm + 4 * (p.cwcap + 2) * p.K + 9 * p.K;      // this workgroup's two coarse rows, columns from c_lo
            const int i0 = in ? x0 - c_lo : 0, i1 = in ? x1 - c_lo : 0;          // (a lane without a pixel / a shift outside the row reads nothing of its own)
            tl = Vs[i0]; tr = Vs[i1]; bl = Vs[p.cwcap + i0]; br = Vs[p.cwcap + i1];
        } else { tl = V0[x0]; tr = V0[x1]; bl = V1[x0]; br = V1[x1]; }
        const float top = tl + (tr - tl) * tx, bot = bl + (br - bl) * tx;
        const float uj = (top + (bot - top) * ty) * p.mul;                 // resize_fwd mode 0
        if (j == p.md) uc = uj;
        const float cx = (float)xq + uj;
        const float f0 = floorf(cx), f1 = f0 + 1.0f;
        const float xmax = (float)(p.W - 1);
        const float f0s = fminf(fmaxf(f0, 0.f), xmax), f1s = fminf(fmaxf(f1, 0.f), xmax);
        w0[j] = in ? (f1 - cx) * (f0 == f0s ? 1.f : 0.f) : 0.f;
        w1[j] = in ? (cx - f0) * (f1 == f1s ? 1.f : 0.f) : 0.f;
        o0[j] = in ? (row + (int)f0s) * p.r_ld * 4 : MH_OOB;
        o1[j] = in ? (row + (int)f1s) * p.r_ld * 4 : MH_OOB;
    }
    float accd[DT];
#pragma unroll
    for (int j = 0; j < DT; ++j) accd[j] = 0.f;
    float* Op = p.out + (int64_t)pp * p.out_ld;
    for (int c4 = sub; c4 < C4; c4 += LPP) {
        const float4 l = mh_buf_load4(rsL, live ? (pp * p.l_ld + c4 * 4) * 4 : MH_OOB);
        float4 a[DT], bb[DT];
#pragma unroll
        for (int j = 0; j < DT; ++j) {
            a[j] = mh_buf_load4(rsR, o0[j] == MH_OOB ? MH_OOB : o0[j] + c4 * 16);
            bb[j] = mh_buf_load4(rsR, o1[j] == MH_OOB ? MH_OOB : o1[j] + c4 * 16);
        }
        if (live) *reinterpret_cast<float4*>(Op + c4 * 4) = l;
        if (live && p.out_hi) {
            uint2 hi, lo;
            mh_split_bf16x2(l.x, l.y, hi.x, lo.x);
            mh_split_bf16x2(l.z, l.w, hi.y, lo.y);
            *reinterpret_cast<uint2*>(p.out_hi + (int64_t)pp * p.out_pld + c4 * 4) = hi;
            if (p.out_lo) *reinterpret_cast<uint2*>(p.out_lo + (int64_t)pp * p.out_pld + c4 * 4) = lo;
        }
#pragma unroll
        for (int j = 0; j < DT; ++j) {
            float4 r;
            r.x = w0[j] * a[j].x + w1[j] * bb[j].x; r.y = w0[j] * a[j].y + w1[j] * bb[j].y;
            r.z = w0[j] * a[j].z + w1[j] * bb[j].z; r.w = w0[j] * a[j].w + w1[j] * bb[j].w;
            accd[j] += l.x * r.x + l.y * r.y + l.z * r.z + l.w * r.w;
            if (j == p.md && live) *reinterpret_cast<float4*>(p.Rw + (int64_t)pp * p.rw_ld + c4 * 4) = r;
        }
    }
#pragma unroll
    for (int j = 0; j < DT; ++j) {
#pragma unroll
        for (int o = LPP >> 1; o > 0; o >>= 1) accd[j] += __shfl_xor(accd[j], o);
    }
    if (live) {
        const int nout = p.zero_tail ? p.out_ld - p.coff : p.D + 1;
        for (int e = sub; e < nout; e += LPP) {
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < DT; ++j) v = (e == j) ? accd[j] * inv_c : v;
            if (e == p.D) v = uc;
            Op[p.coff + e] = v;
            if (p.out_hi && e <= p.D) {
                unsigned hi, lo;
                mh_split_bf16x2(v, 0.f, hi, lo);
                p.out_hi[(int64_t)pp * p.out_pld + p.coff + e] = (unsigned short)hi;
                if (p.out_lo) p.out_lo[(int64_t)pp * p.out_pld + p.coff + e] = (unsigned short)lo;
            }
        }
        if (sub == 0) p.u[pp] = uc;
    }
}

// Generic shift count (DispNet, D = 81): lane = one (pixel, shift) pair, channels looped from
// LDS (left tile + right window staged once, rows padded by 4 floats against bank conflicts).
template <int TW>
__global__ __launch_bounds__(256) void corr_fwd_large(CorrArgs p) {
    HIP_DYNAMIC_SHARED(float, smem)
    const int tid = threadIdx.x;
    const int seg = blockIdx.x % p.segs;
    const int row = blockIdx.x / p.segs;
    const int x0 = seg * TW;
    const int C4 = p.C >> 2;
    const int rs = p.C + 4;                       // padded row stride (floats)
    const int win = TW + 2 * p.md;
    float* Ls = smem;                             // [TW][rs]
    float* Rs = smem + TW * rs;                   // [win][rs]
    const float* Rrow = p.R + (int64_t)row * p.W * p.r_ld;
    const float* Lrow = p.L + (int64_t)row * p.W * p.l_ld;
    for (int q = tid; q < win * C4; q += 256) {
        const int px = q / C4, c4 = q - px * C4;
        const int xs = x0 - p.md + px;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (xs >= 0 && xs < p.W) v = *reinterpret_cast<const float4*>(Rrow + (int64_t)xs * p.r_ld + c4 * 4);
        *reinterpret_cast<float4*>(&Rs[px * rs + c4 * 4]) = v;
    }
    for (int q = tid; q < TW * C4; q += 256) {
        const int px = q / C4, c4 = q - px * C4;
        const int x = x0 + px;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (x < p.W) {
            v = *reinterpret_cast<const float4*>(Lrow + (int64_t)x * p.l_ld + c4 * 4);
            if (p.copy_left) *reinterpret_cast<float4*>(p.out + ((int64_t)row * p.W + x) * p.out_ld + c4 * 4) = v;
        }
        *reinterpret_cast<float4*>(&Ls[px * rs + c4 * 4]) = v;
    }
    __syncthreads();
    const float inv_c = 1.0f / (float)p.C;
    for (int q = tid; q < TW * p.D; q += 256) {
        const int pp = q / p.D, j = q - pp * p.D;
        const int x = x0 + pp;
        if (x >= p.W) continue;
        const float* l = &Ls[pp * rs];
        const float* r = &Rs[(pp + j * p.stride) * rs];
        float acc = 0.f;
        for (int c4 = 0; c4 < C4; ++c4) {
            const float4 a = *reinterpret_cast<const float4*>(l + c4 * 4);
            const float4 b = *reinterpret_cast<const float4*>(r + c4 * 4);
            acc += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
        }
        p.out[((int64_t)row * p.W + x) * p.out_ld + p.coff + j] = acc * inv_c;
    }
    // tail (u / zero padding), one lane per pixel
    for (int pp = tid; pp < TW; pp += 256) {
        const int x = x0 + pp;
        if (x >= p.W) continue;
        const int64_t pix = (int64_t)row * p.W + x;
        float* Op = p.out + pix * p.out_ld;
        int tail = p.coff + p.D;
        if (p.u) { Op[tail] = p.u[pix]; ++tail; }
        if (p.zero_tail)
            for (; tail < p.out_ld; ++tail) Op[tail] = 0.f;
    }
}

struct CorrBwdArgs {
    const float* g; const float* L; const float* R; float* dL; float* dR; float* du;
    int g_ld, coff, l_ld, r_ld, dl_ld, dr_ld;
    int acc_l, acc_r, acc_u;
    int B, H, W, C, md, stride, D, copy_left;
    int64_t total;   // B*H*W*C4
};

// gather form (no atomics, deterministic): one lane = one (pixel, 4-channel group).
__global__ __launch_bounds__(256) void corr_bwd_kernel(CorrBwdArgs p) {
    const int C4 = p.C >> 2;
    const float inv_c = 1.0f / (float)p.C;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < p.total; q += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(q % C4);
        const int64_t pix = q / C4;
        const int x = (int)(pix % p.W);
        const int64_t rowbase = pix - x;
        const float* gp = p.g + pix * p.g_ld;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), r = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < p.D; ++j) {
            const int i = j * p.stride - p.md;
            const int xs = x + i;
            if (xs >= 0 && xs < p.W) {
                const float gv = gp[p.coff + j];
                const float4 rv = *reinterpret_cast<const float4*>(p.R + (rowbase + xs) * p.r_ld + c4 * 4);
                a.x += gv * rv.x; a.y += gv * rv.y; a.z += gv * rv.z; a.w += gv * rv.w;
            }
            const int xl = x - i;
            if (xl >= 0 && xl < p.W) {
                const float gv = p.g[(rowbase + xl) * p.g_ld + p.coff + j];
                const float4 lv = *reinterpret_cast<const float4*>(p.L + (rowbase + xl) * p.l_ld + c4 * 4);
                r.x += gv * lv.x; r.y += gv * lv.y; r.z += gv * lv.z; r.w += gv * lv.w;
            }
        }
        a.x *= inv_c; a.y *= inv_c; a.z *= inv_c; a.w *= inv_c;
        r.x *= inv_c; r.y *= inv_c; r.z *= inv_c; r.w *= inv_c;
        if (p.copy_left) {
            const float4 gl = *reinterpret_cast<const float4*>(gp + c4 * 4);
            a.x += gl.x; a.y += gl.y; a.z += gl.z; a.w += gl.w;
        }
        float4* dl = reinterpret_cast<float4*>(p.dL + pix * p.dl_ld + c4 * 4);
        if (p.acc_l) { const float4 o = *dl; a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w; }
        *dl = a;
        float4* dr = reinterpret_cast<float4*>(p.dR + pix * p.dr_ld + c4 * 4);
        if (p.acc_r) { const float4 o = *dr; r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w; }
        *dr = r;
        if (p.du && c4 == 0) {
            const float gu = gp[p.coff + p.D];
            p.du[pix] = p.acc_u ? p.du[pix] + gu : gu;
        }
    }
}

// Branch-free form of the gather kernel for small shift counts (D <= DT, tensors under 2 GiB): one lane = one (pixel, 4-channel group), every operand --
// DT gradients of both directions, DT right / left feature vectors, the accumulate operands -- requested through range-checked buffer loads before the
// first product (the loop above loads inside `if (in range)` blocks: 2 D dependent round trips per lane), one workgroup per 256 lanes, no persistent loop.
template <int DT>
__global__ __launch_bounds__(256) void corr_bwd_direct(CorrBwdArgs p) {
    const int C4 = p.C >> 2;
    const float inv_c = 1.0f / (float)p.C;
    const int npix = p.B * p.H * p.W;
    const int q = blockIdx.x * 256 + threadIdx.x;
    const bool live = q < npix * C4;
    const int pix = live ? q / C4 : 0, c4 = live ? q - pix * C4 : 0;
    const int x = pix % p.W;
    const int rowbase = pix - x;
    const __amdgpu_buffer_rsrc_t rs_g = mh_make_rsrc(p.g, (unsigned)((size_t)npix * p.g_ld * 4));
    const __amdgpu_buffer_rsrc_t rs_r = mh_make_rsrc(p.R, (unsigned)(((size_t)npix - 1) * p.r_ld * 4 + (size_t)p.C * 4));
    const __amdgpu_buffer_rsrc_t rs_l = mh_make_rsrc(p.L, (unsigned)(((size_t)npix - 1) * p.l_ld * 4 + (size_t)p.C * 4));
    const __amdgpu_buffer_rsrc_t rs_dl = mh_make_rsrc(p.dL, (unsigned)(((size_t)npix - 1) * p.dl_ld * 4 + (size_t)p.C * 4));
    const __amdgpu_buffer_rsrc_t rs_dr = mh_make_rsrc(p.dR, (unsigned)(((size_t)npix - 1) * p.dr_ld * 4 + (size_t)p.C * 4));
    float gvr[DT], gvl[DT];
    float4 rvv[DT], lvv[DT];
    bool okr[DT], okl[DT];
#pragma unroll
    for (int j = 0; j < DT; ++j) {
        const int i = j * p.stride - p.md;
        const int xs = x + i, xl = x - i;
        okr[j] = live && j < p.D && xs >= 0 && xs < p.W; okl[j] = live && j < p.D && xl >= 0 && xl < p.W;
        gvr[j] = mh_buf_load1(rs_g, okr[j] ? (pix * p.g_ld + p.coff + j) * 4 : MH_OOB);
        gvl[j] = mh_buf_load1(rs_g, okl[j] ? ((rowbase + xl) * p.g_ld + p.coff + j) * 4 : MH_OOB);
        rvv[j] = mh_buf_load4(rs_r, okr[j] ? ((rowbase + xs) * p.r_ld + c4 * 4) * 4 : MH_OOB);
        lvv[j] = mh_buf_load4(rs_l, okl[j] ? ((rowbase + xl) * p.l_ld + c4 * 4) * 4 : MH_OOB);
    }
    const float4 gl = mh_buf_load4(rs_g, (live && p.copy_left) ? (pix * p.g_ld + c4 * 4) * 4 : MH_OOB);
    const float4 ol = mh_buf_load4(rs_dl, (live && p.acc_l) ? (pix * p.dl_ld + c4 * 4) * 4 : MH_OOB);
    const float4 orr = mh_buf_load4(rs_dr, (live && p.acc_r) ? (pix * p.dr_ld + c4 * 4) * 4 : MH_OOB);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < DT; ++j) {
        if (okr[j]) { a.x += gvr[j] * rvv[j].x; a.y += gvr[j] * rvv[j].y; a.z += gvr[j] * rvv[j].z; a.w += gvr[j] * rvv[j].w; }
        if (okl[j]) { r.x += gvl[j] * lvv[j].x; r.y += gvl[j] * lvv[j].y; r.z += gvl[j] * lvv[j].z; r.w += gvl[j] * lvv[j].w; }
    }
    a.x *= inv_c; a.y *= inv_c; a.z *= inv_c; a.w *= inv_c;
    r.x *= inv_c; r.y *= inv_c; r.z *= inv_c; r.w *= inv_c;
    if (p.copy_left) { a.x += gl.x; a.y += gl.y; a.z += gl.z; a.w += gl.w; }
    if (p.acc_l) { a.x += ol.x; a.y += ol.y; a.z += ol.z; a.w += ol.w; }
    if (p.acc_r) { r.x += orr.x; r.y += orr.y; r.z += orr.z; r.w += orr.w; }
    if (!live) return;
    *reinterpret_cast<float4*>(p.dL + (int64_t)pix * p.dl_ld + c4 * 4) = a;
    *reinterpret_cast<float4*>(p.dR + (int64_t)pix * p.dr_ld + c4 * 4) = r;
    if (p.du && c4 == 0) {
        const float gu = p.g[(int64_t)pix * p.g_ld + p.coff + p.D];
        p.du[pix] = p.acc_u ? p.du[pix] + gu : gu;
    }
}

// One level's backward front end in ONE launch: the gradient of the fused cost volume + concat (corr_bwd_kernel with the WARPED right features
// as its right operand) and, straight from registers, the gradient of the warp that made them (ops.hip: warp_bwd_kernel): the gradient w.r.t. the
// warped features is never stored -- it is scattered into the right tower's feature gradient (bilinear taps, fp32 atomics onto a zeroed buffer)
// and contracted with the slope of the interpolation into the coordinate gradient, du = g[disparity channel] + sum_c dRw_c (R[i1] - R[i0]).
// Thread layout of warp_bwd_kernel: LPP lanes per pixel walk the 4-channel groups, du reduced with shuffles.  Same arithmetic as the two
// kernels in sequence (the scatter order differs, as between any two runs of the atomic version).
struct CorrWarpBwdArgs {
    const float* g; const float* L; const float* Rw; const float* img; const float* u;
    float* dL; float* dimg; float* du;
    int g_ld, coff, l_ld, rw_ld, img_ld, dl_ld, dimg_ld;
    int acc_l;
    int acc_img;         // 1: dimg += the gathered / scattered taps (it holds zeros or earlier contributions); 0: dimg is OVERWRITTEN (this launch is its first writer)
    int B, H, W, C, md, stride, D, copy_left;
};
// DT > 0: at most DT shifts, tensors under 2 GiB -- every operand of a channel group (DT correlation gradients of both directions, DT right / left
// feature vectors) is requested through range-checked buffer loads before the first product: the generic loop below loads inside `if (in range)`
// blocks, which the compiler cannot hoist, i.e. 2 DT DEPENDENT memory round trips per channel group (round 4: 13 - 15 us per launch on the 12x40 and
// 24x80 levels, which have microseconds of work).  Same products in the same order (out-of-range shifts add 0).
template <int LPP, int DT = 0>
__global__ __launch_bounds__(256) void corr_warp_bwd_kernel(CorrWarpBwdArgs p) {
    const int C4 = p.C >> 2;
    constexpr int PPB = 256 / LPP;
    const float inv_c = 1.0f / (float)p.C;
    const int sub = threadIdx.x % LPP;
    const int64_t npix = (int64_t)p.B * p.H * p.W;
    const int64_t nit = (npix + PPB - 1) / PPB;
    for (int64_t it = blockIdx.x; it < nit; it += gridDim.x) {
        const int64_t pix = it * PPB + threadIdx.x / LPP;
        const bool live = pix < npix;
        const int64_t pp = live ? pix : 0;
        const int x = (int)(pp % p.W);
        const int64_t rowbase = pp - x;
        const float* gp = p.g + pp * p.g_ld;
        // warp geometry of this pixel (as warp_fwd / warp_bwd_kernel)
        const float cx = (float)x + p.u[pp];
        const float x0 = floorf(cx), x1 = x0 + 1.0f;
        const float xmax = (float)(p.W - 1);
        const float x0s = fminf(fmaxf(x0, 0.f), xmax), x1s = fminf(fmaxf(x1, 0.f), xmax);
        const float m0 = (x0 == x0s) ? 1.f : 0.f, m1 = (x1 == x1s) ? 1.f : 0.f;
        const float w0 = (x1 - cx) * m0, w1 = (cx - x0) * m1;
        const int i0 = (int)x0s, i1 = (int)x1s;
        float dcx = 0.f;
        float gvr[DT > 0 ? DT : 1], gvl[DT > 0 ? DT : 1];
        int orw[DT > 0 ? DT : 1], ol[DT > 0 ? DT : 1];
        __amdgpu_buffer_rsrc_t rs_g, rs_rw, rs_l;
        if constexpr (DT > 0) {
            rs_g = mh_make_rsrc(p.g, (unsigned)(npix * p.g_ld * 4));
            rs_rw = mh_make_rsrc(p.Rw, (unsigned)(npix * p.rw_ld * 4));
            rs_l = mh_make_rsrc(p.L, (unsigned)(npix * p.l_ld * 4));
#pragma unroll
            for (int j = 0; j < DT; ++j) {
                const int i = j * p.stride - p.md;
                const int xs = x + i, xl = x - i;
                const bool okr = live && j < p.D && xs >= 0 && xs < p.W, okl = live && j < p.D && xl >= 0 && xl < p.W;
                gvr[j] = mh_buf_load1(rs_g, okr ? (int)((pp * p.g_ld + p.coff + j) * 4) : MH_OOB);
                gvl[j] = mh_buf_load1(rs_g, okl ? (int)(((rowbase + xl) * p.g_ld + p.coff + j) * 4) : MH_OOB);
                orw[j] = okr ? (int)((rowbase + xs) * p.rw_ld * 4) : MH_OOB;
                ol[j] = okl ? (int)((rowbase + xl) * p.l_ld * 4) : MH_OOB;
            }
        }
        for (int c4 = sub; c4 < C4; c4 += LPP) {
            if (!live) continue;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), r = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (DT > 0) {
                float4 rvv[DT], lvv[DT];
#pragma unroll
                for (int j = 0; j < DT; ++j) {
                    rvv[j] = mh_buf_load4(rs_rw, orw[j] == MH_OOB ? MH_OOB : orw[j] + c4 * 16);
                    lvv[j] = mh_buf_load4(rs_l, ol[j] == MH_OOB ? MH_OOB : ol[j] + c4 * 16);
                }
#pragma unroll
                for (int j = 0; j < DT; ++j) {
                    if (orw[j] != MH_OOB) { a.x += gvr[j] * rvv[j].x; a.y += gvr[j] * rvv[j].y; a.z += gvr[j] * rvv[j].z; a.w += gvr[j] * rvv[j].w; }
                    if (ol[j] != MH_OOB) { r.x += gvl[j] * lvv[j].x; r.y += gvl[j] * lvv[j].y; r.z += gvl[j] * lvv[j].z; r.w += gvl[j] * lvv[j].w; }
                }
            } else
            for (int j = 0; j < p.D; ++j) {
                const int i = j * p.stride - p.md;
                const int xs = x + i;
                if (xs >= 0 && xs < p.W) {
                    const float gv = gp[p.coff + j];
                    const float4 rv = *reinterpret_cast<const float4*>(p.Rw + (rowbase + xs) * p.rw_ld + c4 * 4);
                    a.x += gv * rv.x; a.y += gv * rv.y; a.z += gv * rv.z; a.w += gv * rv.w;
                }
                const int xl = x - i;
                if (xl >= 0 && xl < p.W) {
                    const float gv = p.g[(rowbase + xl) * p.g_ld + p.coff + j];
                    const float4 lv = *reinterpret_cast<const float4*>(p.L + (rowbase + xl) * p.l_ld + c4 * 4);
                    r.x += gv * lv.x; r.y += gv * lv.y; r.z += gv * lv.z; r.w += gv * lv.w;
                }
            }
            a.x *= inv_c; a.y *= inv_c; a.z *= inv_c; a.w *= inv_c;
            r.x *= inv_c; r.y *= inv_c; r.z *= inv_c; r.w *= inv_c;
            if (p.copy_left) {
                const float4 gl = *reinterpret_cast<const float4*>(gp + c4 * 4);
                a.x += gl.x; a.y += gl.y; a.z += gl.z; a.w += gl.w;
            }
            float4* dl = reinterpret_cast<float4*>(p.dL + pp * p.dl_ld + c4 * 4);
            if (p.acc_l) { const float4 o = *dl; a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w; }
            *dl = a;
            // r = gradient w.r.t. the warped right features at this pixel: scatter it to its two source pixels ...
            if (p.dimg) {
                float* d0 = p.dimg + (rowbase + i0) * p.dimg_ld + c4 * 4;
                float* d1 = p.dimg + (rowbase + i1) * p.dimg_ld + c4 * 4;
                if (w0 != 0.f) { mh_atomic_add(d0 + 0, w0 * r.x); mh_atomic_add(d0 + 1, w0 * r.y); mh_atomic_add(d0 + 2, w0 * r.z); mh_atomic_add(d0 + 3, w0 * r.w); }
                if (w1 != 0.f) { mh_atomic_add(d1 + 0, w1 * r.x); mh_atomic_add(d1 + 1, w1 * r.y); mh_atomic_add(d1 + 2, w1 * r.z); mh_atomic_add(d1 + 3, w1 * r.w); }
            }
            // ... and contract it with the slope of the interpolation
            if (p.du) {
                const float4 s0 = *reinterpret_cast<const float4*>(p.img + (rowbase + i0) * p.img_ld + c4 * 4);
                const float4 s1 = *reinterpret_cast<const float4*>(p.img + (rowbase + i1) * p.img_ld + c4 * 4);
                dcx += r.x * (m1 * s1.x - m0 * s0.x) + r.y * (m1 * s1.y - m0 * s0.y) + r.z * (m1 * s1.z - m0 * s0.z) + r.w * (m1 * s1.w - m0 * s0.w);
            }
        }
        if (p.du) {
#pragma unroll
            for (int o = LPP >> 1; o > 0; o >>= 1) dcx += __shfl_xor(dcx, o);
            if (live && sub == 0) p.du[pp] = gp[p.coff + p.D] + dcx;
        }
    }
}

// Row-owned form of the same launch (round 5): ONE workgroup owns ONE image row.  The warp of _linear_warping (MadNet.py:400-436) moves pixels along x
// only, so every bilinear tap of a row's gradient lands in the SAME row of the right tower's feature gradient: the scatter runs on an LDS copy of that
// row (ds_add, W x C accumulators) and leaves as plain 16-byte read-modify-write stores -- no global atomics (the atomic form sends 2 x C device-scope
// fp32 atomics per pixel to the memory side: 2 M at 96x320x32, 34.6 MB of WRITE_SIZE for a 3.9 MB result, 89 % of the wave cycles waiting:
// profiles/r04_pmc_roofline.json).  Round 3 rejected a first row kernel (r03_experiments.txt #12) because its walk loaded inside `if (in range)` blocks:
// 2 D dependent memory round trips per channel group on 96 workgroups.  Here the u row is staged first (its taps address the slope loads), then
// every operand of a channel group -- DT gradients of both directions, DT warped-right / left vectors, the two slope taps, the accumulate operand -- is
// requested through range-checked buffer loads before the first product: two round trips per pass whatever D is.
//   DET = false: fp32 LDS atomics (summation order varies from run to run at rounding level, like the global-atomic form);
//   DET = true (a deterministic range is registered, mh_deterministic_add): 64-bit fixed-point LDS accumulators (value * 2^48, integer adds are
//   associative): bit-identical from run to run WITHOUT the global fixed-point twin and its flush launch.
template <int LPP, int DT, bool DET>
__global__ __launch_bounds__(1024) void corr_warp_bwd_row_kernel(CorrWarpBwdArgs p) {
    HIP_DYNAMIC_SHARED(float, smem)
    constexpr int NT = 1024, PPB = NT / LPP;
    using acc_t = typename std::conditional<DET, unsigned long long, float>::type;
    acc_t* const racc = reinterpret_cast<acc_t*>(smem);                        // [W][C]
    float* const su = smem + (size_t)p.W * p.C * (DET ? 2 : 1);               // [W]
    const int tid = threadIdx.x, sub = tid % LPP;
    const int C4 = p.C >> 2;
    const float inv_c = 1.0f / (float)p.C;
    const int row = blockIdx.x;                                               // b*H + y
    const int rowbase = row * p.W;
    const int npix = p.B * p.H * p.W;
    const __amdgpu_buffer_rsrc_t rs_g = mh_make_rsrc(p.g, (unsigned)((size_t)npix * p.g_ld * 4));
    const __amdgpu_buffer_rsrc_t rs_rw = mh_make_rsrc(p.Rw, (unsigned)((size_t)npix * p.rw_ld * 4));
    const __amdgpu_buffer_rsrc_t rs_l = mh_make_rsrc(p.L, (unsigned)((size_t)npix * p.l_ld * 4));
    const __amdgpu_buffer_rsrc_t rs_img = mh_make_rsrc(p.img, (unsigned)((size_t)npix * p.img_ld * 4));
    const __amdgpu_buffer_rsrc_t rs_dl = mh_make_rsrc(p.dL, (unsigned)((size_t)npix * p.dl_ld * 4));
    // fp32 form: the LDS row starts from the row's previous content (requested with the u row: one round trip) and leaves as plain stores;
    // fixed-point form: starts from zero, the previous content is added at the flush (its conversion would round what an earlier launch left)
    const int nq = p.W * C4;
    if constexpr (DET) {
        for (int i = tid; i < p.W * p.C; i += NT) racc[i] = 0ull;
    } else {
        for (int q = tid; q < nq; q += NT) {
            const int x = q / C4, c4 = q - x * C4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.dimg) v = *reinterpret_cast<const float4*>(p.dimg + (int64_t)(rowbase + x) * p.dimg_ld + c4 * 4);
            *reinterpret_cast<float4*>(racc + x * p.C + c4 * 4) = v;
        }
    }
    for (int x = tid; x < p.W; x += NT) su[x] = p.u[rowbase + x];
    __syncthreads();
    for (int xb = 0; xb < p.W; xb += PPB) {
        const int x = xb + tid / LPP;
        const bool live = x < p.W;
        const int pp = rowbase + (live ? x : 0);
        const float cx = (float)x + su[live ? x : 0];
        const float x0 = floorf(cx), x1 = x0 + 1.0f;
        const float xmax = (float)(p.W - 1);
        const float x0s = fminf(fmaxf(x0, 0.f), xmax), x1s = fminf(fmaxf(x1, 0.f), xmax);
        const float m0 = (x0 == x0s) ? 1.f : 0.f, m1 = (x1 == x1s) ? 1.f : 0.f;
        const float w0 = (x1 - cx) * m0, w1 = (cx - x0) * m1;
        const int i0 = (int)x0s, i1 = (int)x1s;
        float gvr[DT], gvl[DT];
        int orw[DT], ol[DT];
#pragma unroll
        for (int j = 0; j < DT; ++j) {
            const int i = j * p.stride - p.md;
            const int xs = x + i, xl = x - i;
            const bool okr = live && j < p.D && xs >= 0 && xs < p.W, okl = live && j < p.D && xl >= 0 && xl < p.W;
            gvr[j] = mh_buf_load1(rs_g, okr ? (pp * p.g_ld + p.coff + j) * 4 : MH_OOB);
            gvl[j] = mh_buf_load1(rs_g, okl ? ((rowbase + xl) * p.g_ld + p.coff + j) * 4 : MH_OOB);
            orw[j] = okr ? (rowbase + xs) * p.rw_ld * 4 : MH_OOB;
            ol[j] = okl ? (rowbase + xl) * p.l_ld * 4 : MH_OOB;
        }
        const float gu = mh_buf_load1(rs_g, (live && p.du) ? (pp * p.g_ld + p.coff + p.D) * 4 : MH_OOB);
        float dcx = 0.f;
        for (int c4 = sub; c4 < C4; c4 += LPP) {
            float4 rvv[DT], lvv[DT];
#pragma unroll
            for (int j = 0; j < DT; ++j) {
                rvv[j] = mh_buf_load4(rs_rw, orw[j] == MH_OOB ? MH_OOB : orw[j] + c4 * 16);
                lvv[j] = mh_buf_load4(rs_l, ol[j] == MH_OOB ? MH_OOB : ol[j] + c4 * 16);
            }
            const float4 gl = mh_buf_load4(rs_g, (live && p.copy_left) ? (pp * p.g_ld + c4 * 4) * 4 : MH_OOB);
            const float4 dlo = mh_buf_load4(rs_dl, (live && p.acc_l) ? (pp * p.dl_ld + c4 * 4) * 4 : MH_OOB);
            const float4 s0 = mh_buf_load4(rs_img, (live && p.du) ? ((rowbase + i0) * p.img_ld + c4 * 4) * 4 : MH_OOB);
            const float4 s1 = mh_buf_load4(rs_img, (live && p.du) ? ((rowbase + i1) * p.img_ld + c4 * 4) * 4 : MH_OOB);
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < DT; ++j) {
                if (orw[j] != MH_OOB) { a.x += gvr[j] * rvv[j].x; a.y += gvr[j] * rvv[j].y; a.z += gvr[j] * rvv[j].z; a.w += gvr[j] * rvv[j].w; }
                if (ol[j] != MH_OOB) { r.x += gvl[j] * lvv[j].x; r.y += gvl[j] * lvv[j].y; r.z += gvl[j] * lvv[j].z; r.w += gvl[j] * lvv[j].w; }
            }
            a.x *= inv_c; a.y *= inv_c; a.z *= inv_c; a.w *= inv_c;
            r.x *= inv_c; r.y *= inv_c; r.z *= inv_c; r.w *= inv_c;
            a.x += gl.x; a.y += gl.y; a.z += gl.z; a.w += gl.w;            // (zeros unless copy_left / acc_l: out-of-range loads)
            a.x += dlo.x; a.y += dlo.y; a.z += dlo.z; a.w += dlo.w;
            if (live) {
                *reinterpret_cast<float4*>(p.dL + (int64_t)pp * p.dl_ld + c4 * 4) = a;
                if (p.dimg) {
                    acc_t* d0 = racc + i0 * p.C + c4 * 4;
                    acc_t* d1 = racc + i1 * p.C + c4 * 4;
                    const float t0[4] = {w0 * r.x, w0 * r.y, w0 * r.z, w0 * r.w}, t1[4] = {w1 * r.x, w1 * r.y, w1 * r.z, w1 * r.w};
                    if constexpr (DET) {
#pragma unroll 1
                        for (int e = 0; e < 4; ++e) {          // (rolled: the 64-bit conversions of an unrolled body spill at 128 VGPRs)
                            if (w0 != 0.f) atomicAdd(d0 + e, (unsigned long long)(long long)llrintf(t0[e] * 281474976710656.0f));
                            if (w1 != 0.f) atomicAdd(d1 + e, (unsigned long long)(long long)llrintf(t1[e] * 281474976710656.0f));
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (w0 != 0.f) atomicAdd(d0 + e, t0[e]);
                            if (w1 != 0.f) atomicAdd(d1 + e, t1[e]);
                        }
                    }
                }
            }
            dcx += r.x * (m1 * s1.x - m0 * s0.x) + r.y * (m1 * s1.y - m0 * s0.y) + r.z * (m1 * s1.z - m0 * s0.z) + r.w * (m1 * s1.w - m0 * s0.w);
        }
        if (p.du) {
#pragma unroll
            for (int o = LPP >> 1; o > 0; o >>= 1) dcx += __shfl_xor(dcx, o);
            if (live && sub == 0) p.du[pp] = gu + dcx;
        }
    }
    if (!p.dimg) return;
    __syncthreads();
    // the row of the right tower's feature gradient: += what the row's taps left in LDS (this workgroup is the row's only writer)
    for (int q = tid; q < nq; q += NT) {
        const int x = q / C4, c4 = q - x * C4;
        float4* dst = reinterpret_cast<float4*>(p.dimg + (int64_t)(rowbase + x) * p.dimg_ld + c4 * 4);
        const acc_t* s = racc + x * p.C + c4 * 4;
        if constexpr (DET) {
            float4 v = *dst;
            v.x += (float)((double)(long long)s[0] * (1.0 / 281474976710656.0)); v.y += (float)((double)(long long)s[1] * (1.0 / 281474976710656.0));
            v.z += (float)((double)(long long)s[2] * (1.0 / 281474976710656.0)); v.w += (float)((double)(long long)s[3] * (1.0 / 281474976710656.0));
            *dst = v;
        } else {
            *dst = *reinterpret_cast<const float4*>(s);
        }
    }
}

// The row-owned form with the row's operands STAGED in LDS and the scatter turned into a GATHER (the default path, round 5).
// Measured on the MI355X (profiles/r05_experiments.txt #2, #3): both scatter forms above take ~35 us per row whatever the level -- 96 workgroups at 96x320x32
// in the replayed step are no faster than the 2 M global atomics (1.457 vs 1.426 ms per step), and 6144 rows on 256 CUs (B = 64) run at 35 us per row and CU:
// the time is the row's 20 k ds_add_f32 lane operations (4-way bank conflicts between the pixel groups of a wave), not its memory traffic.  So:
//   * ONE round trip brings the row in: the left and the warped-right feature rows, the D + 1 correlation / disparity channels of g and the u row go to LDS
//     (each element read once, coalesced), the per-item operands that do not depend on u (the concat-copy part of g, the accumulate operand of dL) to
//     registers; a second round trip fetches the two slope taps of every item (their address needs u).  The 2 DT shifted products read LDS;
//   * the gradient w.r.t. the warped features r(x) of every pixel is stored to LDS (plain 16-byte stores, every item its own slot); the row's taps are
//     bucketed by the source column they land on: two INTEGER LDS atomics per pixel (the float scatter needed 2 C per pixel) count the column's taps and
//     hand out a slot of its 8-entry list; only when some column overflows its list (a compressed stretch of the warp) the taps are bucketed again by a
//     counting sort without capacity (one wave scans the W counters, two more atomics per pixel place the keys).  Every (source column, channel group)
//     item then GATHERS its taps: typically two, one 16-byte read and four FMAs each.  Segments of up to 16 taps are summed in ascending
//     key order whatever order they were placed in (sorting network up to 8, selection beyond): bit-identical from run to run -- no float atomics, no
//     fixed-point twin, the same kernel serves the deterministic mode; only a fold of more than 16 taps onto one column is summed in arrival order.
// A thread owns at most NI = 3 (pixel, channel group) items: LPP >= C / 4 lanes per pixel.
template <int LPP, int DT>
__global__ __launch_bounds__(1024) void corr_warp_bwd_rowlds_kernel(CorrWarpBwdArgs p) {
    HIP_DYNAMIC_SHARED(float, smem)
    constexpr int NT = 1024, PPB = NT / LPP, NI = 3, GS = 8, KL = 8;
    static_assert(DT + 1 <= GS, "g staging holds D + 1 channels");
    const int C4 = p.C >> 2, WC = p.W * p.C;
    float* const sL = smem;                     // [W][C]
    float* const sR = smem + WC;                // [W][C]   warped right features
    float* const rb = smem + 2 * WC;            // [W][C]   gradient w.r.t. the warped right features
    float* const sg = smem + 3 * WC;            // [W][GS]  g[coff .. coff + D]
    float* const su = sg + p.W * GS;            // [W]
    float* const sw0 = su + p.W;                // [W] tap weights (0 where the tap is masked) and tap columns of every pixel
    float* const sw1 = sw0 + p.W;
    int* const cnt = reinterpret_cast<int*>(sw1 + p.W);          // [W]      taps that land on a source column (counted, then counted down by the placement)
    int* const base = cnt + p.W;                                 // [W + 1]  exclusive prefix sum of cnt: the column's segment of ent
    int* const ent_all = base + p.W + 1;                         // [2 W]    (pixel << 1) | tap, bucketed by source column (counting sort: only when a column overflows its list)
    int* const lst = ent_all + 2 * p.W;                              // [W][KL]  the first KL taps of every column, placed while they are counted
    int* const over = lst + p.W * KL;                            // [1]      some column received more than KL taps
    const int tid = threadIdx.x, sub = tid % LPP;
    const float inv_c = 1.0f / (float)p.C;
    const int row = blockIdx.x;
    const int rowbase = row * p.W;
    const int npix = p.B * p.H * p.W;
    const __amdgpu_buffer_rsrc_t rs_g = mh_make_rsrc(p.g, (unsigned)((size_t)npix * p.g_ld * 4));
    const __amdgpu_buffer_rsrc_t rs_rw = mh_make_rsrc(p.Rw, (unsigned)((size_t)npix * p.rw_ld * 4));
    const __amdgpu_buffer_rsrc_t rs_l = mh_make_rsrc(p.L, (unsigned)((size_t)npix * p.l_ld * 4));
    const __amdgpu_buffer_rsrc_t rs_img = mh_make_rsrc(p.img, (unsigned)((size_t)npix * p.img_ld * 4));
    const __amdgpu_buffer_rsrc_t rs_dl = mh_make_rsrc(p.dL, (unsigned)((size_t)npix * p.dl_ld * 4));
    const __amdgpu_buffer_rsrc_t rs_di = mh_make_rsrc(p.dimg ? p.dimg : p.dL, (unsigned)((size_t)npix * (p.dimg ? p.dimg_ld : p.dl_ld) * 4));
    // ---- round trip 1: everything whose address does not depend on u ------------------------------------------------------------------------------
    const int nq = p.W * C4;
    float4 vl[NI], vr[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int q = tid + k * NT;
        const int x = q / C4, c4 = q - x * C4;
        const bool ok = q < nq;
        vl[k] = mh_buf_load4(rs_l, ok ? ((rowbase + x) * p.l_ld + c4 * 4) * 4 : MH_OOB);
        vr[k] = mh_buf_load4(rs_rw, ok ? ((rowbase + x) * p.rw_ld + c4 * 4) * 4 : MH_OOB);
    }
    float vg[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {              // W * GS <= NI * NT (host check)
        const int q = tid + k * NT;
        const int x = q / GS, j = q - x * GS;
        vg[k] = mh_buf_load1(rs_g, (x < p.W && j <= p.D) ? ((rowbase + x) * p.g_ld + p.coff + j) * 4 : MH_OOB);
    }
    const float vu = (tid < p.W) ? p.u[rowbase + tid] : 0.f;        // W <= NT (host check)
    float4 gl[NI], dlo[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int x = k * PPB + tid / LPP;
        const bool ok = x < p.W && sub < C4;
        const int pp = rowbase + x;
        gl[k] = mh_buf_load4(rs_g, (ok && p.copy_left) ? (pp * p.g_ld + sub * 4) * 4 : MH_OOB);
        dlo[k] = mh_buf_load4(rs_dl, (ok && p.acc_l) ? (pp * p.dl_ld + sub * 4) * 4 : MH_OOB);
    }
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int q = tid + k * NT;
        if (q < nq) {
            const int x = q / C4, c4 = q - x * C4;
            *reinterpret_cast<float4*>(sL + x * p.C + c4 * 4) = vl[k];
            *reinterpret_cast<float4*>(sR + x * p.C + c4 * 4) = vr[k];
        }
    }
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int q = tid + k * NT;
        if (q < p.W * GS) sg[q] = vg[k];
    }
    if (tid < p.W) { su[tid] = vu; cnt[tid] = 0; }
    if (tid == 0) over[0] = 0;
    __syncthreads();
    // ---- round trip 2: the slope taps (right features at the two source columns of every pixel's warp) ------------------------------------------------
    float4 s0[NI], s1[NI];
    float m0[NI], m1[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int x = k * PPB + tid / LPP;
        const bool live = x < p.W && sub < C4;
        const float cx = (float)x + su[x < p.W ? x : 0];
        const float x0 = floorf(cx), x1 = x0 + 1.0f;
        const float xmax = (float)(p.W - 1);
        const float x0s = fminf(fmaxf(x0, 0.f), xmax), x1s = fminf(fmaxf(x1, 0.f), xmax);
        m0[k] = (x0 == x0s) ? 1.f : 0.f; m1[k] = (x1 == x1s) ? 1.f : 0.f;
        const int i0 = (int)x0s, i1 = (int)x1s;
        s0[k] = mh_buf_load4(rs_img, (live && p.du) ? ((rowbase + i0) * p.img_ld + sub * 4) * 4 : MH_OOB);
        s1[k] = mh_buf_load4(rs_img, (live && p.du) ? ((rowbase + i1) * p.img_ld + sub * 4) * 4 : MH_OOB);
        if (x < p.W && sub == 0 && p.dimg) {
            // the pixel's two taps are counted at their source columns: integer LDS atomics, two per PIXEL
            const float w0 = (x1 - cx) * m0[k], w1 = (cx - x0) * m1[k];
            sw0[x] = w0; sw1[x] = w1;
            if (w0 != 0.f) { const int sl = atomicAdd(cnt + i0, 1); if (sl < KL) lst[i0 * KL + sl] = x << 1; else over[0] = 1; }
            if (w1 != 0.f) { const int sl = atomicAdd(cnt + i1, 1); if (sl < KL) lst[i1 * KL + sl] = (x << 1) | 1; else over[0] = 1; }
        }
    }
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int x = k * PPB + tid / LPP;
        const bool live = x < p.W && sub < C4;
        const int xq = x < p.W ? x : 0;
        const int pp = rowbase + xq;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), r = make_float4(0.f, 0.f, 0.f, 0.f);
        const int cc = (sub < C4 ? sub : 0) * 4;
#pragma unroll
        for (int j = 0; j < DT; ++j) {
            const int i = j * p.stride - p.md;
            const int xs = xq + i, xl = xq - i;
            const bool okr = j < p.D && xs >= 0 && xs < p.W, okl = j < p.D && xl >= 0 && xl < p.W;
            const int xsc = okr ? xs : xq, xlc = okl ? xl : xq;
            const float gr = sg[xq * GS + j], gq = sg[xlc * GS + j];
            const float4 rv = *reinterpret_cast<const float4*>(sR + xsc * p.C + cc);
            const float4 lv = *reinterpret_cast<const float4*>(sL + xlc * p.C + cc);
            if (okr) { a.x += gr * rv.x; a.y += gr * rv.y; a.z += gr * rv.z; a.w += gr * rv.w; }
            if (okl) { r.x += gq * lv.x; r.y += gq * lv.y; r.z += gq * lv.z; r.w += gq * lv.w; }
        }
        a.x *= inv_c; a.y *= inv_c; a.z *= inv_c; a.w *= inv_c;
        r.x *= inv_c; r.y *= inv_c; r.z *= inv_c; r.w *= inv_c;
        if (p.copy_left) { a.x += gl[k].x; a.y += gl[k].y; a.z += gl[k].z; a.w += gl[k].w; }
        if (p.acc_l) { a.x += dlo[k].x; a.y += dlo[k].y; a.z += dlo[k].z; a.w += dlo[k].w; }
        float dcx = 0.f;
        if (live) {
            *reinterpret_cast<float4*>(p.dL + (int64_t)pp * p.dl_ld + cc) = a;
            *reinterpret_cast<float4*>(rb + xq * p.C + cc) = r;
            dcx = r.x * (m1[k] * s1[k].x - m0[k] * s0[k].x) + r.y * (m1[k] * s1[k].y - m0[k] * s0[k].y) + r.z * (m1[k] * s1[k].z - m0[k] * s0[k].z) +
                  r.w * (m1[k] * s1[k].w - m0[k] * s0[k].w);
        }
        if (p.du) {
#pragma unroll
            for (int o = LPP >> 1; o > 0; o >>= 1) dcx += __shfl_xor(dcx, o);
            if (x < p.W && sub == 0) p.du[pp] = sg[xq * GS + p.D] + dcx;
        }
    }
    if (!p.dimg) return;
    // ---- the row of the right tower's gradient: previous content (requested now, used behind the barriers) + the gathered taps -----------------------------
    float4 vd[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int q = tid + k * NT;
        const int x = q / C4, c4 = q - x * C4;
        vd[k] = mh_buf_load4(rs_di, (q < nq && p.acc_img) ? ((rowbase + x) * p.dimg_ld + c4 * 4) * 4 : MH_OOB);      // (first writer: zeros, nothing read)
    }
    __syncthreads();
    // The common case: no column received more than KL taps -- every tap already sits in its column's list.  Otherwise (a compressed stretch of the
    // warp) the taps are bucketed again without a capacity: counting sort -- one wave scans the W counters, every tap takes a slot of its column's segment.
    const bool spill = over[0] != 0;
    if (spill) {
        if (tid < 64) {
            // exclusive prefix sum of the W column counts by ONE wave: lane l owns columns [l * per, (l + 1) * per)
            const int per = (p.W + 63) >> 6;
            int sum = 0;
            for (int i = 0; i < per; ++i) { const int x = tid * per + i; if (x < p.W) sum += cnt[x]; }
            int inc = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (tid >= o) inc += t; }
            int run = inc - sum;
            for (int i = 0; i < per; ++i) { const int x = tid * per + i; if (x < p.W) { base[x] = run; run += cnt[x]; } }
            if (tid == 63) base[p.W] = inc;
        }
        __syncthreads();
        // placement, counting the column's counter down: the segment fills from its end
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int x = k * PPB + tid / LPP;
            if (x < p.W && sub == 0) {
                const float cx = (float)x + su[x];
                const float x0 = floorf(cx);
                const float xmax = (float)(p.W - 1);
                const int i0 = (int)fminf(fmaxf(x0, 0.f), xmax), i1 = (int)fminf(fmaxf(x0 + 1.0f, 0.f), xmax);
                if (sw0[x] != 0.f) ent_all[base[i0] + atomicAdd(cnt + i0, -1) - 1] = x << 1;
                if (sw1[x] != 0.f) ent_all[base[i1] + atomicAdd(cnt + i1, -1) - 1] = (x << 1) | 1;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int q = tid + k * NT;
        if (q >= nq) continue;
        const int xs = q / C4, c4 = q - xs * C4;
        float4 acc = vd[k];
        const int* const ent = spill ? ent_all + base[xs] : lst + xs * KL;
        const int b0 = 0, n = spill ? base[xs + 1] - base[xs] : cnt[xs];
        auto add = [&](int key) {
            const int x = key >> 1;
            const float w = (key & 1) ? sw1[x] : sw0[x];
            const float4 rv = *reinterpret_cast<const float4*>(rb + x * p.C + c4 * 4);
            acc.x += w * rv.x; acc.y += w * rv.y; acc.z += w * rv.z; acc.w += w * rv.w;
        };
        if (n <= KL) {
            // the column's taps in ascending (pixel, tap) order whatever order they were placed in: the sum is bit-identical from run to run
            int e[KL];
#pragma unroll
            for (int i = 0; i < KL; ++i) e[i] = i < n ? ent[b0 + i] : 0x7fffffff;
            if (n > 2) {
#define MH_CSWAP(i, j) { const int lo_ = min(e[i], e[j]), hi_ = max(e[i], e[j]); e[i] = lo_; e[j] = hi_; }
                MH_CSWAP(0, 1) MH_CSWAP(2, 3) MH_CSWAP(4, 5) MH_CSWAP(6, 7) MH_CSWAP(0, 2) MH_CSWAP(1, 3) MH_CSWAP(4, 6) MH_CSWAP(5, 7) MH_CSWAP(1, 2) MH_CSWAP(5, 6)
                MH_CSWAP(0, 4) MH_CSWAP(1, 5) MH_CSWAP(2, 6) MH_CSWAP(3, 7) MH_CSWAP(2, 4) MH_CSWAP(3, 5) MH_CSWAP(1, 2) MH_CSWAP(3, 4) MH_CSWAP(5, 6)
#undef MH_CSWAP
            } else if (n == 2 && e[1] < e[0]) { const int t = e[0]; e[0] = e[1]; e[1] = t; }
#pragma unroll
            for (int i = 0; i < KL; ++i)
                if (i < n) add(e[i]);
        } else {
            // a compressed stretch of the warp: selection, smallest key first (n^2 LDS reads of the segment).  No cap: a fold of many taps onto
            // one column -- a diverged disparity map -- is slow here (n <= 2 W), never summed in arrival order: the launch has no order-dependent arithmetic.
            int last = -1;
            for (int t = 0; t < n; ++t) {
                int best = 0x7fffffff;
                for (int i = 0; i < n; ++i) { const int kq = ent[b0 + i]; if (kq > last && kq < best) best = kq; }
                add(best);
                last = best;
            }
        }
        *reinterpret_cast<float4*>(p.dimg + (int64_t)(rowbase + xs) * p.dimg_ld + c4 * 4) = acc;
    }
}

// Large shift counts (DispNet, D = 81) on the matrix cores: out[x][d] = mean_c L[x][c] * R[x + d - md][c] is the band
// |x' - x| <= md of the row-wise product L_row (W x C) * R_row^T (C x W).  One workgroup = one 64-pixel row segment,
// one wave = 16 pixels; the right-feature window [x0 - md, x0 + 64 + md) is staged once in LDS (k-contiguous rows,
// +4 floats of padding), the wave's left tile stays in registers as MFMA A operands, and the wave walks the
// ceil((2 md + 16) / 16) 16-column blocks of its band with exact-fp32 v_mfma_f32_16x16x4_f32 (each lane feeds 4 MFMAs
// from one ds_read_b128: the k permutation k = s*16 + (lane>>4)*4 + t is the same for both operands).  Band overhead
// (2md+16)/(2md+1) = 1.19x of the useful MACs; the kernel is HBM-bound on L + R + the D-channel output.
template <int CS16>     // C / 16 (compile time so the A tile stays in registers)
__global__ __launch_bounds__(256) void corr_fwd_mfma(CorrArgs p) {
    HIP_DYNAMIC_SHARED(float, smem)                   // [rows][C + 4]
    constexpr int C = CS16 * 16, RS = C + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int seg = blockIdx.x % p.segs;
    const int row = blockIdx.x / p.segs;              // b*H + y
    const int x0 = seg * 64;
    const int nb = (2 * p.md + 16 + 15) / 16;         // 16-column blocks per wave
    const int rows = 48 + 16 * nb;                    // window rows any wave touches
    const __amdgpu_buffer_rsrc_t rsL = mh_make_rsrc(p.L, p.l_bytes);
    const __amdgpu_buffer_rsrc_t rsR = mh_make_rsrc(p.R, p.r_bytes);
    // stage the right window (zeros outside the image row = correlation_tf's zero padding)
    for (int q = tid; q < rows * (C / 4); q += 256) {
        const int xw = q / (C / 4), c4 = q - xw * (C / 4);
        const int xs = x0 - p.md + xw;
        const bool ok = xs >= 0 && xs < p.W;
        const float4 v = mh_buf_load4(rsR, ok ? ((row * p.W + xs) * p.r_ld + c4 * 4) * 4 : MH_OOB);
        *reinterpret_cast<float4*>(smem + xw * RS + c4 * 4) = v;
    }
    // this wave's left tile: pixel xb + li, channels s*16 + lq*4 .. +3
    const int xb = x0 + 16 * wave;
    float4 a[CS16];
    {
        const bool ok = xb + li < p.W;
#pragma unroll
        for (int s = 0; s < CS16; ++s)
            a[s] = mh_buf_load4(rsL, ok ? ((row * p.W + xb + li) * p.l_ld + s * 16 + lq * 4) * 4 : MH_OOB);
    }
    __syncthreads();
    const float inv_c = 1.0f / (float)C;
    for (int blk = 0; blk < nb; ++blk) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* Rb = smem + (16 * wave + 16 * blk + li) * RS + lq * 4;
#pragma unroll
        for (int s = 0; s < CS16; ++s) {
            const float4 b = *reinterpret_cast<const float4*>(Rb + s * 16);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s].x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s].y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s].z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s].w, b.w, acc, 0, 0, 0);
        }
        // acc[r]: row = pixel xb + 4*lq + r, column = window pixel 16*wave + 16*blk + li  ->  shift index d
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int x = xb + 4 * lq + r;
            const int d = 16 * blk + li - (4 * lq + r);
            if (x < p.W && d >= 0 && d < p.D) p.out[(int64_t)(row * p.W + x) * p.out_ld + p.coff + d] = acc[r] * inv_c;
        }
    }
}

// The same band on the bf16 matrix cores (precision codes 1 / 2 of mh_corr_fwd_prec).  The exact-fp32 form above runs
// 6 x C/4 = 192 v_mfma_f32_16x16x4_f32 per wave (6144 cycles: at 31 FLOP/B the fp32 MFMA rate -- 157 TFLOP/s -- is the ceiling
// it hits at 32 % of the HBM peak); with bf16 operands (v_mfma_f32_16x16x32_bf16, fp32 accumulate) the same band costs
// 6 x C/32 = 24 MFMAs (408 cycles) and the kernel becomes what the op is: a stream of L + R + the D-channel output.
//   X3 = false: operands rounded to bf16 (RNE);  X3 = true: split-bf16, hi + lo planes, 3 MFMAs per product (~2^-16 relative).
// Window rows are bf16 [pixel][C + 16 halfs] (the 8-dword-mod-16 row stride of conv_patch.hip: conflict-free b128 reads).
template <int CS32, bool X3>    // C / 32
__global__ __launch_bounds__(256) void corr_fwd_mfma_bf16(CorrArgs p) {
    HIP_DYNAMIC_SHARED(float, smem)
    constexpr int C = CS32 * 32, RS = C + 16;            // halfs
    unsigned short* const Wh = reinterpret_cast<unsigned short*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int bid = p.remap ? mh_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int seg = bid % p.segs;
    const int row = bid / p.segs;                     // b*H + y
    const int x0 = seg * 64;
    const int nb = (2 * p.md + 16 + 15) / 16;         // 16-column blocks per wave
    const int rows = 48 + 16 * nb;                    // window rows any wave touches
    const int PLO = rows * RS;                        // lo plane offset (X3)
    const __amdgpu_buffer_rsrc_t rsL = mh_make_rsrc(p.L, p.l_bytes);
    const __amdgpu_buffer_rsrc_t rsR = mh_make_rsrc(p.R, p.r_bytes);
    // stage the right window: U independent 16-byte loads in flight per thread, converted at the LDS store
    constexpr int U = 8;
    const int items = rows * (C / 4);
    for (int q0 = tid; q0 < items; q0 += 256 * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = q0 + 256 * u;
            const int xw = q / (C / 4), c4 = q - xw * (C / 4);
            const int xs = x0 - p.md + xw;
            const bool ok = q < items && xs >= 0 && xs < p.W;
            v[u] = mh_buf_load4(rsR, ok ? ((row * p.W + xs) * p.r_ld + c4 * 4) * 4 : MH_OOB);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = q0 + 256 * u;
            if (q < items) {
                const int xw = q / (C / 4), c4 = q - xw * (C / 4);
                unsigned short* d = Wh + xw * RS + c4 * 4;
                if constexpr (X3) {
                    uint2 hi, lo;
                    mh_split_bf16x2(v[u].x, v[u].y, hi.x, lo.x);
                    mh_split_bf16x2(v[u].z, v[u].w, hi.y, lo.y);
                    *reinterpret_cast<uint2*>(d) = hi;
                    *reinterpret_cast<uint2*>(d + PLO) = lo;
                } else {
                    *reinterpret_cast<uint2*>(d) = make_uint2(mh_pack_bf16(v[u].x, v[u].y), mh_pack_bf16(v[u].z, v[u].w));
                }
            }
        }
    }
    // this wave's left tile as MFMA A operands: pixel xb + li, channels s*32 + lq*8 .. +7
    const int xb = x0 + 16 * wave;
    u32x4 a[CS32], al[X3 ? CS32 : 1];
    {
        const bool ok = xb + li < p.W;
#pragma unroll
        for (int s = 0; s < CS32; ++s) {
            const int off = ok ? ((row * p.W + xb + li) * p.l_ld + s * 32 + lq * 8) * 4 : MH_OOB;
            const float4 v0 = mh_buf_load4(rsL, off), v1 = mh_buf_load4(rsL, off == MH_OOB ? MH_OOB : off + 16);
            if constexpr (X3) {
                unsigned h0, h1, h2, h3, l0, l1, l2, l3;
                mh_split_bf16x2(v0.x, v0.y, h0, l0); mh_split_bf16x2(v0.z, v0.w, h1, l1);
                mh_split_bf16x2(v1.x, v1.y, h2, l2); mh_split_bf16x2(v1.z, v1.w, h3, l3);
                a[s] = (u32x4){h0, h1, h2, h3}; al[s] = (u32x4){l0, l1, l2, l3};
            } else {
                a[s] = (u32x4){mh_pack_bf16(v0.x, v0.y), mh_pack_bf16(v0.z, v0.w), mh_pack_bf16(v1.x, v1.y), mh_pack_bf16(v1.z, v1.w)};
            }
        }
    }
    __syncthreads();
    const float inv_c = 1.0f / (float)C;
    for (int blk = 0; blk < nb; ++blk) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        const unsigned short* Rb = Wh + (16 * wave + 16 * blk + li) * RS + lq * 8;
#pragma unroll
        for (int s = 0; s < CS32; ++s) {
            const u32x4 b = *reinterpret_cast<const u32x4*>(Rb + s * 32);
            if constexpr (X3) {
                const u32x4 bl = *reinterpret_cast<const u32x4*>(Rb + PLO + s * 32);
                acc = mh_mfma_bf16(al[s], b, acc);
                acc = mh_mfma_bf16(a[s], bl, acc);
            }
            acc = mh_mfma_bf16(a[s], b, acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int x = xb + 4 * lq + r;
            const int d = 16 * blk + li - (4 * lq + r);
            if (x < p.W && d >= 0 && d < p.D) p.out[(int64_t)(row * p.W + x) * p.out_ld + p.coff + d] = acc[r] * inv_c;
        }
    }
}

// Gradient of the large-shift cost volume on the matrix cores (same banded-GEMM view as corr_fwd_mfma):
//   RIGHT = false: dL[x][c]  = 1/C * sum_x' G[x][x'] * R[x'][c]      G[x][x'] = g[x][x' - x + md] inside the band, else 0
//   RIGHT = true : dR[x'][c] = 1/C * sum_x  G[x][x'] * L[x][c]
// One workgroup = 64 output pixels of a row, one wave = 16 of them; the other operand's window (64 + 2 md pixels, k-contiguous
// rows + 4 floats of padding) and the needed rows of g are staged in LDS; the A operand (the band of g) is gathered from
// the LDS copy of g with per-lane shift indices, the B operand is read column-wise from the window (conflict free:
// row stride = 4 banks mod 64); exact-fp32 v_mfma_f32_16x16x4_f32, ceil((2md+16)/16) * 4 * C/16 MFMAs per wave.
template <int CS16, bool RIGHT>
__global__ __launch_bounds__(CS16 >= 2 ? 512 : 256) void corr_bwd_mfma(CorrBwdArgs p, int segs) {
    HIP_DYNAMIC_SHARED(float, smem)
    constexpr int C = CS16 * 16, RS = C + 4;
    constexpr int NT = CS16 >= 2 ? 512 : 256;              // 8 waves: two per 16-pixel strip, each owning half the channel blocks
    constexpr int CBW = CS16 >= 2 ? CS16 / 2 : 1;          // channel blocks per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, half = tid >> 8;
    const int li = lane & 15, lq = lane >> 4;
    const int seg = blockIdx.x % segs;
    const int row = blockIdx.x / segs;
    const int x0 = seg * 64;
    const int nb = (2 * p.md + 16 + 15) / 16;
    const int rows = 48 + 16 * nb;                         // window pixels any wave touches
    const int DP = (p.D + 3) & ~3;
    float* const Ws = smem;                                // [rows][RS]   R (left gradient) or L (right gradient) window
    float* const Gs = smem + rows * RS;                    // [grows][DP]  g rows: the 64 outputs (left) / the window (right)
    const int grows = RIGHT ? rows : 64;
    const float* Wsrc = RIGHT ? p.L : p.R;
    const int w_ld = RIGHT ? p.l_ld : p.r_ld;
    // staging in batches of U independent loads per thread (4-8 waves per CU: a load-store loop would serialise on the latency)
    constexpr int U = 6;
    const int wtot = rows * (C / 4);
    for (int q0 = tid; q0 < wtot; q0 += NT * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = q0 + u * NT;
            const int xw = q / (C / 4), c4 = q - xw * (C / 4);
            const int xs = x0 - p.md + xw;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < wtot && xs >= 0 && xs < p.W) v[u] = *reinterpret_cast<const float4*>(Wsrc + ((int64_t)row * p.W + xs) * w_ld + c4 * 4);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = q0 + u * NT;
            if (q < wtot) *reinterpret_cast<float4*>(Ws + (q / (C / 4)) * RS + (q % (C / 4)) * 4) = v[u];
        }
    }
    const int gx0 = RIGHT ? x0 - p.md : x0;
    if (((p.g_ld | p.coff) & 3) == 0 && p.coff + DP <= p.g_ld) {
        // 16-byte rows: the pad columns (d >= D) hold whatever follows the volume in the row; the band gather below never reads them
        const int D4 = DP >> 2, gtot = grows * D4;
        const float inv = 1.0f / (float)D4;
        for (int q0 = tid; q0 < gtot; q0 += NT * U) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = q0 + u * NT;
                const int gr = (int)(((float)q + 0.5f) * inv), d4 = q - gr * D4;
                const int xs = gx0 + gr;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (q < gtot && xs >= 0 && xs < p.W) v[u] = *reinterpret_cast<const float4*>(p.g + ((int64_t)row * p.W + xs) * p.g_ld + p.coff + d4 * 4);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = q0 + u * NT;
                if (q < gtot) *reinterpret_cast<float4*>(Gs + q * 4) = v[u];
            }
        }
    } else {
        const int gtot = grows * DP;
        const float inv = 1.0f / (float)DP;
        for (int q0 = tid; q0 < gtot; q0 += NT * U) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = q0 + u * NT;
                const int gr = (int)(((float)q + 0.5f) * inv), d = q - gr * DP;
                const int xs = gx0 + gr;
                v[u] = 0.f;
                if (q < gtot && xs >= 0 && xs < p.W && d < p.D) v[u] = p.g[((int64_t)row * p.W + xs) * p.g_ld + p.coff + d];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = q0 + u * NT;
                if (q < gtot) Gs[q] = v[u];
            }
        }
    }
    __syncthreads();
    // A operand (band of g), 4 consecutive k per lane and k-step: k = ks*16 + lq*4 + t  <->  window pixel 16*wave + k
    f32x4 acc[CBW];
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < nb; ++ks) {
        float a[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int kw = ks * 16 + lq * 4 + t;
            // left: row = output pixel 16*wave + li, shift d = kw - li ; right: row = window pixel 16*wave + kw, d = li - kw + 2 md
            const int d = RIGHT ? li - kw + 2 * p.md : kw - li;
            const int gr = RIGHT ? 16 * wave + kw : 16 * wave + li;
            a[t] = (d >= 0 && d < p.D) ? Gs[gr * DP + d] : 0.f;
        }
        const float* Wb = Ws + (16 * wave + ks * 16 + lq * 4) * RS + half * (CBW * 16) + li;
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], Wb[t * RS + cb * 16], acc[cb], 0, 0, 0);
        }
    }
    const float inv_c = 1.0f / (float)C;
    float* const out = RIGHT ? p.dR : p.dL;
    const int o_ld = RIGHT ? p.dr_ld : p.dl_ld;
    const int accf = RIGHT ? p.acc_r : p.acc_l;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int x = x0 + 16 * wave + 4 * lq + r;
        if (x >= p.W) continue;
        float* o = out + ((int64_t)row * p.W + x) * o_ld + half * (CBW * 16) + li;
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb) {
            const float v = acc[cb][r] * inv_c;
            o[cb * 16] = accf ? o[cb * 16] + v : v;
        }
    }
}

// The same gradient on the bf16 matrix cores (precision code 1 of mh_corr_bwd_prec: the arithmetic of every other gradient of the 'mixed' / 'bf16' modes).
// The exact-fp32 pair above issues 6 x 4 x C/16 = 192 v_mfma_f32_16x16x4_f32 per 16 pixels and direction: at 35 FLOP/B the fp32 MFMA rate is its ceiling
// (27 % of the HBM peak at the protocol shape, profiles/r02_microbench_corr.txt).  Here the other feature's window sits in LDS as bf16 [pixel][C + 16]
// -- NHWC rows as they come from memory, converted at the store -- and the MFMA B operand (8 consecutive WINDOW PIXELS of one channel per lane) is
// fetched with the transposing LDS read (ds_read_b64_tr_b16: two reads per operand); the band of g is gathered from a bf16 LDS copy as the A operand and
// reused by all C/16 channel blocks: 3 x C/16 = 24 v_mfma_f32_16x16x32_bf16 per wave and direction at C = 128.  Results leave through an LDS transpose
// as whole 4 C-byte pixel rows (16-byte stores), not as 64-byte column fragments.  Workgroup ids are remapped so that the segments of an image row run on
// ONE XCD: the 2.25x window overlap between neighbouring segments is served by that XCD's L2 instead of being fetched once per XCD.
template <int CS16, bool RIGHT>
__device__ __forceinline__ void corr_bwd_mfma_bf16_body(const CorrBwdArgs& p, int segs, int bid, float* smem) {
    constexpr int C = CS16 * 16, RS = C + 16;              // halfs per window row (8 dwords of padding: the 4 rows of a transposing read hit distinct banks)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int seg = bid % segs;
    const int row = bid / segs;
    const int x0 = seg * 64;
    const int nks = (2 * p.md + 16 + 31) / 32;             // 32-pixel k-steps of a wave's band
    const int rows = 48 + 32 * nks;                        // window pixels any wave touches (rows past the band read as zeros: 0 x garbage must not happen)
    const int DP = (p.D + 7) & ~7;                         // halfs per g row
    unsigned short* const Ws = reinterpret_cast<unsigned short*>(smem);            // [rows][RS]
    const int grows = RIGHT ? rows : 64;
    unsigned short* const Gs = Ws + rows * RS;                                     // [grows][DP]
    const float* Wsrc = RIGHT ? p.L : p.R;
    const int w_ld = RIGHT ? p.l_ld : p.r_ld;
    const int npix = p.B * p.H * p.W;
    const __amdgpu_buffer_rsrc_t rsW = mh_make_rsrc(Wsrc, (unsigned)(((size_t)npix - 1) * w_ld * 4 + (size_t)C * 4));
    const __amdgpu_buffer_rsrc_t rsG = mh_make_rsrc(p.g, (unsigned)((size_t)npix * p.g_ld * 4));
    constexpr int U = 8;
    {   // the other operand's window, U independent 16-byte loads in flight per thread, rounded to bf16 at the LDS store
        const int items = rows * (C / 4);
        for (int q0 = tid; q0 < items; q0 += 256 * U) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = q0 + 256 * u;
                const int xw = q / (C / 4), c4 = q - xw * (C / 4);
                const int xs = x0 - p.md + xw;
                const bool ok = q < items && xs >= 0 && xs < p.W;
                v[u] = mh_buf_load4(rsW, ok ? ((row * p.W + xs) * w_ld + c4 * 4) * 4 : MH_OOB);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = q0 + 256 * u;
                if (q < items) {
                    const int xw = q / (C / 4), c4 = q - xw * (C / 4);
                    *reinterpret_cast<uint2*>(Ws + xw * RS + c4 * 4) = make_uint2(mh_pack_bf16(v[u].x, v[u].y), mh_pack_bf16(v[u].z, v[u].w));
                }
            }
        }
    }
    {   // g rows: the 64 outputs (left gradient) / the window (right gradient), rounded to bf16
        const int gx0 = RIGHT ? x0 - p.md : x0;
        const int D4 = (p.D + 3) >> 2;
        if (((p.g_ld | p.coff) & 3) == 0 && p.coff + 4 * D4 <= p.g_ld) {
            // 16-byte rows: the columns d >= D of the last group hold whatever follows the volume in the row; the band gather never reads them
            const int items = grows * D4;
            const float inv = 1.0f / (float)D4;
            for (int q0 = tid; q0 < items; q0 += 256 * U) {
                float4 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int q = q0 + 256 * u;
                    const int gr = (int)(((float)q + 0.5f) * inv), d4 = q - gr * D4;
                    const int xs = gx0 + gr;
                    const bool ok = q < items && xs >= 0 && xs < p.W;
                    v[u] = mh_buf_load4(rsG, ok ? ((row * p.W + xs) * p.g_ld + p.coff + d4 * 4) * 4 : MH_OOB);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int q = q0 + 256 * u;
                    if (q < items) {
                        const int gr = (int)(((float)q + 0.5f) * inv), d4 = q - gr * D4;
                        *reinterpret_cast<uint2*>(Gs + gr * DP + d4 * 4) = make_uint2(mh_pack_bf16(v[u].x, v[u].y), mh_pack_bf16(v[u].z, v[u].w));
                    }
                }
            }
        } else {
            const int DP2 = DP >> 1, items = grows * DP2;
            const float inv = 1.0f / (float)DP2;
            for (int q0 = tid; q0 < items; q0 += 256 * U) {
                float a[U], b[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int q = q0 + 256 * u;
                    const int gr = (int)(((float)q + 0.5f) * inv), d = 2 * (q - gr * DP2);
                    const int xs = gx0 + gr;
                    const bool ok = q < items && xs >= 0 && xs < p.W;
                    const int base = ((row * p.W + xs) * p.g_ld + p.coff + d) * 4;
                    a[u] = mh_buf_load1(rsG, (ok && d < p.D) ? base : MH_OOB);
                    b[u] = mh_buf_load1(rsG, (ok && d + 1 < p.D) ? base + 4 : MH_OOB);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int q = q0 + 256 * u;
                    if (q < items) reinterpret_cast<unsigned*>(Gs)[q] = mh_pack_bf16(a[u], b[u]);
                }
            }
        }
    }
    __syncthreads();
    f32x4 acc[CS16];
#pragma unroll
    for (int cb = 0; cb < CS16; ++cb) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < nks; ++ks) {
        // A operand (band of g): row m = li, k = 8 lq + t  <->  window pixel 16 wave + 32 ks + 8 lq + t
        unsigned short ah[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int kw = ks * 32 + lq * 8 + t;
            // left: output pixel 16 wave + li, shift d = kw - li ; right: g pixel = window pixel 16 wave + kw, d = li - kw + 2 md
            const int d = RIGHT ? li - kw + 2 * p.md : kw - li;
            const int gr = RIGHT ? 16 * wave + kw : 16 * wave + li;
            ah[t] = (d >= 0 && d < p.D) ? Gs[gr * DP + d] : (unsigned short)0;
        }
        const u32x4 a = (u32x4){(unsigned)ah[0] | ((unsigned)ah[1] << 16), (unsigned)ah[2] | ((unsigned)ah[3] << 16),
                                (unsigned)ah[4] | ((unsigned)ah[5] << 16), (unsigned)ah[6] | ((unsigned)ah[7] << 16)};
        // B operand: k = window pixel, n = channel: lane s of a 16-lane group addresses row (s >> 2), channels 4 (s & 3) .. + 3 of a [4][16] block
        const unsigned short* Wb = Ws + (16 * wave + 32 * ks + 8 * lq + (li >> 2)) * RS + 4 * (li & 3);
#pragma unroll
        for (int cb = 0; cb < CS16; ++cb) {
            const uint2 b0 = mh_lds_read_tr16(Wb + cb * 16), b1 = mh_lds_read_tr16(Wb + 4 * RS + cb * 16);
            acc[cb] = mh_mfma_bf16(a, (u32x4){b0.x, b0.y, b1.x, b1.y}, acc[cb]);
        }
    }
    __syncthreads();                                       // every wave is done with the window: its space takes the result tile
    // acc[cb][r]: pixel 16 wave + 4 lq + r, channel 16 cb + li  ->  LDS [64 pixels][C + 4] fp32  ->  whole pixel rows
    float* const Ts = smem;
    constexpr int TS = C + 4;
    const float inv_c = 1.0f / (float)C;
#pragma unroll
    for (int cb = 0; cb < CS16; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) Ts[(16 * wave + 4 * lq + r) * TS + cb * 16 + li] = acc[cb][r] * inv_c;
    __syncthreads();
    float* const out = RIGHT ? p.dR : p.dL;
    const int o_ld = RIGHT ? p.dr_ld : p.dl_ld;
    const int accf = RIGHT ? p.acc_r : p.acc_l;
    for (int q = tid; q < 64 * (C / 4); q += 256) {
        const int px = q / (C / 4), c4 = q - px * (C / 4);
        const int x = x0 + px;
        if (x >= p.W) continue;
        float4 v = *reinterpret_cast<const float4*>(Ts + px * TS + c4 * 4);
        float4* o = reinterpret_cast<float4*>(out + ((int64_t)row * p.W + x) * o_ld + c4 * 4);
        if (accf) { const float4 t = *o; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
        *o = v;
    }
}

// BOTH directions in ONE grid: workgroup 2 s computes the left gradient of segment s, workgroup 2 s + 1 the right gradient of the same segment.  With the
// XCD-aware order the pair runs on one XCD at about the same time, so the g rows the two share (the left one's 64 are a subset of the right one's window)
// come from HBM once and from that L2 the second time (two launches read g twice: 14 % of the gradient's traffic), and the step has one launch less.
template <int CS16>
__global__ __launch_bounds__(256) void corr_bwd_mfma_bf16_pair(CorrBwdArgs p, int segs, int remap) {
    HIP_DYNAMIC_SHARED(float, smem)
    const int lin = remap ? mh_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    if (lin & 1) corr_bwd_mfma_bf16_body<CS16, true>(p, segs, lin >> 1, smem);
    else corr_bwd_mfma_bf16_body<CS16, false>(p, segs, lin >> 1, smem);
}

// (one direction per launch: the A/B partner behind mh_tune_corr bit 2)
template <int CS16, bool RIGHT>
__global__ __launch_bounds__(256) void corr_bwd_mfma_bf16_one(CorrBwdArgs p, int segs, int remap) {
    HIP_DYNAMIC_SHARED(float, smem)
    corr_bwd_mfma_bf16_body<CS16, RIGHT>(p, segs, remap ? mh_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x, smem);
}

}  // namespace

int mh_corr_init() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&corr_fwd_large<32>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) { mh_set_error("corr: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
#define MH_CORR_ATTR(CSv)                                                                                                     \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&corr_fwd_mfma<CSv>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); \
    if (e != hipSuccess) { mh_set_error("corr: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    MH_CORR_ATTR(1) MH_CORR_ATTR(2) MH_CORR_ATTR(4) MH_CORR_ATTR(8) MH_CORR_ATTR(16)
#undef MH_CORR_ATTR
#define MH_CORRH_ATTR(CSv, Xv)                                                                                                 \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&corr_fwd_mfma_bf16<CSv, Xv>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); \
    if (e != hipSuccess) { mh_set_error("corr: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    MH_CORRH_ATTR(1, false) MH_CORRH_ATTR(2, false) MH_CORRH_ATTR(4, false) MH_CORRH_ATTR(8, false)
    MH_CORRH_ATTR(1, true) MH_CORRH_ATTR(2, true) MH_CORRH_ATTR(4, true) MH_CORRH_ATTR(8, true)
#undef MH_CORRH_ATTR
#define MH_CORRB_ATTR(CSv, Rv)                                                                                                 \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&corr_bwd_mfma<CSv, Rv>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); \
    if (e != hipSuccess) { mh_set_error("corr: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    MH_CORRB_ATTR(1, false) MH_CORRB_ATTR(2, false) MH_CORRB_ATTR(4, false) MH_CORRB_ATTR(8, false) MH_CORRB_ATTR(16, false)
    MH_CORRB_ATTR(1, true) MH_CORRB_ATTR(2, true) MH_CORRB_ATTR(4, true) MH_CORRB_ATTR(8, true) MH_CORRB_ATTR(16, true)
#undef MH_CORRB_ATTR
#define MH_CORRBH_ATTR(CSv)                                                                                                    \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&corr_bwd_mfma_bf16_pair<CSv>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); \
    if (e != hipSuccess) { mh_set_error("corr: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }                 \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&corr_bwd_mfma_bf16_one<CSv, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); \
    if (e != hipSuccess) { mh_set_error("corr: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }                 \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&corr_bwd_mfma_bf16_one<CSv, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); \
    if (e != hipSuccess) { mh_set_error("corr: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    MH_CORRBH_ATTR(2) MH_CORRBH_ATTR(4) MH_CORRBH_ATTR(8) MH_CORRBH_ATTR(16)
#undef MH_CORRBH_ATTR
#define MH_CWBL_ATTR(LPPv)                                                                                                    \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&corr_warp_bwd_rowlds_kernel<LPPv, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024); \
    if (e != hipSuccess) { mh_set_error("corr: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    MH_CWBL_ATTR(8) MH_CWBL_ATTR(16) MH_CWBL_ATTR(32)
#undef MH_CWBL_ATTR
#define MH_CWBR_ATTR(LPPv, Dv)                                                                                                 \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&corr_warp_bwd_row_kernel<LPPv, 5, Dv>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); \
    if (e != hipSuccess) { mh_set_error("corr: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    MH_CWBR_ATTR(4, false) MH_CWBR_ATTR(8, false) MH_CWBR_ATTR(16, false) MH_CWBR_ATTR(4, true) MH_CWBR_ATTR(8, true) MH_CWBR_ATTR(16, true)
#undef MH_CWBR_ATTR
    return 0;
}

static std::atomic<int> g_corr_direct{1};
// mh_tune_corr_row: 1 (default) = row-owned backward front end with the row's operands staged in LDS, 3 = row-owned, operands from L1 / L2, 0 = the global-atomic form
static std::atomic<int> g_corr_row{1};
static std::atomic<int> g_front_rows2{1};            // level front end with the head inside: two fine rows per workgroup (mh_tune_corr bit 3 clears it: one row)
static std::atomic<int> g_corr_pair{1};              // large-D bf16 gradient: both directions in one grid (mh_tune_corr bit 2 clears it: two launches)
static std::atomic<int> g_corr_remap{1};             // XCD-aware workgroup order of the large-D bf16 kernels (mh_tune_corr bit 1 clears it)
static std::atomic<int> g_corr_det_ranges{0};        // how many deterministic ranges are registered (mh_det_sync_corr keeps it in step)
extern "C" int mh_tune_corr_row(int on) { g_corr_row = on; return 0; }
// tuning hook: 0 = LDS-staged window kernel, 1 = direct (no LDS) kernel for D <= 9
extern "C" int mh_tune_corr(int direct) { g_corr_direct = direct & 1; g_corr_remap = (direct & 2) ? 0 : 1; g_corr_pair = (direct & 4) ? 0 : 1; g_front_rows2 = (direct & 8) ? 0 : 1; return 0; }

extern "C" int mh_corr_fwd(const float* L, int32_t l_ld, const float* R, int32_t r_ld, const float* u,
                           float* out, int32_t out_ld, int32_t coff,
                           int32_t B, int32_t H, int32_t W, int32_t C, int32_t max_disp, int32_t stride,
                           int32_t copy_left, int32_t zero_tail, void* stream) {
    return mh_corr_fwd_prec(L, l_ld, R, r_ld, u, out, out_ld, coff, B, H, W, C, max_disp, stride, copy_left, zero_tail, 0, stream);
}

extern "C" int mh_corr_fwd_prec(const float* L, int32_t l_ld, const float* R, int32_t r_ld, const float* u,
                                float* out, int32_t out_ld, int32_t coff,
                                int32_t B, int32_t H, int32_t W, int32_t C, int32_t max_disp, int32_t stride,
                                int32_t copy_left, int32_t zero_tail, int32_t precision, void* stream) {
    MH_REQUIRE(precision >= 0 && precision <= 2, MH_ERR_ARG, "mh_corr_fwd_prec: precision must be 0 (fp32), 1 (bf16) or 2 (split-bf16)");
    MH_REQUIRE(L && R && out, MH_ERR_ARG, "mh_corr_fwd: null argument");
    MH_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && max_disp >= 0 && stride >= 1, MH_ERR_ARG, "mh_corr_fwd: bad dimension");
    MH_REQUIRE(C % 4 == 0 && l_ld % 4 == 0 && r_ld % 4 == 0 && mh_aligned16(L) && mh_aligned16(R), MH_ERR_ALIGN,
               "mh_corr_fwd: C, l_ld, r_ld must be multiples of 4 and L/R 16-byte aligned");
    const int D = 2 * max_disp / stride + 1;
    MH_REQUIRE(coff + D + (u ? 1 : 0) <= out_ld, MH_ERR_ARG, "mh_corr_fwd: out_ld too small");
    MH_REQUIRE(!copy_left || (out_ld % 4 == 0 && mh_aligned16(out) && coff >= C), MH_ERR_ALIGN,
               "mh_corr_fwd: copy_left needs 16-byte aligned rows and coff >= C");
    CorrArgs a;
    a.L = L; a.R = R; a.u = u; a.out = out; a.l_ld = l_ld; a.r_ld = r_ld; a.out_ld = out_ld; a.coff = coff;
    a.B = B; a.H = H; a.W = W; a.C = C; a.md = max_disp; a.stride = stride; a.D = D;
    a.copy_left = copy_left; a.zero_tail = zero_tail; a.remap = g_corr_remap.load();
    hipStream_t s = (hipStream_t)stream;
    const int C4 = C / 4;
    const int64_t lb = (((int64_t)B * H * W - 1) * l_ld + C) * 4, rb = (((int64_t)B * H * W - 1) * r_ld + C) * 4;
    if (D <= MAXD_SMALL && g_corr_direct && lb < (1ll << 31) - 64 && rb < (1ll << 31) - 64) {
        a.l_bytes = (unsigned)lb; a.r_bytes = (unsigned)rb;
        const int64_t npix = (int64_t)B * H * W;
        MH_REQUIRE(npix < (1ll << 31) - 256, MH_ERR_UNSUPPORTED, "mh_corr_fwd: too many pixels");
        auto grid = [&](int lpp) { return dim3((unsigned)((npix * lpp + 255) / 256)); };
#define MH_CORR(LPPv)                                                                                           \
        { if (D <= 5) hipLaunchKernelGGL((corr_fwd_direct<LPPv, 5>), grid(LPPv), dim3(256), 0, s, a);            \
          else hipLaunchKernelGGL((corr_fwd_direct<LPPv, MAXD_SMALL>), grid(LPPv), dim3(256), 0, s, a); }
        if (C4 <= 4) MH_CORR(4) else if (C4 <= 8) MH_CORR(8) else MH_CORR(16)
#undef MH_CORR
        mh_note_kernel("corr_fwd_direct<LPP=%d,DT=%d>", C4 <= 4 ? 4 : (C4 <= 8 ? 8 : 16), D <= 5 ? 5 : MAXD_SMALL);
        return mh_check_launch("corr_fwd_direct");
    }
    if (D <= MAXD_SMALL) {
        // lanes per pixel: cover the channel groups with at most 16 lanes
        const int TW = 64;
        a.segs = mh_cdiv(W, TW);
        const size_t lds = (size_t)(TW + 2 * max_disp) * C * sizeof(float);
        MH_REQUIRE(lds <= 64 * 1024, MH_ERR_UNSUPPORTED, "mh_corr_fwd: window does not fit LDS (C=%d, md=%d)", C, max_disp);
        const dim3 grid(a.segs * B * H);
        if (C4 <= 4) hipLaunchKernelGGL((corr_fwd_small<4, 64>), grid, dim3(256), lds, s, a);
        else if (C4 <= 8) hipLaunchKernelGGL((corr_fwd_small<8, 64>), grid, dim3(256), lds, s, a);
        else hipLaunchKernelGGL((corr_fwd_small<16, 64>), grid, dim3(256), lds, s, a);
    } else if (g_corr_direct && stride == 1 && !copy_left && !u && !zero_tail && (C == 16 || C == 32 || C == 64 || C == 128 || C == 256) &&
               lb < (1ll << 31) - 64 && rb < (1ll << 31) - 64 &&
               (size_t)(48 + 16 * ((2 * max_disp + 31) / 16)) * (C + 4) * sizeof(float) <= 150 * 1024) {
        a.l_bytes = (unsigned)lb; a.r_bytes = (unsigned)rb;
        a.segs = mh_cdiv(W, 64);
        const dim3 grid(a.segs * B * H);
        if (precision != 0 && C >= 32 && C <= 256) {
            // bf16 / split-bf16 operands on v_mfma_f32_16x16x32_bf16 (the large-D volume only; D <= 9 is pure bandwidth and stays fp32)
            const size_t ldsh = (size_t)(48 + 16 * ((2 * max_disp + 31) / 16)) * (C + 16) * 2 * (precision == 2 ? 2 : 1);
            if (ldsh <= 150 * 1024) {
#define MH_CH(CSv) { if (precision == 2) hipLaunchKernelGGL((corr_fwd_mfma_bf16<CSv, true>), grid, dim3(256), ldsh, s, a);         \
                     else hipLaunchKernelGGL((corr_fwd_mfma_bf16<CSv, false>), grid, dim3(256), ldsh, s, a); }
                switch (C) { case 32: MH_CH(1) break; case 64: MH_CH(2) break; case 128: MH_CH(4) break; default: MH_CH(8) break; }
#undef MH_CH
                mh_note_kernel("corr_fwd_mfma_bf16<C/32=%d,%s>", C / 32, precision == 2 ? "bf16x3" : "bf16");
                return mh_check_launch("corr_fwd_mfma_bf16");
            }
        }
        const size_t lds = (size_t)(48 + 16 * ((2 * max_disp + 31) / 16)) * (C + 4) * sizeof(float);
        switch (C) {
            case 16: hipLaunchKernelGGL((corr_fwd_mfma<1>), grid, dim3(256), lds, s, a); break;
            case 32: hipLaunchKernelGGL((corr_fwd_mfma<2>), grid, dim3(256), lds, s, a); break;
            case 64: hipLaunchKernelGGL((corr_fwd_mfma<4>), grid, dim3(256), lds, s, a); break;
            case 128: hipLaunchKernelGGL((corr_fwd_mfma<8>), grid, dim3(256), lds, s, a); break;
            default: hipLaunchKernelGGL((corr_fwd_mfma<16>), grid, dim3(256), lds, s, a); break;
        }
        mh_note_kernel("corr_fwd_mfma<C/16=%d>", C / 16);
        return mh_check_launch("corr_fwd_mfma");
    } else {
        const int TW = 32;
        a.segs = mh_cdiv(W, TW);
        const size_t lds = (size_t)(2 * TW + 2 * max_disp) * (C + 4) * sizeof(float);
        MH_REQUIRE(lds <= 150 * 1024, MH_ERR_UNSUPPORTED, "mh_corr_fwd: tiles do not fit LDS (C=%d, md=%d)", C, max_disp);
        // (the > 64 KiB dynamic-LDS opt-in of this kernel is done once in mh_init(), never during a capture)
        hipLaunchKernelGGL((corr_fwd_large<32>), dim3(a.segs * B * H), dim3(256), lds, s, a);
    }
    return mh_check_launch("corr_fwd");
}

extern "C" int mh_level_front_fwd(const float* Vc, int32_t Hc, int32_t Wc, float mul, const float* L, int32_t l_ld, const float* R,
                                  int32_t r_ld, float* out, int32_t out_ld, int32_t coff, float* Rw, int32_t rw_ld, float* u,
                                  int32_t B, int32_t H, int32_t W, int32_t C, int32_t max_disp, int32_t zero_tail, void* stream) {
    return mh_level_front_fwd_planes(Vc, Hc, Wc, mul, L, l_ld, R, r_ld, out, out_ld, coff, Rw, rw_ld, u, B, H, W, C, max_disp, zero_tail, nullptr, nullptr, 0, stream);
}
// columns of the coarse map a workgroup of `ppb` pixels (+ max_disp shifts either side) can touch: the HEAD instances' LDS row length
static int front_head_cwcap(int Wc, int W, int ppb, int max_disp) {
    const int span = ppb - 1 + 2 * max_disp;                          // fine columns between the first and the last tap's pixel
    const int64_t c = ((int64_t)span * Wc + W - 1) / W + 3;            // floor(hi * sx) - floor(lo * sx) <= ceil(span * sx); + the x1 column, + rounding slack
    return (int)(c < Wc ? c : Wc);
}
extern "C" int mh_level_front_head_ok(int32_t Hc, int32_t Wc, int32_t H, int32_t W, int32_t C, int32_t K, int32_t max_disp) {
    if (Hc <= 0 || Wc <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || max_disp < 0) return 0;
    if (K % 4 != 0 || K > 32 || 2 * max_disp + 1 > MAXD_SMALL) return 0;           // 8 lanes per head value, one channel group each
    if (H < Hc || W < Wc) return 0;            // every coarse pixel is stored by the workgroup whose fine pixels interpolate from it: needs an up-scaling geometry
    const int C4 = C / 4, lpp = C4 <= 4 ? 4 : (C4 <= 8 ? 8 : 16);
    const int cw = front_head_cwcap(Wc, W, 256 / lpp, max_disp);
    if (4 * (cw + 2) * (K / 4) > 6 * 256) return 0;                                 // the patch staging is six straight-line loads per thread
    return ((size_t)4 * (cw + 2) * K + 9 * K + 2 * cw) * sizeof(float) <= 48 * 1024 ? 1 : 0;
}

static int level_front_launch(const float* X, int32_t x_ld, int32_t K, const float* hw, const float* hb,
                              const float* Vc, int32_t Hc, int32_t Wc, float mul, const float* L, int32_t l_ld, const float* R,
                              int32_t r_ld, float* out, int32_t out_ld, int32_t coff, float* Rw, int32_t rw_ld, float* u,
                              int32_t B, int32_t H, int32_t W, int32_t C, int32_t max_disp, int32_t zero_tail,
                              void* out_hi, void* out_lo, int32_t out_pld, void* stream) {
    MH_REQUIRE(!out_lo || out_hi, MH_ERR_ARG, "mh_level_front_fwd_planes: the lo plane needs the hi plane");
    MH_REQUIRE(!out_hi || (out_pld >= coff + 2 * max_disp + 2 && (out_pld & 3) == 0 && (((uintptr_t)out_hi) & 7u) == 0 && (((uintptr_t)out_lo) & 7u) == 0), MH_ERR_ARG,
               "mh_level_front_fwd_planes: out_pld must cover coff + D + 1 (multiple of 4), planes 8-byte aligned");
    MH_REQUIRE(Vc && L && R && out && Rw && u, MH_ERR_ARG, "mh_level_front_fwd: null argument");
    MH_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && Hc > 0 && Wc > 0 && max_disp >= 0, MH_ERR_ARG, "mh_level_front_fwd: bad dimension");
    const int D = 2 * max_disp + 1;
    MH_REQUIRE(D <= MAXD_SMALL, MH_ERR_UNSUPPORTED, "mh_level_front_fwd: at most %d shifts (use mh_resize_fwd + mh_warp_fwd + mh_corr_fwd)", MAXD_SMALL);
    MH_REQUIRE(C % 4 == 0 && l_ld % 4 == 0 && r_ld % 4 == 0 && rw_ld % 4 == 0 && out_ld % 4 == 0 && mh_aligned16(L) && mh_aligned16(R) &&
               mh_aligned16(Rw) && mh_aligned16(out), MH_ERR_ALIGN, "mh_level_front_fwd: 16-byte rows required");
    MH_REQUIRE(coff >= C && coff + D + 1 <= out_ld && rw_ld >= C, MH_ERR_ARG, "mh_level_front_fwd: out_ld / coff / rw_ld too small");
    const int64_t npix = (int64_t)B * H * W;
    const int64_t lb = ((npix - 1) * l_ld + C) * 4, rb = ((npix - 1) * r_ld + C) * 4;
    MH_REQUIRE(npix < (1ll << 31) - 256 && lb < (1ll << 31) - 64 && rb < (1ll << 31) - 64, MH_ERR_UNSUPPORTED, "mh_level_front_fwd: tensors must be < 2 GiB");
    FrontArgs a;
    a.Vc = Vc; a.L = L; a.R = R; a.out = out; a.Rw = Rw; a.u = u; a.Hc = Hc; a.Wc = Wc; a.mul = mul;
    a.sy = (float)Hc / (float)H; a.sx = (float)Wc / (float)W;
    a.l_ld = l_ld; a.r_ld = r_ld; a.out_ld = out_ld; a.rw_ld = rw_ld; a.coff = coff;
    a.B = B; a.H = H; a.W = W; a.C = C; a.md = max_disp; a.D = D; a.zero_tail = zero_tail;
    a.l_bytes = (unsigned)lb; a.r_bytes = (unsigned)rb;
    a.out_hi = (unsigned short*)out_hi; a.out_lo = (unsigned short*)out_lo; a.out_pld = out_pld;
    a.X = nullptr; a.hw = nullptr; a.hb = nullptr; a.Vw = nullptr; a.x_ld = 0; a.K = 0; a.segs = 0; a.cwcap = 0; a.x_bytes = 0; a.rows2 = 0; a.rowpairs = 0;
    hipStream_t s = (hipStream_t)stream;
    const int C4 = C / 4;
    const int lpp = C4 <= 4 ? 4 : (C4 <= 8 ? 8 : 16);
    if (X) {
        // the coarser level's disparity head inside this launch (mh_level_front_head_fwd)
        MH_REQUIRE(mh_level_front_head_ok(Hc, Wc, H, W, C, K, max_disp) == 1, MH_ERR_UNSUPPORTED,
                   "mh_level_front_head_fwd: K = %d channels / %dx%d -> %dx%d not served (ask mh_level_front_head_ok; run the head as a launch of its own)", K, Hc, Wc, H, W);
        MH_REQUIRE(hw && x_ld >= K && x_ld % 4 == 0 && mh_aligned16(X) && mh_aligned16(hw), MH_ERR_ALIGN, "mh_level_front_head_fwd: X rows / head weights must be 16-byte aligned");
        const int64_t xb = (((int64_t)B * Hc * Wc - 1) * x_ld + K) * 4;
        MH_REQUIRE(xb < (1ll << 31) - 64, MH_ERR_UNSUPPORTED, "mh_level_front_head_fwd: X must be < 2 GiB");
        a.X = X; a.hw = hw; a.hb = hb; a.Vw = const_cast<float*>(Vc); a.x_ld = x_ld; a.K = K; a.x_bytes = (unsigned)xb;
        const int ppb = 256 / lpp;
        a.rows2 = (H == 2 * Hc && g_front_rows2.load(std::memory_order_relaxed)) ? 1 : 0;
        const int wpx = a.rows2 ? ppb / 2 : ppb;
        a.rowpairs = H / 2;
        a.segs = mh_cdiv(W, wpx);
        a.cwcap = front_head_cwcap(Wc, W, wpx, max_disp);
        const size_t lds = ((size_t)4 * (a.cwcap + 2) * K + 9 * K + 2 * a.cwcap) * sizeof(float);
        const dim3 g((unsigned)((int64_t)a.segs * B * (a.rows2 ? H / 2 : H)));
#define MH_FRONTH(LPPv)                                                                                                      \
    { if (D <= 5) hipLaunchKernelGGL((level_front_kernel<LPPv, 5, true>), g, dim3(256), lds, s, a);                          \
      else hipLaunchKernelGGL((level_front_kernel<LPPv, MAXD_SMALL, true>), g, dim3(256), lds, s, a); }
        if (lpp == 4) MH_FRONTH(4) else if (lpp == 8) MH_FRONTH(8) else MH_FRONTH(16)
#undef MH_FRONTH
        mh_note_kernel("level_front_kernel<LPP=%d,DT=%d,HEAD=%d> grid %d lds %d", lpp, D <= 5 ? 5 : MAXD_SMALL, K, (int)g.x, (int)lds);
        return mh_check_launch("level_front_head_fwd");
    }
    auto grid = [&](int l) { return dim3((unsigned)((npix * l + 255) / 256)); };
#define MH_FRONT(LPPv)                                                                                              \
    { if (D <= 5) hipLaunchKernelGGL((level_front_kernel<LPPv, 5, false>), grid(LPPv), dim3(256), 0, s, a);                \
      else hipLaunchKernelGGL((level_front_kernel<LPPv, MAXD_SMALL, false>), grid(LPPv), dim3(256), 0, s, a); }
    if (lpp == 4) MH_FRONT(4) else if (lpp == 8) MH_FRONT(8) else MH_FRONT(16)
#undef MH_FRONT
    mh_note_kernel("level_front_kernel<LPP=%d,DT=%d>", lpp, D <= 5 ? 5 : MAXD_SMALL);
    return mh_check_launch("level_front_fwd");
}

extern "C" int mh_level_front_fwd_planes(const float* Vc, int32_t Hc, int32_t Wc, float mul, const float* L, int32_t l_ld, const float* R,
                                         int32_t r_ld, float* out, int32_t out_ld, int32_t coff, float* Rw, int32_t rw_ld, float* u,
                                         int32_t B, int32_t H, int32_t W, int32_t C, int32_t max_disp, int32_t zero_tail,
                                         void* out_hi, void* out_lo, int32_t out_pld, void* stream) {
    return level_front_launch(nullptr, 0, 0, nullptr, nullptr, Vc, Hc, Wc, mul, L, l_ld, R, r_ld, out, out_ld, coff, Rw, rw_ld, u, B, H, W, C, max_disp, zero_tail, out_hi, out_lo,
                              out_pld, stream);
}
extern "C" int mh_level_front_head_fwd(const float* X, int32_t x_ld, int32_t K, const float* hw, const float* hb, float* Vc, int32_t Hc, int32_t Wc, float mul,
                                       const float* L, int32_t l_ld, const float* R, int32_t r_ld, float* out, int32_t out_ld, int32_t coff, float* Rw, int32_t rw_ld,
                                       float* u, int32_t B, int32_t H, int32_t W, int32_t C, int32_t max_disp, int32_t zero_tail,
                                       void* out_hi, void* out_lo, int32_t out_pld, void* stream) {
    MH_REQUIRE(X, MH_ERR_ARG, "mh_level_front_head_fwd: null argument");
    return level_front_launch(X, x_ld, K, hw, hb, Vc, Hc, Wc, mul, L, l_ld, R, r_ld, out, out_ld, coff, Rw, rw_ld, u, B, H, W, C, max_disp, zero_tail, out_hi, out_lo, out_pld,
                              stream);
}

extern "C" int mh_corr_warp_bwd(const float* g, int32_t g_ld, int32_t coff, const float* L, int32_t l_ld, const float* Rw, int32_t rw_ld,
                                const float* img, int32_t img_ld, const float* u, float* dL, int32_t dl_ld, int32_t acc_l,
                                float* dimg, int32_t dimg_ld, float* du,
                                int32_t B, int32_t H, int32_t W, int32_t C, int32_t max_disp, int32_t stride, int32_t copy_left, void* stream) {
    MH_REQUIRE(g && L && Rw && img && u && dL && (dimg || du), MH_ERR_ARG, "mh_corr_warp_bwd: null argument");
    MH_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && max_disp >= 0 && stride >= 1, MH_ERR_ARG, "mh_corr_warp_bwd: bad dimension");
    MH_REQUIRE(C % 4 == 0 && l_ld % 4 == 0 && rw_ld % 4 == 0 && img_ld % 4 == 0 && dl_ld % 4 == 0 && (!dimg || dimg_ld % 4 == 0) &&
               mh_aligned16(L) && mh_aligned16(Rw) && mh_aligned16(img) && mh_aligned16(dL), MH_ERR_ALIGN,
               "mh_corr_warp_bwd: channel counts / lds must be multiples of 4 and pointers 16-byte aligned");
    MH_REQUIRE(!copy_left || (g_ld % 4 == 0 && mh_aligned16(g)), MH_ERR_ALIGN, "mh_corr_warp_bwd: copy_left needs aligned g rows");
    CorrWarpBwdArgs a;
    a.g = g; a.L = L; a.Rw = Rw; a.img = img; a.u = u; a.dL = dL; a.dimg = dimg; a.du = du;
    a.g_ld = g_ld; a.coff = coff; a.l_ld = l_ld; a.rw_ld = rw_ld; a.img_ld = img_ld; a.dl_ld = dl_ld; a.dimg_ld = dimg_ld;
    a.acc_l = acc_l & 1;
    a.acc_img = (acc_l & MH_CORR_WARP_OVERWRITE_DIMG) ? 0 : 1;
    a.B = B; a.H = H; a.W = W; a.C = C; a.md = max_disp; a.stride = stride; a.D = 2 * max_disp / stride + 1; a.copy_left = copy_left;
    const int C4 = C / 4;
    const int64_t npix = (int64_t)B * H * W;
    hipStream_t s = (hipStream_t)stream;
    // the forms that SCATTER into dimg (LDS copy of the row / global atomics) start from what dimg holds: an overwriting call zeroes it for them
    auto zero_dimg = [&]() -> int {
        if (a.acc_img || !dimg) return 0;
        hipError_t e = (dimg_ld == C) ? hipMemsetAsync(dimg, 0, (size_t)npix * C * 4, s) : hipMemset2DAsync(dimg, (size_t)dimg_ld * 4, 0, (size_t)C * 4, (size_t)npix, s);
        if (e != hipSuccess) { mh_set_error("mh_corr_warp_bwd: zeroing dimg: %s", hipGetErrorString(e)); return (int)e; }
        a.acc_img = 1;
        return 0;
    };
    auto grid = [&](int lpp) { int64_t b = (npix * lpp + 255) / 256; return (int)(b > (1 << 20) ? (1 << 20) : b); };
    const int64_t ldmax = g_ld > rw_ld ? (g_ld > l_ld ? g_ld : l_ld) : (rw_ld > l_ld ? rw_ld : l_ld);
    const bool fast = a.D <= 5 && npix * ldmax * 4 < (1ll << 31) - 64;        // MADNet's radius-2 volumes: the branch-free form
    // row-owned form: LDS copy of the row's right-tower gradient (W x C accumulators) + the u row; every operand under 2 GiB (buffer descriptors)
    const bool det = g_corr_det_ranges.load() > 0;
    const size_t row_lds = ((size_t)W * C * (det ? 8 : 4) + (size_t)W * 4 + 15) & ~(size_t)15;
    const int64_t ldmax2 = std::max<int64_t>(std::max<int64_t>(ldmax, img_ld), std::max<int64_t>(dl_ld, dimg ? dimg_ld : 0));
    if (g_corr_row.load() && fast && row_lds <= 150 * 1024 && npix * ldmax2 * 4 < (1ll << 31) - 64 && (int64_t)B * H < (1 << 30)) {
        const dim3 grid((unsigned)(B * H));
        // operands staged in LDS: 3 rows of W x C floats + the g / u / tap rows; a thread owns <= 3 (pixel, channel group) items.  Gather, no atomics:
        // deterministic as it is (mh_tune_corr_row bit 1 set = the scatter form below; bit 2 = this launch without its scatter / gather part: timing)
        const int lpp = C4 <= 8 ? 8 : C4 <= 16 ? 16 : 32;
        const size_t lds_st = ((size_t)3 * W * C + (size_t)W * (8 + 1 + 2 + 1 + 1 + 2 + 8) + 8) * 4;
        if ((g_corr_row.load() & 2) == 0 && C4 <= 32 && W <= 3 * (1024 / lpp) && (int64_t)W * C4 <= 3 * 1024 && W * 8 <= 3 * 1024 && W < 32768 && lds_st <= 155 * 1024) {
            if (g_corr_row.load() & 4) a.dimg = nullptr;
            if (lpp == 8) hipLaunchKernelGGL((corr_warp_bwd_rowlds_kernel<8, 5>), grid, dim3(1024), lds_st, s, a);
            else if (lpp == 16) hipLaunchKernelGGL((corr_warp_bwd_rowlds_kernel<16, 5>), grid, dim3(1024), lds_st, s, a);
            else hipLaunchKernelGGL((corr_warp_bwd_rowlds_kernel<32, 5>), grid, dim3(1024), lds_st, s, a);
            mh_note_kernel("corr_warp_bwd_rowlds_kernel<LPP=%d,DT=5> C=%d D=%d grid %d x 16 waves lds %d", lpp, C, a.D, B * H, (int)lds_st);
            return mh_check_launch("corr_warp_bwd_rowlds");
        }
        if (int rc = zero_dimg()) return rc;
#define MH_CWB_ROW(LPPv) { if (det) hipLaunchKernelGGL((corr_warp_bwd_row_kernel<LPPv, 5, true>), grid, dim3(1024), row_lds, s, a);   \
                           else hipLaunchKernelGGL((corr_warp_bwd_row_kernel<LPPv, 5, false>), grid, dim3(1024), row_lds, s, a); }
        if (C4 <= 4) MH_CWB_ROW(4) else if (C4 <= 8) MH_CWB_ROW(8) else MH_CWB_ROW(16)
#undef MH_CWB_ROW
        mh_note_kernel("corr_warp_bwd_row_kernel<LPP=%d,DT=5%s> C=%d D=%d grid %d x 16 waves lds %d", C4 <= 4 ? 4 : C4 <= 8 ? 8 : 16, det ? ",det" : "", C, a.D, B * H, (int)row_lds);
        return mh_check_launch("corr_warp_bwd_row");
    }
    if (int rc = zero_dimg()) return rc;
    if (fast) {
        if (C4 <= 4) hipLaunchKernelGGL((corr_warp_bwd_kernel<4, 5>), dim3(grid(4)), dim3(256), 0, s, a);
        else if (C4 <= 8) hipLaunchKernelGGL((corr_warp_bwd_kernel<8, 5>), dim3(grid(8)), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((corr_warp_bwd_kernel<16, 5>), dim3(grid(16)), dim3(256), 0, s, a);
    } else if (C4 <= 4) hipLaunchKernelGGL((corr_warp_bwd_kernel<4>), dim3(grid(4)), dim3(256), 0, s, a);
    else if (C4 <= 8) hipLaunchKernelGGL((corr_warp_bwd_kernel<8>), dim3(grid(8)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((corr_warp_bwd_kernel<16>), dim3(grid(16)), dim3(256), 0, s, a);
    mh_note_kernel("corr_warp_bwd_kernel<LPP=%d%s> C=%d D=%d", C4 <= 4 ? 4 : C4 <= 8 ? 8 : 16, fast ? ",DT=5" : "", C, a.D);
    return mh_check_launch("corr_warp_bwd");
}

extern "C" int mh_corr_bwd(const float* g, int32_t g_ld, int32_t coff, const float* L, int32_t l_ld,
                           const float* R, int32_t r_ld, float* dL, int32_t dl_ld, int32_t acc_l,
                           float* dR, int32_t dr_ld, int32_t acc_r, float* du, int32_t acc_u,
                           int32_t B, int32_t H, int32_t W, int32_t C, int32_t max_disp, int32_t stride,
                           int32_t copy_left, void* stream) {
    return mh_corr_bwd_prec(g, g_ld, coff, L, l_ld, R, r_ld, dL, dl_ld, acc_l, dR, dr_ld, acc_r, du, acc_u, B, H, W, C, max_disp, stride, copy_left, 0, stream);
}

extern "C" int mh_corr_bwd_prec(const float* g, int32_t g_ld, int32_t coff, const float* L, int32_t l_ld,
                                const float* R, int32_t r_ld, float* dL, int32_t dl_ld, int32_t acc_l,
                                float* dR, int32_t dr_ld, int32_t acc_r, float* du, int32_t acc_u,
                                int32_t B, int32_t H, int32_t W, int32_t C, int32_t max_disp, int32_t stride,
                                int32_t copy_left, int32_t precision, void* stream) {
    MH_REQUIRE(precision >= 0 && precision <= 2, MH_ERR_ARG, "mh_corr_bwd_prec: precision must be 0 (fp32), 1 (bf16) or 2 (split-bf16: runs as fp32)");
    MH_REQUIRE(g && L && R && dL && dR, MH_ERR_ARG, "mh_corr_bwd: null argument");
    MH_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && max_disp >= 0 && stride >= 1, MH_ERR_ARG, "mh_corr_bwd: bad dimension");
    MH_REQUIRE(C % 4 == 0 && l_ld % 4 == 0 && r_ld % 4 == 0 && dl_ld % 4 == 0 && dr_ld % 4 == 0 &&
               mh_aligned16(L) && mh_aligned16(R) && mh_aligned16(dL) && mh_aligned16(dR), MH_ERR_ALIGN,
               "mh_corr_bwd: channel counts / lds must be multiples of 4 and pointers 16-byte aligned");
    MH_REQUIRE(!copy_left || (g_ld % 4 == 0 && mh_aligned16(g)), MH_ERR_ALIGN, "mh_corr_bwd: copy_left needs aligned g rows");
    CorrBwdArgs a;
    a.g = g; a.L = L; a.R = R; a.dL = dL; a.dR = dR; a.du = du;
    a.g_ld = g_ld; a.coff = coff; a.l_ld = l_ld; a.r_ld = r_ld; a.dl_ld = dl_ld; a.dr_ld = dr_ld;
    a.acc_l = acc_l; a.acc_r = acc_r; a.acc_u = acc_u;
    a.B = B; a.H = H; a.W = W; a.C = C; a.md = max_disp; a.stride = stride; a.D = 2 * max_disp / stride + 1;
    a.copy_left = copy_left;
    a.total = (int64_t)B * H * W * (C / 4);
    if (precision == 1 && g_corr_direct && a.D > MAXD_SMALL && stride == 1 && !copy_left && !du && C % 16 == 0 && C >= 32 && C <= 256 &&
        (C == 32 || C == 64 || C == 128 || C == 256) && (int64_t)B * H * W * std::max(std::max(g_ld, l_ld), r_ld) * 4 < (1ll << 31) - 64) {
        // bf16 operands on v_mfma_f32_16x16x32_bf16 (the arithmetic of the other gradients in the 'mixed' / 'bf16' modes)
        const int nks = (2 * max_disp + 16 + 31) / 32, rows = 48 + 32 * nks, DPh = (((a.D + 3) & ~3) + 7) & ~7;
        const size_t tile = (size_t)64 * (C + 4) * 4;
        const size_t lds_r = std::max(tile, ((size_t)rows * (C + 16) + (size_t)rows * DPh) * 2);       // (the right-gradient workgroups need more: g rows of the whole window)
        if (lds_r <= 150 * 1024) {
            const int segs = mh_cdiv(W, 64);
            const dim3 grid(segs * B * H);
            hipStream_t s = (hipStream_t)stream;
            const int remap = g_corr_remap.load();
#define MH_CORRBH(CSv) { if (g_corr_pair.load()) hipLaunchKernelGGL((corr_bwd_mfma_bf16_pair<CSv>), dim3(2 * segs * B * H), dim3(256), lds_r, s, a, segs, remap);  \
                         else { hipLaunchKernelGGL((corr_bwd_mfma_bf16_one<CSv, false>), dim3(segs * B * H), dim3(256), lds_r, s, a, segs, remap);                   \
                                hipLaunchKernelGGL((corr_bwd_mfma_bf16_one<CSv, true>), dim3(segs * B * H), dim3(256), lds_r, s, a, segs, remap); } }
            switch (C) {
                case 32: MH_CORRBH(2) break;
                case 64: MH_CORRBH(4) break;
                case 128: MH_CORRBH(8) break;
                default: MH_CORRBH(16) break;
            }
#undef MH_CORRBH
            mh_note_kernel("corr_bwd_mfma_bf16_pair<C/16=%d> left + right in %s of %d x 4 waves, lds %d", C / 16, g_corr_pair.load() ? "one grid" : "two grids", (g_corr_pair.load() ? 2 : 1) * segs * B * H, (int)lds_r);
            return mh_check_launch("corr_bwd_mfma_bf16");
        }
    }
    {   // large shift counts: banded GEMM on the MFMA (two launches: left and right gradient)
        const int nb = (2 * max_disp + 31) / 16, rows = 48 + 16 * nb, DP = (a.D + 3) & ~3;
        const size_t lds_l = ((size_t)rows * (C + 4) + 64 * DP) * sizeof(float), lds_r = ((size_t)rows * (C + 4) + (size_t)rows * DP) * sizeof(float);
        if (g_corr_direct && a.D > MAXD_SMALL && stride == 1 && !copy_left && !du && (C == 16 || C == 32 || C == 64 || C == 128 || C == 256) &&
            lds_r <= 150 * 1024 && (int64_t)B * H * W < (1ll << 31) / 64) {
            const int segs = mh_cdiv(W, 64);
            const dim3 grid(segs * B * H);
            hipStream_t s = (hipStream_t)stream;
#define MH_CORRB(CSv)                                                                                        \
            hipLaunchKernelGGL((corr_bwd_mfma<CSv, false>), grid, dim3(CSv >= 2 ? 512 : 256), lds_l, s, a, segs);  \
            hipLaunchKernelGGL((corr_bwd_mfma<CSv, true>), grid, dim3(CSv >= 2 ? 512 : 256), lds_r, s, a, segs);
            switch (C) {
                case 16: MH_CORRB(1) break;
                case 32: MH_CORRB(2) break;
                case 64: MH_CORRB(4) break;
                case 128: MH_CORRB(8) break;
                default: MH_CORRB(16) break;
            }
#undef MH_CORRB
            mh_note_kernel("corr_bwd_mfma<C/16=%d> left + right (exact fp32)", C / 16);
            return mh_check_launch("corr_bwd_mfma");
        }
    }
    {
        const int64_t npix = (int64_t)B * H * W;
        const int64_t ldm = std::max(std::max(std::max(g_ld, l_ld), std::max(r_ld, dl_ld)), dr_ld);
        if (g_corr_direct && a.D <= 5 && npix * ldm * 4 < (1ll << 31) - 64 && a.total < (1ll << 31) - 256) {
            hipLaunchKernelGGL(corr_bwd_direct<5>, dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
            mh_note_kernel("corr_bwd_direct<DT=5> C=%d D=%d", C, a.D);
            return mh_check_launch("corr_bwd_direct");
        }
    }
    int blocks = (int)((a.total + 255) / 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(corr_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    mh_note_kernel("corr_bwd_kernel C=%d D=%d", C, a.D);
    return mh_check_launch("corr_bwd");
}

// this translation unit's copy of the deterministic-accumulation table (mh_common.h)
extern "C" int mh_det_sync_corr(const void* t) {
    g_corr_det_ranges = reinterpret_cast<const mh_det_table*>(t)->n;
    return mh_det_upload(*reinterpret_cast<const mh_det_table*>(t));
}
extern "C" int mh_det_ovf_corr(void) { return mh_det_overflow_take(); }      // this translation unit's saturation flag of the deterministic twin (mh_common.h)
