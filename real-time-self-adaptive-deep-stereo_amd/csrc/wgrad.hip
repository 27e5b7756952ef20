// wgrad.hip -- weight/bias gradient of the conv stacks on the gfx950 matrix cores (exact
// fp32 v_mfma_f32_16x16x4_f32).  Gradient of tf.nn.conv2d / atrous_conv2d + bias_add
// (Nets/sharedLayers.py:54-77) as TF's Conv2DBackpropFilter / BiasAddGrad compute it.
//
// GEMM view per tap:  dW[tap][k][n] += sum_m  X[m][(tap,k)] * dZ[m][n]
//   reduction index m = output pixel (b,oy,ox);  X[m][(tap,k)] = in[b, oy*s+ky*d-pt, ox*s+kx*d-pl, k]
// The pixel axis is split over workgroups (grid = taps x k-tiles x n-tiles x splits) and
// partial tiles are combined with fp32 atomics into the (pre-zeroed) flat gradient buffer.
// LDS tiles are [16 pixels][channels] so both MFMA operands are read conflict-free with the
// lane index running along channels.
#include "mh_common.h"

namespace {

struct WgradArgs {
    const float* in; const float* dz; float* dw; float* db;
    int in_ld, dz_ld;
    int B, Hi, Wi, Ho, Wo, K, N, kh, kw, stride, dil, pad_t, pad_l;
    int M, taps, ktiles, ntiles, splits, chunk;
    int vecA, vecB;
};

constexpr int PT = 16;   // pixels per reduction tile

template <int WM, int WN, int MT, int NT>
__global__ __launch_bounds__(64 * WM * WN) void wgrad_kernel(WgradArgs p) {
    constexpr int NTH = 64 * WM * WN;
    constexpr int BK = WM * MT * 16;     // k (input-channel) rows of the dW tile
    constexpr int BN = WN * NT * 16;
    constexpr int ASs = BK + 4, BSs = BN + 4;
    constexpr int AVEC = PT * BK / 4, BVEC = PT * BN / 4;
    constexpr int AITEMS = (AVEC + NTH - 1) / NTH, BITEMS = (BVEC + NTH - 1) / NTH;

    __shared__ __attribute__((aligned(16))) float As[2][PT * ASs];
    __shared__ __attribute__((aligned(16))) float Bs[2][PT * BSs];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, lq = lane >> 4;

    int bid = blockIdx.x;
    const int split = bid % p.splits; bid /= p.splits;
    const int tn = bid % p.ntiles; bid /= p.ntiles;
    const int tk = bid % p.ktiles; bid /= p.ktiles;
    const int tap = bid;
    const int ky = tap / p.kw, kx = tap - ky * p.kw;
    const int dy = ky * p.dil - p.pad_t, dx = kx * p.dil - p.pad_l;
    const int k0 = tk * BK, n0 = tn * BN;
    const int mbeg = split * p.chunk;
    const int mend = min(p.M, mbeg + p.chunk);
    if (mbeg >= mend) return;
    const int ntile = (mend - mbeg + PT - 1) / PT;
    const int Kr = (p.K + 3) & ~3;

    // per-item pixel cursors for the A loads (advance by PT pixels per tile)
    int a_ox[AITEMS], a_oy[AITEMS], a_b[AITEMS], a_m[AITEMS];
#pragma unroll
    for (int j = 0; j < AITEMS; ++j) {
        const int q = tid + NTH * j;
        const int kp = q / (BK / 4);
        const int m = mbeg + kp;
        a_m[j] = m;
        a_ox[j] = m % p.Wo;
        const int t2 = m / p.Wo;
        a_oy[j] = t2 % p.Ho;
        a_b[j] = t2 / p.Ho;
    }

    float4 ra_v[AITEMS], rb_v[BITEMS];
    int tile_ld = 0;   // tiles loaded so far

    auto load_tile = [&]() {
#pragma unroll
        for (int j = 0; j < AITEMS; ++j) {
            const int q = tid + NTH * j;
            const int c4 = q % (BK / 4);
            const int k = k0 + c4 * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < AVEC && a_m[j] < mend && k < Kr) {
                const int iy = a_oy[j] * p.stride + dy, ix = a_ox[j] * p.stride + dx;
                if (iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi) {
                    const float* src = p.in + (((int64_t)a_b[j] * p.Hi + iy) * p.Wi + ix) * p.in_ld + k;
                    if (p.vecA) {
                        v = *reinterpret_cast<const float4*>(src);
                    } else {
                        if (k + 0 < p.K) v.x = src[0];
                        if (k + 1 < p.K) v.y = src[1];
                        if (k + 2 < p.K) v.z = src[2];
                        if (k + 3 < p.K) v.w = src[3];
                    }
                }
            }
            ra_v[j] = v;
            // advance this item's pixel by PT
            a_m[j] += PT;
            a_ox[j] += PT;
            while (a_ox[j] >= p.Wo) {
                a_ox[j] -= p.Wo;
                if (++a_oy[j] == p.Ho) { a_oy[j] = 0; ++a_b[j]; }
            }
        }
#pragma unroll
        for (int j = 0; j < BITEMS; ++j) {
            const int q = tid + NTH * j;
            const int kp = q / (BN / 4), n4 = q % (BN / 4);
            const int m = mbeg + tile_ld * PT + kp;
            const int n = n0 + n4 * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < BVEC && m < mend && n < p.N) {
                const float* src = p.dz + (int64_t)m * p.dz_ld + n;
                if (p.vecB) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    v.x = src[0];
                    if (n + 1 < p.N) v.y = src[1];
                    if (n + 2 < p.N) v.z = src[2];
                    if (n + 3 < p.N) v.w = src[3];
                }
            }
            rb_v[j] = v;
        }
        ++tile_ld;
    };

    auto store_tile = [&](int buf) {
#pragma unroll
        for (int j = 0; j < AITEMS; ++j) {
            const int q = tid + NTH * j;
            if (q < AVEC) {
                const int kp = q / (BK / 4), c4 = q % (BK / 4);
                *reinterpret_cast<float4*>(&As[buf][kp * ASs + c4 * 4]) = ra_v[j];
            }
        }
#pragma unroll
        for (int j = 0; j < BITEMS; ++j) {
            const int q = tid + NTH * j;
            if (q < BVEC) {
                const int kp = q / (BN / 4), n4 = q % (BN / 4);
                *reinterpret_cast<float4*>(&Bs[buf][kp * BSs + n4 * 4]) = rb_v[j];
            }
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const bool do_bias = (p.db != nullptr) && tap == 0 && tk == 0 && tid < BN;
    float bsum = 0.f;

    load_tile();
    store_tile(0);
    __syncthreads();

    for (int t = 0; t < ntile; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntile) load_tile();
        const float* Ab = &As[buf][lq * ASs + wm * MT * 16 + li];
        const float* Bb = &Bs[buf][lq * BSs + wn * NT * 16 + li];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float a[MT], b[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) a[i] = Ab[ks * 4 * ASs + i * 16];
#pragma unroll
            for (int j = 0; j < NT; ++j) b[j] = Bb[ks * 4 * BSs + j * 16];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (do_bias) {
#pragma unroll
            for (int kp = 0; kp < PT; ++kp) bsum += Bs[buf][kp * BSs + tid];
        }
        if (t + 1 < ntile) store_tile(buf ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = k0 + wm * MT * 16 + i * 16 + lq * 4 + r;
            if (k >= p.K) continue;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = n0 + wn * NT * 16 + j * 16 + li;
                if (n < p.N) atomicAdd(p.dw + ((int64_t)tap * p.K + k) * p.N + n, acc[i][j][r]);
            }
        }
    if (do_bias && n0 + tid < p.N) atomicAdd(p.db + n0 + tid, bsum);
}

template <int WM, int WN, int MT, int NT>
int launch_wgrad(WgradArgs& a, hipStream_t s) {
    constexpr int BK = WM * MT * 16, BN = WN * NT * 16;
    a.ktiles = mh_cdiv(a.K, BK);
    a.ntiles = mh_cdiv(a.N, BN);
    const int base = a.taps * a.ktiles * a.ntiles;
    // enough pixel splits for ~3 workgroups per CU, but keep >= 8 reduction tiles per split
    int splits = mh_cdiv(768, base);
    const int maxs = mh_cdiv(a.M, PT * 8);
    if (splits > maxs) splits = maxs;
    if (splits < 1) splits = 1;
    int chunk = mh_cdiv(a.M, splits);
    chunk = (chunk + PT - 1) / PT * PT;
    a.splits = mh_cdiv(a.M, chunk);
    a.chunk = chunk;
    hipLaunchKernelGGL((wgrad_kernel<WM, WN, MT, NT>), dim3(base * a.splits), dim3(64 * WM * WN), 0, s, a);
    return mh_check_launch("wgrad");
}

}  // namespace

extern "C" int mh_conv2d_wgrad(const mh_conv_desc* d, const float* in, const float* dout, int32_t dout_ld,
                               float* dw, float* db, void* stream) {
    MH_REQUIRE(d && in && dout && dw, MH_ERR_ARG, "mh_conv2d_wgrad: null argument");
    MH_REQUIRE(d->mode == 0, MH_ERR_ARG, "mh_conv2d_wgrad: descriptor must be the forward (mode 0) geometry");
    MH_REQUIRE(d->B > 0 && d->Hi > 0 && d->Wi > 0 && d->Ho > 0 && d->Wo > 0 && d->K > 0 && d->N > 0,
               MH_ERR_ARG, "mh_conv2d_wgrad: non-positive dimension");
    MH_REQUIRE(d->in_ld >= d->K && dout_ld >= d->N, MH_ERR_ARG, "mh_conv2d_wgrad: ld smaller than channel count");
    MH_REQUIRE((int64_t)d->B * d->Ho * d->Wo < (1ll << 31) - 64, MH_ERR_ARG, "mh_conv2d_wgrad: too many pixels");
    WgradArgs a;
    a.in = in; a.dz = dout; a.dw = dw; a.db = db;
    a.in_ld = d->in_ld; a.dz_ld = dout_ld;
    a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.Ho = d->Ho; a.Wo = d->Wo; a.K = d->K; a.N = d->N;
    a.kh = d->kh; a.kw = d->kw; a.stride = d->stride; a.dil = d->dil; a.pad_t = d->pad_t; a.pad_l = d->pad_l;
    a.M = d->B * d->Ho * d->Wo; a.taps = d->kh * d->kw;
    a.vecA = mh_aligned16(in) && (d->in_ld % 4 == 0) && (d->in_ld >= ((d->K + 3) & ~3));
    a.vecB = mh_aligned16(dout) && (dout_ld % 4 == 0) && (d->N % 4 == 0);
    hipStream_t s = (hipStream_t)stream;
    const int K = a.K, N = a.N;
    // dW tile shape from the channel counts (rows = K, cols = N)
    if (K > 64 && N > 64) return launch_wgrad<2, 2, 4, 4>(a, s);      // 128 x 128
    if (K > 64 && N > 32) return launch_wgrad<2, 2, 4, 2>(a, s);      // 128 x 64
    if (K > 64) return launch_wgrad<4, 1, 2, (1)>(a, s);              // 128 x 16  (N <= 32: 2 n-tiles at most)
    if (K > 32 && N > 64) return launch_wgrad<2, 2, 2, 4>(a, s);      // 64 x 128
    if (K > 32 && N > 32) return launch_wgrad<2, 2, 2, 2>(a, s);      // 64 x 64
    if (K > 32) return launch_wgrad<4, 1, 1, 1>(a, s);                // 64 x 16
    if (K > 16 && N > 64) return launch_wgrad<1, 4, 2, 2>(a, s);      // 32 x 128
    if (K > 16 && N > 16) return launch_wgrad<2, 2, 1, 1>(a, s);      // 32 x 32
    if (K > 16) return launch_wgrad<2, 1, 1, 1>(a, s);                // 32 x 16
    if (N > 64) return launch_wgrad<1, 4, 1, 2>(a, s);                // 16 x 128
    if (N > 16) return launch_wgrad<1, 2, 1, 1>(a, s);                // 16 x 32
    return launch_wgrad<1, 1, 1, 1>(a, s);                            // 16 x 16
}
