// wgrad.hip -- weight/bias gradient of the conv stacks on the gfx950 matrix cores: exact fp32 (wgrad_kernel,
// v_mfma_f32_16x16x4_f32) and bf16 inputs / fp32 accumulate (wgrad_bf16_kernel, v_mfma_f32_16x16x32_bf16).  Gradient of
// tf.nn.conv2d / atrous_conv2d + bias_add (Nets/sharedLayers.py:54-77) as TF's Conv2DBackpropFilter / BiasAddGrad compute it.
//
// GEMM view per tap:  dW[tap][k][n] = sum_m  X[m][(tap,k)] * dZ[m][n]
//   reduction index m = output pixel (b,oy,ox);  X[m][(tap,k)] = in[b, oy*s+ky*d-pt, ox*s+kx*d-pl, k]
// The pixel axis is split over workgroups (grid = splits x k-tiles x n-tiles x taps, taps fastest and co-located on one
// XCD so a pixel chunk streams from HBM once).  Two ways to combine the splits: mh_conv2d_wgrad accumulates with fp32
// atomics into a pre-zeroed dW; mh_conv2d_wgrad_partial stores every split's tile to a workspace and mh_wgrad_reduce
// sums them (what the engines use: no atomics, one reduction launch per batch of layers).  Also here: wgrad_n1_kernel
// (single output channel = a pixel reduction) and the tap-flattened mode of the bf16 kernel for Cin <= 4.
#include "mh_common.h"
#include <atomic>
#include <stdlib.h>

namespace {

struct WgradArgs {
    const float* in; const float* dz; float* dw; float* db;
    int in_ld, dz_ld;
    int B, Hi, Wi, Ho, Wo, K, N, kh, kw, stride, dil, pad_t, pad_l;
    int M, taps, ktiles, ntiles, splits, chunk;
    int vecA, vecB;
    unsigned in_bytes, dz_bytes;
    int bf16;              // throughput mode (bf16 MFMA inputs, fp32 accumulate)
    int flat;              // bf16 kernel, Cin <= 4: the tile rows are (tap, channel) pairs -- BK/4 taps per workgroup share ONE dz tile
    float* ws;             // != null: split s stores its partial dW to ws[s][taps*K*N] (no atomics)
    int forced_splits;     // > 0: use exactly this split count (the workspace was sized for it)
    int query;             // 1: compute `splits` only, launch nothing
    int dbg_plain_store;   // timing experiment only: plain stores instead of atomics (WRONG results)
};

// Bias gradients in partial-sum mode (round 6: the product path replays bit-identically).  With a workspace and MORE THAN ONE pixel split the bias partial
// sums of split s go to ws[splits * taps*K*N + s * N + n] -- behind the filter partials, plain stores, summed in split order by the same mh_wgrad_reduce launch
// -- instead of one float atomic per workgroup and channel, whose arrival order changed the last bits from replay to replay.  One split (or no workspace): a
// channel receives a single addend, the atomic onto the zeroed db is already order-free.
__device__ __forceinline__ float* wgrad_bias_ws(const WgradArgs& p) {
    return (p.ws && p.splits > 1) ? p.ws + (int64_t)p.splits * ((int64_t)p.taps * p.K * p.N) : nullptr;
}

static std::atomic<int> g_wgrad_target_wgs{0};
// tuning hook (microbenchmarks): number of workgroups the pixel split aims for (0 = heuristic)
static std::atomic<int> g_wgrad_target_pct{0};     // mh_tune_wgrad_target_pct: scale of the pixel-split workgroup targets while a plan is recorded (0 = default)
extern "C" int mh_tune_wgrad_target_pct(int pct) { return g_wgrad_target_pct.exchange(pct > 0 ? pct : 0); }      // returns the PREVIOUS value (not a status): callers that scope the setting restore it
static std::atomic<int> g_wgrad_plain{0};
static std::atomic<int> g_wgrad_tile64{0};
static std::atomic<int> g_wgrad_w8{0};
extern "C" int mh_tune_wgrad_wgs(int target) {
    g_wgrad_plain = target < 0;                    // negative: timing experiment with plain stores (wrong results)
    if (target < 0) target = -target;
    g_wgrad_tile64 = (target / 100000) == 1;       // + 100000: timing experiment, 64x64 tiles for the 128-wide layers (4x fewer pixel splits)
    g_wgrad_w8 = (target / 100000) == 2 ? 1 : (target / 100000) == 3 ? 2 : 0;      // + 200000 / + 300000: 8-wave workgroups for the 128x128 tile (2x4 / 4x2 waves)
    target %= 100000;
    g_wgrad_target_wgs = target > 1 ? target : 0;
    return 0;
}

// ---- grouped launches (mh_conv2d_wgrad_partial_group): the bf16 tile shapes and the N = 1 reduction get an id; in capture mode the
// launchers fill this record instead of launching, and wgrad_group_kernel runs up to MH_WG_GROUP_MAX layers in one grid ------------
#define MH_WGRAD_CFGS(X) X(0, 2, 2, 4, 4) X(1, 2, 2, 4, 2) X(2, 4, 1, 2, 1) X(3, 2, 2, 2, 4) X(4, 2, 2, 2, 2) X(5, 4, 1, 1, 1) \
    X(6, 1, 4, 2, 2) X(7, 2, 2, 1, 1) X(8, 2, 1, 1, 1) X(9, 1, 4, 1, 2) X(10, 1, 2, 1, 1) X(11, 1, 1, 1, 1)
constexpr int MH_WG_CFG_N1 = 12;
template <int WM, int WN, int MT, int NT>
constexpr int wgrad_cfg_id() {
#define X(id, a, b, c, d) if (WM == a && WN == b && MT == c && NT == d) return id;
    MH_WGRAD_CFGS(X)
#undef X
    return -1;
}
struct WgradCapture { int cfg; int nblocks; size_t lds; };
static thread_local WgradCapture* t_capture = nullptr;     // != null: record (cfg, grid, LDS) of the next dispatch, launch nothing

template <int GPT>
__device__ __forceinline__ int swz_group(int row, int g) { return g ^ ((row >> 2) & (GPT - 1)); }

// LDS tiles are [channel][pixel] (pixel-contiguous, row stride PT+4) so that each lane reads 4
// consecutive reduction indices of its channel with one ds_read_b128 (4 MFMAs per read).  Global
// loads run along channels (NHWC), so the tiles are written transposed with scalar stores; the
// 4-float group index is XOR-swizzled with row bits to spread those stores over the banks.
template <int WM, int WN, int MT, int NT, int PT, bool VEC>
__global__ __launch_bounds__(64 * WM * WN) void wgrad_kernel(WgradArgs p) {
    constexpr int NTH = 64 * WM * WN;
    constexpr int BK = WM * MT * 16;     // k (input-channel) rows of the dW tile
    constexpr int BN = WN * NT * 16;
    constexpr int LS = PT + 4;
    constexpr int GPT = PT / 4;
    constexpr int AVEC = PT * BK / 4, BVEC = PT * BN / 4;
    constexpr int AITEMS = (AVEC + NTH - 1) / NTH, BITEMS = (BVEC + NTH - 1) / NTH;

    HIP_DYNAMIC_SHARED(float, smem)
    float* const As = smem;                    // [2][BK*LS]
    float* const Bs = smem + 2 * BK * LS;      // [2][BN*LS]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, lq = lane >> 4;

    // Workgroups that read the same pixel chunk (all taps / channel tiles of one split) get consecutive
    // logical ids on ONE XCD, so the chunk comes from HBM once and the other taps*ktiles*ntiles-1 reads hit
    // that XCD's L2 (the activations of a layer do not fit the 4 MiB L2: tap-major order re-streamed them
    // from memory once per tap).
    int bid = mh_xcd_remap(blockIdx.x, gridDim.x);
    const int tap = bid % p.taps; bid /= p.taps;
    const int tn = bid % p.ntiles; bid /= p.ntiles;
    const int tk = bid % p.ktiles; bid /= p.ktiles;
    const int split = bid;
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

    const __amdgpu_buffer_rsrc_t rs_in = mh_make_rsrc(p.in, p.in_bytes);
    const __amdgpu_buffer_rsrc_t rs_dz = mh_make_rsrc(p.dz, p.dz_bytes);

    // all tile loads are unconditional buffer loads (out-of-range offset => 0), see mh_common.h
    auto load_tile = [&]() {
#pragma unroll
        for (int j = 0; j < AITEMS; ++j) {
            const int q = tid + NTH * j;
            const int c4 = q % (BK / 4);
            const int k = k0 + c4 * 4;
            const int iy = a_oy[j] * p.stride + dy, ix = a_ox[j] * p.stride + dx;
            const bool ok = (q < AVEC) && (a_m[j] < mend) && (k < Kr) &&
                            (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            const int off = (((a_b[j] * p.Hi + iy) * p.Wi + ix) * p.in_ld + k) * 4;
            float4 v;
            if (VEC) {
                v = mh_buf_load4(rs_in, ok ? off : MH_OOB);
            } else {
                v.x = mh_buf_load1(rs_in, (ok && k + 0 < p.K) ? off : MH_OOB);
                v.y = mh_buf_load1(rs_in, (ok && k + 1 < p.K) ? off + 4 : MH_OOB);
                v.z = mh_buf_load1(rs_in, (ok && k + 2 < p.K) ? off + 8 : MH_OOB);
                v.w = mh_buf_load1(rs_in, (ok && k + 3 < p.K) ? off + 12 : MH_OOB);
            }
            ra_v[j] = v;
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
            const bool ok = (q < BVEC) && (m < mend) && (n < p.N);
            const int off = (m * p.dz_ld + n) * 4;
            float4 v;
            if (VEC) {
                v = mh_buf_load4(rs_dz, ok ? off : MH_OOB);
            } else {
                v.x = mh_buf_load1(rs_dz, ok ? off : MH_OOB);
                v.y = mh_buf_load1(rs_dz, (ok && n + 1 < p.N) ? off + 4 : MH_OOB);
                v.z = mh_buf_load1(rs_dz, (ok && n + 2 < p.N) ? off + 8 : MH_OOB);
                v.w = mh_buf_load1(rs_dz, (ok && n + 3 < p.N) ? off + 12 : MH_OOB);
            }
            rb_v[j] = v;
        }
        ++tile_ld;
    };

    auto store_tile = [&](int buf) {
        float* Ab = As + buf * (BK * LS);
        float* Bb = Bs + buf * (BN * LS);
#pragma unroll
        for (int j = 0; j < AITEMS; ++j) {
            const int q = tid + NTH * j;
            if (q < AVEC) {
                const int kp = q / (BK / 4), c4 = q % (BK / 4);
                float* d = &Ab[(c4 * 4) * LS + swz_group<GPT>(c4 * 4, kp >> 2) * 4 + (kp & 3)];
                d[0] = ra_v[j].x; d[LS] = ra_v[j].y; d[2 * LS] = ra_v[j].z; d[3 * LS] = ra_v[j].w;
            }
        }
#pragma unroll
        for (int j = 0; j < BITEMS; ++j) {
            const int q = tid + NTH * j;
            if (q < BVEC) {
                const int kp = q / (BN / 4), n4 = q % (BN / 4);
                float* d = &Bb[(n4 * 4) * LS + swz_group<GPT>(n4 * 4, kp >> 2) * 4 + (kp & 3)];
                d[0] = rb_v[j].x; d[LS] = rb_v[j].y; d[2 * LS] = rb_v[j].z; d[3 * LS] = rb_v[j].w;
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
        const float* Ab = As + buf * (BK * LS) + (wm * MT * 16 + li) * LS;
        const float* Bb = Bs + buf * (BN * LS) + (wn * NT * 16 + li) * LS;
#pragma unroll
        for (int s = 0; s < PT / 16; ++s) {
            float4 a[MT], b[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i)
                a[i] = *reinterpret_cast<const float4*>(Ab + i * 16 * LS + swz_group<GPT>(wm * MT * 16 + i * 16 + li, s * 4 + lq) * 4);
#pragma unroll
            for (int j = 0; j < NT; ++j)
                b[j] = *reinterpret_cast<const float4*>(Bb + j * 16 * LS + swz_group<GPT>(wn * NT * 16 + j * 16 + li, s * 4 + lq) * 4);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (do_bias) {
            const float* row = Bs + buf * (BN * LS) + tid * LS;   // swizzle permutes within the row: the sum is unaffected
#pragma unroll
            for (int g = 0; g < GPT; ++g) {
                const float4 v = *reinterpret_cast<const float4*>(row + g * 4);
                bsum += (v.x + v.y) + (v.z + v.w);
            }
        }
        if (t + 1 < ntile) store_tile(buf ^ 1);
        __syncthreads();
    }

    float* const dwb = p.ws ? p.ws + (int64_t)split * ((int64_t)p.taps * p.K * p.N) : p.dw;
    const bool plain = (p.ws != nullptr) || p.dbg_plain_store;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = k0 + wm * MT * 16 + i * 16 + lq * 4 + r;
            if (k >= p.K) continue;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = n0 + wn * NT * 16 + j * 16 + li;
                if (n < p.N) {
                    float* d = dwb + ((int64_t)tap * p.K + k) * p.N + n;
                    if (plain) *d = acc[i][j][r]; else mh_atomic_add(d, acc[i][j][r]);
                }
            }
        }
    if (do_bias && n0 + tid < p.N) {
        float* const wsb = wgrad_bias_ws(p);
        if (wsb) wsb[(int64_t)split * p.N + n0 + tid] = bsum; else mh_atomic_add(p.db + n0 + tid, bsum);
    }
}


// ---- bf16 throughput-mode variant ----------------------------------------------------------------------
// Same decomposition, but the reduction tile is 64 pixels, both LDS tiles are bf16 [channel][pixel]
// (pixel-contiguous, row stride 64+8 halfs) and the contraction runs on v_mfma_f32_16x16x32_bf16 (lane
// (i, q) reads the 8 consecutive pixels 8q..8q+7 of its channel with one ds_read_b128).  A "unit" = 4
// consecutive pixels x one 4-channel group: 4 coalesced 16-byte loads along the NHWC rows, transposed in
// registers and rounded to bf16 (v_cvt_pk_bf16_f32), stored as four 8-byte LDS writes.
template <int WM, int WN, int MT, int NT>
__device__ __forceinline__ void wgrad_bf16_body(const WgradArgs& p, const int block_id, const int grid_dim, float* const smem) {
    constexpr int NTH = 64 * WM * WN;
    if ((int)threadIdx.x >= NTH) return;      // grouped launches run every tile shape in 256-thread workgroups (whole waves retire)
    constexpr int PT = 64;
    constexpr int BK = WM * MT * 16, BN = WN * NT * 16;
    constexpr int LS = PT + 8;           // halfs.  (PT + 16 = 40-dword rows would make the operand reads conflict free, but the
                                         // 128x128 tile then needs 82 KB and only ONE workgroup fits a CU: measured 28 % slower)
    constexpr int AUN = (BK / 4) * (PT / 4), BUN = (BN / 4) * (PT / 4);
    constexpr int AU = (AUN + NTH - 1) / NTH, BU = (BUN + NTH - 1) / NTH;

    unsigned short* const Ah = reinterpret_cast<unsigned short*>(smem);   // [2][BK*LS]
    unsigned short* const Bh = Ah + 2 * BK * LS;                         // [2][BN*LS]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, lq = lane >> 4;

    // Workgroups that read the same pixel chunk (all taps / channel tiles of one split) get consecutive
    // logical ids on ONE XCD, so the chunk comes from HBM once and the other taps*ktiles*ntiles-1 reads hit
    // that XCD's L2 (the activations of a layer do not fit the 4 MiB L2: tap-major order re-streamed them
    // from memory once per tap).
    // flat mode (Cin <= 4, e.g. the 7x7 / 3x3 image layers): tile row r = (tap0 + r/4, channel r%4), so TPW = BK/4 taps
    // share one workgroup and ONE dz tile (per-tap workgroups re-read dz once per tap: 49x for the 7x7 layer).
    const int TPW = p.flat ? BK / 4 : 1;
    const int ntapg = (p.taps + TPW - 1) / TPW;
    int bid = mh_xcd_remap(block_id, grid_dim);
    const int tap = (bid % ntapg) * TPW; bid /= ntapg;           // first tap of this workgroup
    const int tn = bid % p.ntiles; bid /= p.ntiles;
    const int tk = bid % p.ktiles; bid /= p.ktiles;
    const int split = bid;
    const int k0 = tk * BK, n0 = tn * BN;
    const int mbeg = split * p.chunk;
    const int mend = min(p.M, mbeg + p.chunk);
    if (mbeg >= mend) return;
    const int ntile = (mend - mbeg + PT - 1) / PT;
    const int Kr = (p.K + 3) & ~3;

    // Pixel table: tab[buf][t][i] = byte offset of the input pixel that reduction-pixel i of a tile reads for tap
    // tap + t of this workgroup, or -1 (padding / past the chunk / past the last tap).  64 threads keep one incremental
    // (b, oy, ox) cursor each; the loaders fetch the 4 offsets of their unit with one ds_read_b128.
    int* const tab = reinterpret_cast<int*>(Bh + 2 * BN * LS);            // [2][TPW][PT]
    int c_m = mbeg + tid, c_ox = 0, c_oy = 0, c_b = 0;
    if (tid < PT) {
        c_ox = c_m % p.Wo;
        const int t2 = c_m / p.Wo;
        c_oy = t2 % p.Ho;
        c_b = t2 / p.Ho;
    }
    auto table_step = [&](int buf) {
        if (tid < PT) {
            for (int t = 0; t < TPW; ++t) {
                const int tp = tap + t;
                const int ky = tp / p.kw, kx = tp - ky * p.kw;
                const int iy = c_oy * p.stride + ky * p.dil - p.pad_t, ix = c_ox * p.stride + kx * p.dil - p.pad_l;
                const bool ok = tp < p.taps && c_m < mend && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
                tab[(buf * TPW + t) * PT + tid] = ok ? ((c_b * p.Hi + iy) * p.Wi + ix) * p.in_ld * 4 : -1;
            }
            c_m += PT;
            c_ox += PT;
            while (c_ox >= p.Wo) {
                c_ox -= p.Wo;
                if (++c_oy == p.Ho) { c_oy = 0; ++c_b; }
            }
        }
    };

    float4 ra_v[AU][4], rb_v[BU][4];
    int tile_ld = 0;
    const __amdgpu_buffer_rsrc_t rs_in = mh_make_rsrc(p.in, p.in_bytes);
    const __amdgpu_buffer_rsrc_t rs_dz = mh_make_rsrc(p.dz, p.dz_bytes);

    auto load_tile = [&]() {
        const int* tb = tab + (tile_ld & 1) * TPW * PT;
#pragma unroll
        for (int j = 0; j < AU; ++j) {
            const int u = tid + NTH * j;
            const int r4 = u % (BK / 4);                          // row group: channel group, or (flat) tap of the workgroup
            const int k = p.flat ? 0 : k0 + r4 * 4;
            const bool uok = (u < AUN) && (k < Kr);
            const int4 o = *reinterpret_cast<const int4*>(tb + (p.flat ? r4 * PT : 0) + ((u / (BK / 4)) & (PT / 4 - 1)) * 4);
            ra_v[j][0] = mh_buf_load4(rs_in, (uok && o.x >= 0) ? o.x + k * 4 : MH_OOB);
            ra_v[j][1] = mh_buf_load4(rs_in, (uok && o.y >= 0) ? o.y + k * 4 : MH_OOB);
            ra_v[j][2] = mh_buf_load4(rs_in, (uok && o.z >= 0) ? o.z + k * 4 : MH_OOB);
            ra_v[j][3] = mh_buf_load4(rs_in, (uok && o.w >= 0) ? o.w + k * 4 : MH_OOB);
        }
#pragma unroll
        for (int j = 0; j < BU; ++j) {
            const int u = tid + NTH * j;
            const int m = mbeg + tile_ld * PT + (u / (BN / 4)) * 4;
            const int n = n0 + (u % (BN / 4)) * 4;
            const bool uok = (u < BUN) && (n < p.N);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                rb_v[j][e] = mh_buf_load4(rs_dz, (uok && m + e < mend) ? ((m + e) * p.dz_ld + n) * 4 : MH_OOB);
        }
        ++tile_ld;
    };

    // LDS position of (row, pixel-group pb): 8-pixel octets are XOR-swizzled by the row's 4-row block (= the unit's
    // channel group): the 16 lanes of a ds_write_b64 group hold 16 different channel groups at the same pixels
    // (rows 4*LS apart -> one bank unswizzled, 2-way with the XOR); operand reads stay conflict free (40-dword rows).
    auto store_unit = [&](unsigned short* base, int row0, int pb, const float4 (&v)[4], bool zero_pad) {
        unsigned short* d = base + row0 * LS + ((((pb >> 1) ^ (row0 >> 2)) & 7) * 2 + (pb & 1)) * 4;
        *reinterpret_cast<uint2*>(d) = make_uint2(mh_pack_bf16(v[0].x, v[1].x), mh_pack_bf16(v[2].x, v[3].x));
        *reinterpret_cast<uint2*>(d + LS) = make_uint2(mh_pack_bf16(v[0].y, v[1].y), mh_pack_bf16(v[2].y, v[3].y));
        *reinterpret_cast<uint2*>(d + 2 * LS) = make_uint2(mh_pack_bf16(v[0].z, v[1].z), mh_pack_bf16(v[2].z, v[3].z));
        *reinterpret_cast<uint2*>(d + 3 * LS) = make_uint2(mh_pack_bf16(v[0].w, v[1].w), mh_pack_bf16(v[2].w, v[3].w));
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int j = 0; j < AU; ++j) {
            const int u = tid + NTH * j;
            if (u < AUN) store_unit(Ah + buf * (BK * LS), (u % (BK / 4)) * 4, u / (BK / 4), ra_v[j], false);
        }
#pragma unroll
        for (int j = 0; j < BU; ++j) {
            const int u = tid + NTH * j;
            if (u < BUN) store_unit(Bh + buf * (BN * LS), (u % (BN / 4)) * 4, u / (BN / 4), rb_v[j], false);
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // bias gradient (exact fp32): the tap-0 / k-tile-0 workgroups sum their dz rows from the loaded registers
    const bool do_bias = (p.db != nullptr) && tap == 0 && tk == 0;
    float4 bsum[BU];
#pragma unroll
    for (int j = 0; j < BU; ++j) bsum[j] = make_float4(0.f, 0.f, 0.f, 0.f);

    table_step(0);
    table_step(1);
    __syncthreads();
    load_tile();
    store_tile(0);
    __syncthreads();
    for (int t = 0; t < ntile; ++t) {
        const int buf = t & 1;
        if (do_bias) {      // registers still hold tile t's dz rows until the next load overwrites them
#pragma unroll
            for (int j = 0; j < BU; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bsum[j].x += rb_v[j][e].x; bsum[j].y += rb_v[j][e].y; bsum[j].z += rb_v[j][e].z; bsum[j].w += rb_v[j][e].w;
                }
        }
        if (t + 1 < ntile) load_tile();
        table_step(buf);            // entries of tile t+2 (buffer last read at the top of iteration t-1)
        const unsigned short* Ab = Ah + buf * (BK * LS) + (wm * MT * 16 + li) * LS;
        const unsigned short* Bb = Bh + buf * (BN * LS) + (wn * NT * 16 + li) * LS;
#pragma unroll
        for (int s = 0; s < PT / 32; ++s) {
            u32x4 a[MT], b[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i)
                a[i] = *reinterpret_cast<const u32x4*>(Ab + i * 16 * LS + (((s * 4 + lq) ^ (((wm * MT + i) * 16 + li) >> 2)) & 7) * 8);
#pragma unroll
            for (int j = 0; j < NT; ++j)
                b[j] = *reinterpret_cast<const u32x4*>(Bb + j * 16 * LS + (((s * 4 + lq) ^ (((wn * NT + j) * 16 + li) >> 2)) & 7) * 8);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = mh_mfma_bf16(a[i], b[j], acc[i][j]);
        }
        if (t + 1 < ntile) store_tile(buf ^ 1);
        __syncthreads();
    }

    float* const dwb = p.ws ? p.ws + (int64_t)split * ((int64_t)p.taps * p.K * p.N) : p.dw;
    const bool plain = (p.ws != nullptr) || p.dbg_plain_store;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = wm * MT * 16 + i * 16 + lq * 4 + r;
            const int k = p.flat ? r : k0 + row;                   // flat: row = (tap offset, channel) -- lq*4 + r has channel r
            const int tp = p.flat ? tap + (row >> 2) : tap;
            if (k >= p.K || tp >= p.taps) continue;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = n0 + wn * NT * 16 + j * 16 + li;
                if (n < p.N) {
                    float* d = dwb + ((int64_t)tp * p.K + k) * p.N + n;
                    if (plain) *d = acc[i][j][r]; else mh_atomic_add(d, acc[i][j][r]);
                }
            }
        }
    if (do_bias) {      // uniform per workgroup.  Reduce through LDS first: one global atomic per output channel
        static_assert(NTH % (BN / 4) == 0, "unit -> channel-group mapping must not depend on j");
        float4 bs = bsum[0];
#pragma unroll
        for (int j = 1; j < BU; ++j) { bs.x += bsum[j].x; bs.y += bsum[j].y; bs.z += bsum[j].z; bs.w += bsum[j].w; }
        float* red = smem;                                      // [NTH / (BN/4)][BN]; the tiles are dead by now
        *reinterpret_cast<float4*>(red + (tid / (BN / 4)) * BN + (tid % (BN / 4)) * 4) = bs;
        __syncthreads();
        if (tid < BN && n0 + tid < p.N) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < NTH / (BN / 4); ++r) t += red[r * BN + tid];
            float* const wsb = wgrad_bias_ws(p);
            if (wsb) wsb[(int64_t)split * p.N + n0 + tid] = t; else mh_atomic_add(p.db + n0 + tid, t);
        }
    }
}

template <int WM, int WN, int MT, int NT>
__global__ __launch_bounds__(64 * WM * WN) void wgrad_bf16_kernel(WgradArgs p) {
    HIP_DYNAMIC_SHARED(float, smem)
    wgrad_bf16_body<WM, WN, MT, NT>(p, (int)blockIdx.x, (int)gridDim.x, smem);
}

template <int WM, int WN, int MT, int NT>
int launch_wgrad_bf16(WgradArgs& a, hipStream_t s) {
    constexpr int BK = WM * MT * 16, BN = WN * NT * 16, PT = 64;
    constexpr size_t tiles_b = (size_t)(2 * (BK + BN) * (PT + 8)) * 2;
    constexpr size_t lds_max = tiles_b + 2 * (BK / 4) * PT * 4;        // flat mode keeps BK/4 pixel tables per buffer
    const size_t lds = tiles_b + 2 * (a.flat ? BK / 4 : 1) * PT * 4;
    static std::atomic<uint64_t> attr_done{0};
    const uint64_t attr_dev = mh_device_bit();
    if (!(attr_done.load(std::memory_order_relaxed) & attr_dev)) {
        if (lds_max > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_bf16_kernel<WM, WN, MT, NT>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
            if (e != hipSuccess) { mh_set_error("wgrad_bf16: hipFuncSetAttribute(%d B LDS): %s", (int)lds_max, hipGetErrorString(e)); return (int)e; }
        }
        attr_done.fetch_or(attr_dev);
    }
    if (a.M < 0) return 0;
    a.ktiles = a.flat ? 1 : mh_cdiv(a.K, BK);
    a.ntiles = mh_cdiv(a.N, BN);
    const int base = (a.flat ? mh_cdiv(a.taps, BK / 4) : a.taps) * a.ktiles * a.ntiles;
    constexpr int units = WM * WN * MT * NT;
    // workgroup targets of the pixel split: 256 / 512 / 1024 by tile size (round 1: 384 / 768 / 1536, tuned while the filter gradients overlapped the
    // input-gradient chain; in the deferred one-lane step they mostly run alone and fewer splits = less partial-sum traffic: 1.961 -> 1.912 ms at 2/3,
    // 1.911 at 1/2, 1.964 at 0.4, 2.03 at 1/3; experiments #34).  mh_tune_wgrad_target_pct scales them.
    const int tuned = g_wgrad_target_pct.load(std::memory_order_relaxed);
    const int env_scale = tuned > 0 ? tuned : 100;
    // (layers with more than 65536 reduction pixels -- several streams batched through one model, DispNet's / the pyramid's full-size layers -- are
    //  throughput bound and keep the round-1 targets: B = 4 batched 834 vs 796 pairs/s)
    const int base_t = a.M > 65536 ? (units >= 32 ? 384 : (units >= 8 ? 768 : 1536)) : (units >= 32 ? 256 : (units >= 8 ? 512 : 1024));
    const int forced_wgs = g_wgrad_target_wgs.load(std::memory_order_relaxed);
    const int target = forced_wgs > 0 ? forced_wgs : base_t * env_scale / 100;
    int splits = a.forced_splits > 0 ? a.forced_splits : mh_cdiv(target, base);
    int maxs = mh_cdiv(a.M, PT * 2);
    constexpr int cap = 192;
    if (maxs > cap) maxs = cap;          // the split reduction walks the splits of an element serially (1280 splits of the flat
                                         // 3->16 layer cost a 190 us single-block tail in wgrad_reduce_kernel)
    if (splits > maxs) splits = maxs;
    if (splits < 1) splits = 1;
    int chunk = mh_cdiv(a.M, splits);
    chunk = (chunk + PT - 1) / PT * PT;
    a.splits = mh_cdiv(a.M, chunk);         // idempotent: forcing the returned count reproduces it
    a.chunk = chunk;
    if (a.query) return 0;
    if (t_capture) { t_capture->cfg = wgrad_cfg_id<WM, WN, MT, NT>(); t_capture->nblocks = base * a.splits; t_capture->lds = lds; return 0; }
    mh_note_kernel("wgrad_bf16_kernel<%d,%d,%d,%d> tile %dx%d splits %d grid %d%s", WM, WN, MT, NT, BK, BN, a.splits, base * a.splits, a.flat ? " flat" : "");
    hipLaunchKernelGGL((wgrad_bf16_kernel<WM, WN, MT, NT>), dim3(base * a.splits), dim3(64 * WM * WN), lds, s, a);
    return mh_check_launch("wgrad_bf16");
}

template <int WM, int WN, int MT, int NT, int PT, bool VEC>
int launch_wgrad_one(WgradArgs& a, hipStream_t s) {
    constexpr int BK = WM * MT * 16, BN = WN * NT * 16;
    constexpr size_t lds = (size_t)(2 * (BK + BN) * (PT + 4)) * sizeof(float);
    static std::atomic<uint64_t> attr_done{0};
    const uint64_t attr_dev = mh_device_bit();
    if (!(attr_done.load(std::memory_order_relaxed) & attr_dev)) {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<WM, WN, MT, NT, PT, VEC>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) { mh_set_error("wgrad: hipFuncSetAttribute(%d B LDS): %s", (int)lds, hipGetErrorString(e)); return (int)e; }
        }
        attr_done.fetch_or(attr_dev);
    }
    if (a.M < 0) return 0;               // mh_init(): attribute set-up only
    a.ktiles = mh_cdiv(a.K, BK);
    a.ntiles = mh_cdiv(a.N, BN);
    const int base = a.taps * a.ktiles * a.ntiles;
    // enough pixel splits for ~3 workgroups per CU, but keep >= 4 reduction tiles per split
    // measured (profiles/r01_microbench.txt): big dW tiles want ~1.5 workgroups per CU, tiny ones (a
    // few MFMAs per reduction tile) need many more to hide their latency
    constexpr int units = WM * WN * MT * NT;
    const int forced_wgs = g_wgrad_target_wgs.load(std::memory_order_relaxed);
    const int target = forced_wgs > 0 ? forced_wgs : (units >= 32 ? 384 : (units >= 8 ? 768 : 1536));
    int splits = a.forced_splits > 0 ? a.forced_splits : mh_cdiv(target, base);
    const int maxs = mh_cdiv(a.M, PT * 4);
    if (splits > maxs) splits = maxs;
    if (splits < 1) splits = 1;
    int chunk = mh_cdiv(a.M, splits);
    chunk = (chunk + PT - 1) / PT * PT;
    a.splits = mh_cdiv(a.M, chunk);         // idempotent: forcing the returned count reproduces it
    a.chunk = chunk;
    if (a.query) return 0;
    if (t_capture) { t_capture->cfg = -1; return 0; }       // exact-fp32 tiles are not grouped
    mh_note_kernel("wgrad_kernel<%d,%d,%d,%d,PT=%d,%s> tile %dx%d splits %d grid %d", WM, WN, MT, NT, PT, VEC ? "vec" : "scalar", BK, BN, a.splits, base * a.splits);
    hipLaunchKernelGGL((wgrad_kernel<WM, WN, MT, NT, PT, VEC>), dim3(base * a.splits), dim3(64 * WM * WN), lds, s, a);
    return mh_check_launch("wgrad");
}

template <int WM, int WN, int MT, int NT, int PT>
int launch_wgrad(WgradArgs& a, hipStream_t s) {
    const bool all = a.M < 0;
    const bool vec = a.vecA && a.vecB;
    int rc = 0;
    if (all || (vec && a.bf16)) { rc = launch_wgrad_bf16<WM, WN, MT, NT>(a, s); if (!all || rc) return rc; }
    if (all || vec) { rc = launch_wgrad_one<WM, WN, MT, NT, PT, true>(a, s); if (!all || rc) return rc; }
    if (all || !vec) { rc = launch_wgrad_one<WM, WN, MT, NT, PT, false>(a, s); if (!all || rc) return rc; }
    return rc;
}

// ---- single-output-channel layers (the disparity heads: 3x3 Cin->1) ----------------------------------------------
// dw[tap][k] = sum_p x[p + tap][k] * dz[p] is a memory-bound reduction, not a GEMM (a 16-column MFMA tile would be
// 15/16 padding and N = 1 defeats the 16-byte dz/weight loads of the tiled kernels).  Thread = (4-channel group g,
// pixel slot): TAPS float4 accumulators in registers, pixels of the split strided over the slots, slots reduced
// through LDS, one store (workspace) or atomic (dw) per element per workgroup.
template <int TAPS>
__device__ __forceinline__ void wgrad_n1_body(const WgradArgs& p, const int block_id, float* const smem, float* const bred) {
    // smem: [slots][TAPS][K] floats
    const int tid = threadIdx.x;
    const int G4 = p.K >> 2;                      // <= 256, power-of-two padded by the launcher: G4p
    const int G4p = p.ktiles;                     // (reused field) padded group count, divides 256
    const int slots = 256 / G4p;
    const int g = tid % G4p, slot = tid / G4p;
    const int split = block_id;
    const int mbeg = split * p.chunk, mend = min(p.M, mbeg + p.chunk);
    const bool gok = g < G4;
    const __amdgpu_buffer_rsrc_t rs_in = mh_make_rsrc(p.in, p.in_bytes);
    float4 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    float bsum = 0.f;
    for (int m = mbeg + slot; m < mend; m += slots) {
        const float dzv = p.dz[(int64_t)m * p.dz_ld];
        const int ox = m % p.Wo;
        const int t2 = m / p.Wo;
        const int oy = t2 % p.Ho, b = t2 / p.Ho;
        bsum += dzv;
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            const int ky = t / p.kw, kx = t - ky * p.kw;
            const int iy = oy * p.stride + ky * p.dil - p.pad_t, ix = ox * p.stride + kx * p.dil - p.pad_l;
            const bool ok = gok && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            const float4 x = mh_buf_load4(rs_in, ok ? (((b * p.Hi + iy) * p.Wi + ix) * p.in_ld + g * 4) * 4 : MH_OOB);
            acc[t].x += x.x * dzv; acc[t].y += x.y * dzv; acc[t].z += x.z * dzv; acc[t].w += x.w * dzv;
        }
    }
    if (gok) {
#pragma unroll
        for (int t = 0; t < TAPS; ++t) *reinterpret_cast<float4*>(smem + ((slot * TAPS + t) * G4 + g) * 4) = acc[t];
    }
    bred[tid] = (g == 0) ? bsum : 0.f;
    __syncthreads();
    float* const dst = p.ws ? p.ws + (int64_t)split * TAPS * p.K : p.dw;
    for (int e = tid; e < TAPS * p.K; e += 256) {
        float v = 0.f;
        for (int sl = 0; sl < slots; ++sl) v += smem[sl * TAPS * p.K + e];
        if (p.ws) dst[e] = v; else mh_atomic_add(dst + e, v);
    }
    if (p.db && tid == 0) {
        float v = 0.f;
        for (int i = 0; i < 256; ++i) v += bred[i];
        float* const wsb = wgrad_bias_ws(p);
        if (wsb) wsb[split] = v; else mh_atomic_add(p.db, v);
    }
}
template <int TAPS>
__global__ __launch_bounds__(256) void wgrad_n1_kernel(WgradArgs p) {
    HIP_DYNAMIC_SHARED(float, smem)
    __shared__ float bred[256];
    wgrad_n1_body<TAPS>(p, (int)blockIdx.x, smem, bred);
}

// One grid for several layers' filter gradients (each segment = the grid the layer's own launch would use, padded to a multiple of 8
// workgroups so that blockIdx & 7 is still the XCD inside every segment).  256-thread workgroups: narrower tile shapes retire their spare waves.
constexpr int MH_WG_GROUP_MAX = 8;
struct WgradGroup {
    int n;
    int blk0[MH_WG_GROUP_MAX + 1];
    int cfg[MH_WG_GROUP_MAX];
    int nblk[MH_WG_GROUP_MAX];
    WgradArgs a[MH_WG_GROUP_MAX];
};
__global__ __launch_bounds__(256) void wgrad_group_kernel(WgradGroup g) {
    HIP_DYNAMIC_SHARED(float, smem)
    __shared__ float bred[256];
    int seg = 0;
#pragma unroll
    for (int q = 1; q < MH_WG_GROUP_MAX; ++q)
        if (q < g.n && g.blk0[q] <= (int)blockIdx.x) seg = q;
    seg = __builtin_amdgcn_readfirstlane(seg);
    // the segment's record by a select chain over static indices: a dynamically indexed by-value argument is copied to scratch
    int blk0 = g.blk0[0], nb = g.nblk[0], cfg = g.cfg[0];
    WgradArgs p = g.a[0];
#pragma unroll
    for (int q = 1; q < MH_WG_GROUP_MAX; ++q)
        if (seg == q) { blk0 = g.blk0[q]; nb = g.nblk[q]; cfg = g.cfg[q]; p = g.a[q]; }
    const int bid = (int)blockIdx.x - blk0;
    if (bid >= nb) return;                       // padding workgroup
    switch (cfg) {
#define X(id, a_, b_, c_, d_) case id: wgrad_bf16_body<a_, b_, c_, d_>(p, bid, nb, smem); break;
        MH_WGRAD_CFGS(X)
#undef X
        case MH_WG_CFG_N1: wgrad_n1_body<9>(p, bid, smem, bred); break;
        default: break;
    }
}

// ---- the image layer (3x3, Cin <= 3, Cout = 16: MADNet's conv1) ---------------------------------------------------------------------------
// dW[(tap, c)][n] = sum over output pixels of x[pixel + tap][c] * dz[pixel][n] is a 16 x 32 x M product with M = 245 760 at 1242x375: 432 results, 0.2 GFLOP,
// and 28 MB of operands.  The tiled kernel spends 20 us on it (167 pixel splits of a 16 x 16 tile, 1503 workgroups) and the reduction of its 167
// partial gradients another 12 -- both at the very END of the step, behind the last input gradient, where nothing hides them.  Here a wave owns 32
// output pixels per MFMA pair: lane (li, lq) gathers dz[8 lq .. + 7][n = li] (A operand) and the two columns li / li + 16 of the im2col row of the
// same 8 pixels (B operands: 8 + 16 independent 4-byte buffer loads, out-of-image taps = out-of-range offsets = 0), rounds to bf16 like the tiled
// kernel and issues two MFMAs; the 16 waves of a workgroup meet in LDS, so a launch leaves ONE partial gradient per workgroup (64 instead of 167) for
// the step's reduction launch.  The bias gradient stays exact fp32: the lanes sum the dz values they load.  History: the first A/B (start of round 4)
// showed nothing -- the 27 us split reduction behind it dominated the tail; with the reduction at ~6 us this launch IS the tail: 256 workgroups (64 .. 192: level with the tiled kernel),
// step 1.465 -> 1.453 ms (profiles/r04_experiments.txt #14).  mh_tune_wgrad_image(0) = the tiled kernel.
constexpr int IMG_WAVES = 16;
static std::atomic<int> g_wgrad_image{256};
extern "C" int mh_tune_wgrad_image(int on) { return g_wgrad_image.exchange(on > 0 ? on : 0); }      // returns the previous setting; > 1 = workgroup count

__global__ __launch_bounds__(64 * IMG_WAVES) void wgrad_image_kernel(WgradArgs p, unsigned mulWo, unsigned mulHo) {
    __shared__ float part[IMG_WAVES][2][256];
    __shared__ float bpart[IMG_WAVES][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const __amdgpu_buffer_rsrc_t rs_in = mh_make_rsrc(p.in, p.in_bytes);
    const __amdgpu_buffer_rsrc_t rs_dz = mh_make_rsrc(p.dz, p.dz_bytes);
    const int ncol = 9 * p.K;
    // this lane's two im2col columns: (tap, channel) -> row / column shift and the element offset from the pixel's own position
    int dy[2], dx[2], coff[2];
    bool colok[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int col = li + 16 * t;
        colok[t] = col < ncol;
        const int tap = colok[t] ? col / p.K : 0;
        const int c = colok[t] ? col - tap * p.K : 0;
        dy[t] = tap / 3 - p.pad_t;
        dx[t] = tap % 3 - p.pad_l;
        coff[t] = ((dy[t] * p.Wi + dx[t]) * p.in_ld + c) * 4;
    }
    const unsigned step_x = (unsigned)(p.stride * p.in_ld * 4), step_z = (unsigned)(p.dz_ld * 4);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    const int m_begin = blockIdx.x * p.chunk;
    const int m_end = (m_begin + p.chunk < p.M) ? m_begin + p.chunk : p.M;          // multiples of 8 (chunk % 32 == 0, Wo % 8 == 0)
    for (int m0 = m_begin + wave * 32; m0 < m_end; m0 += IMG_WAVES * 32) {
        // this lane's 8 pixels m .. m + 7 lie in ONE output row (Wo % 8 == 0) and are live or dead together
        const int m = m0 + lq * 8;
        const bool live = m < m_end;
        const int t2 = (int)__umulhi((unsigned)m, mulWo), ox = m - t2 * p.Wo;
        const int b = (int)__umulhi((unsigned)t2, mulHo), oy = t2 - b * p.Ho;
        const int iy0 = oy * p.stride, ix0 = ox * p.stride;
        const unsigned base = (unsigned)(((b * p.Hi + iy0) * p.Wi + ix0) * p.in_ld * 4);
        const bool row0 = live && colok[0] && (unsigned)(iy0 + dy[0]) < (unsigned)p.Hi;
        const bool row1 = live && colok[1] && (unsigned)(iy0 + dy[1]) < (unsigned)p.Hi;
        unsigned zoff = live ? (unsigned)((m * p.dz_ld + li) * 4) : (unsigned)MH_OOB;      // MH_OOB + 7 steps stays out of range (unsigned)
        unsigned xo0 = base + (unsigned)coff[0], xo1 = base + (unsigned)coff[1];
        int ixa = ix0 + dx[0], ixb = ix0 + dx[1];
        float av[8], b0[8], b1[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            av[i] = mh_buf_load1(rs_dz, (int)zoff);
            b0[i] = mh_buf_load1(rs_in, (row0 && (unsigned)ixa < (unsigned)p.Wi) ? (int)xo0 : MH_OOB);
            b1[i] = mh_buf_load1(rs_in, (row1 && (unsigned)ixb < (unsigned)p.Wi) ? (int)xo1 : MH_OOB);
            zoff += step_z; xo0 += step_x; xo1 += step_x; ixa += p.stride; ixb += p.stride;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) bsum += av[i];
        const u32x4 A = {mh_pack_bf16(av[0], av[1]), mh_pack_bf16(av[2], av[3]), mh_pack_bf16(av[4], av[5]), mh_pack_bf16(av[6], av[7])};
        const u32x4 B0 = {mh_pack_bf16(b0[0], b0[1]), mh_pack_bf16(b0[2], b0[3]), mh_pack_bf16(b0[4], b0[5]), mh_pack_bf16(b0[6], b0[7])};
        const u32x4 B1 = {mh_pack_bf16(b1[0], b1[1]), mh_pack_bf16(b1[2], b1[3]), mh_pack_bf16(b1[4], b1[5]), mh_pack_bf16(b1[6], b1[7])};
        acc0 = mh_mfma_bf16(A, B0, acc0);            // D[n = 4 lq + r][column li]
        acc1 = mh_mfma_bf16(A, B1, acc1);            // D[n][column li + 16]
    }
    // the workgroup's partial gradient: part[wave][t][column * 16 + n]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        part[wave][0][li * 16 + lq * 4 + r] = acc0[r];
        part[wave][1][li * 16 + lq * 4 + r] = acc1[r];
    }
    bsum += __shfl_xor(bsum, 16);
    bsum += __shfl_xor(bsum, 32);
    if (lane < 16) bpart[wave][lane] = bsum;
    __syncthreads();
    const int e = threadIdx.x;
    if (e < 512) {
        const int t = e >> 8, idx = e & 255;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < IMG_WAVES; ++w) v += part[w][t][idx];
        const int col = t * 16 + (idx >> 4), n = idx & 15;
        if (col < ncol) p.ws[(int64_t)blockIdx.x * (ncol * 16) + col * 16 + n] = v;
    } else if (e < 512 + 16 && p.db) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < IMG_WAVES; ++w) v += bpart[w][e - 512];
        float* const wsb = wgrad_bias_ws(p);
        if (wsb) wsb[(int64_t)blockIdx.x * 16 + (e - 512)] = v; else mh_atomic_add(p.db + (e - 512), v);
    }
}

static bool wgrad_image_ok(const WgradArgs& a) {
    return g_wgrad_image.load(std::memory_order_relaxed) > 0 && a.bf16 && a.kh == 3 && a.kw == 3 && a.dil == 1 && a.K * 9 <= 32 && a.N == 16 &&
           (a.ws || a.query) && !a.dw && a.M >= 8192 && (a.stride == 1 || a.stride == 2) && a.Wo % 8 == 0 &&
           (int64_t)a.M * (a.Wo > a.Ho ? a.Wo : a.Ho) < (1ll << 32);          // (the multiply-high divisions of the pixel index are exact below that)
}

static int launch_wgrad_image(WgradArgs& a, hipStream_t s) {
    const int tuned = g_wgrad_image.load(std::memory_order_relaxed);
    int splits = a.forced_splits > 0 ? a.forced_splits : (tuned > 1 ? tuned : 64);
    const int maxs = mh_cdiv(a.M, 32 * IMG_WAVES);            // at least one 32-pixel step per wave
    if (splits > maxs) splits = maxs;
    if (splits < 1) splits = 1;
    int chunk = mh_cdiv(a.M, splits);
    chunk = (chunk + 31) / 32 * 32;
    a.splits = mh_cdiv(a.M, chunk);                           // idempotent: forcing the returned count reproduces it
    a.chunk = chunk;
    if (a.query) return 0;
    if (t_capture) { t_capture->cfg = -1; t_capture->nblocks = 0; t_capture->lds = 0; return 0; }      // never part of a grouped launch
    mh_note_kernel("wgrad_image_kernel K=%d splits %d grid %d x %d waves", a.K, a.splits, a.splits, IMG_WAVES);
    const unsigned mulWo = (unsigned)(((1ull << 32) + a.Wo - 1) / a.Wo), mulHo = (unsigned)(((1ull << 32) + a.Ho - 1) / a.Ho);
    hipLaunchKernelGGL(wgrad_image_kernel, dim3(a.splits), dim3(64 * IMG_WAVES), 0, s, a, mulWo, mulHo);
    return mh_check_launch("wgrad_image");
}

static bool wgrad_n1_ok(const WgradArgs& a) {
    return a.N == 1 && a.taps == 9 && a.vecA && (a.K % 4 == 0) && a.K <= 1024 && a.M > 0;
}

static int launch_wgrad_n1(WgradArgs& a, hipStream_t s) {
    int g4p = 1;
    while (g4p < a.K / 4) g4p <<= 1;
    const int slots = 256 / g4p;
    // ~8 pixels per slot per workgroup, at most 1024 workgroups
    int splits = a.forced_splits > 0 ? a.forced_splits : mh_cdiv(a.M, slots * 8);
    if (splits > 1024) splits = 1024;
    if (splits < 1) splits = 1;
    const int chunk = mh_cdiv(a.M, splits);
    a.splits = mh_cdiv(a.M, chunk);
    a.chunk = chunk;
    a.ktiles = g4p;
    if (a.query) return 0;
    const size_t lds = (size_t)slots * 9 * a.K * sizeof(float);          // <= 36 KiB
    if (t_capture) { t_capture->cfg = MH_WG_CFG_N1; t_capture->nblocks = a.splits; t_capture->lds = lds; return 0; }
    hipLaunchKernelGGL(wgrad_n1_kernel<9>, dim3(a.splits), dim3(256), lds, s, a);
    return mh_check_launch("wgrad_n1");
}

static int wgrad_dispatch(WgradArgs& a, hipStream_t s) {
    const int K = a.flat ? a.taps * 4 : a.K, N = a.N;        // flat: the dW tile rows are (tap, channel) pairs
    const bool all = a.M < 0;
    int rc = 0;
#define MH_WG(cond, ...)                                       \
    if (all || (cond)) {                                       \
        rc = launch_wgrad<__VA_ARGS__>(a, s);                  \
        if (!all || rc) return rc;                             \
    }
    // dW tile shape from the channel counts (rows = K = Cin, cols = N = Cout)
    MH_WG(K > 64 && N > 64 && g_wgrad_tile64, 2, 2, 2, 2, 32)   // (experiment) 64 x 64 tiles
    MH_WG(K > 64 && N > 64 && g_wgrad_w8 == 1, 2, 4, 4, 2, 32)   // (experiment) 128 x 128 tile, 8 waves of 64 x 32
    // 128 x 128 tile with 8 waves of 32 x 64 for the layers that are launched on their own (> 4096 reduction pixels): 126 VGPRs instead of
    // 208 -> 4 waves per SIMD instead of 2 for the same two workgroups per CU; 32.4 -> 28.6 us at 96x320 (profiles/r02_microbench_wgrad_tiles.txt;
    // the 128x64 / 64x128 tiles gain nothing from 8 waves).  Smaller layers keep the 4-wave shape: it is the one the grouped launch runs.
    MH_WG(K > 64 && N > 64 && a.M > 4096 && a.bf16 && g_wgrad_w8 != 3, 4, 2, 2, 4, 32)
    MH_WG(K > 64 && N > 64, 2, 2, 4, 4, 32)                 // 128 x 128
    MH_WG(K > 64 && N > 32 && N <= 64, 2, 2, 4, 2, 32)      // 128 x 64
    MH_WG(K > 64 && N <= 32, 4, 1, 2, 1, 32)                // 128 x 16
    MH_WG(K > 32 && K <= 64 && N > 64, 2, 2, 2, 4, 32)      // 64 x 128
    MH_WG(K > 32 && K <= 64 && N > 32 && N <= 64, 2, 2, 2, 2, 32)   // 64 x 64
    MH_WG(K > 32 && K <= 64 && N <= 32, 4, 1, 1, 1, 32)     // 64 x 16
    MH_WG(K > 16 && K <= 32 && N > 64, 1, 4, 2, 2, 32)      // 32 x 128
    MH_WG(K > 16 && K <= 32 && N > 16 && N <= 64, 2, 2, 1, 1, 32)   // 32 x 32
    MH_WG(K > 16 && K <= 32 && N <= 16, 2, 1, 1, 1, 32)     // 32 x 16
    MH_WG(K <= 16 && N > 64, 1, 4, 1, 2, 32)                // 16 x 128
    MH_WG(K <= 16 && N > 16 && N <= 64, 1, 2, 1, 1, 32)     // 16 x 32
    MH_WG(K <= 16 && N <= 16, 1, 1, 1, 1, 32)               // 16 x 16
#undef MH_WG
    return 0;
}

}  // namespace

constexpr int MH_WG_GROUP_LDS = 96 * 1024;       // >= the largest tile shape's need (128x128 tiles, flat tables: 90 112 B)
int mh_wgrad_init() {
    WgradArgs a{};
    a.M = -1; a.K = 1; a.N = 1;
    if (int rc = wgrad_dispatch(a, nullptr)) return rc;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_group_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, MH_WG_GROUP_LDS);
    if (e != hipSuccess) { mh_set_error("wgrad_group: hipFuncSetAttribute(%d B LDS): %s", MH_WG_GROUP_LDS, hipGetErrorString(e)); return (int)e; }
    return 0;
}

static int wgrad_entry(const mh_conv_desc* d, const float* in, const float* dout, int32_t dout_ld, float* dw, float* db,
                       float* ws, int forced_splits, int query, int* splits_out, void* stream,
                       WgradArgs* out_args = nullptr, WgradCapture* cap = nullptr) {
    MH_REQUIRE(d && in && dout && (dw || ws || query), MH_ERR_ARG, "mh_conv2d_wgrad: null argument");
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
    a.dbg_plain_store = g_wgrad_plain;
    a.bf16 = (d->precision == 1);
    // measured: the 7x7 image layer of DispNet 210 -> ~60 us per tower, but MADNet's 3x3 one is 1 % slower flat (9 taps only
    // re-read dz 9x from L2, and the flat tables cost more than they save) -> many-tap layers only
    a.flat = (a.bf16 && a.vecA && a.vecB && d->K <= 4 && a.taps >= 16) ? 1 : 0;
    a.ws = ws; a.forced_splits = forced_splits; a.query = query;
    {
        const int64_t inb = (((int64_t)d->B * d->Hi * d->Wi - 1) * d->in_ld + (int64_t)((d->K + 3) / 4) * 4) * 4;
        const int64_t dzb = (((int64_t)a.M - 1) * dout_ld + d->N) * 4;
        MH_REQUIRE(inb < (1ll << 31) - 64 && dzb < (1ll << 31) - 64, MH_ERR_UNSUPPORTED, "mh_conv2d_wgrad: tensors must be < 2 GiB");
        a.in_bytes = (unsigned)inb; a.dz_bytes = (unsigned)dzb;
    }
    t_capture = cap;
    const int rc = wgrad_n1_ok(a) ? launch_wgrad_n1(a, (hipStream_t)stream)
                   : wgrad_image_ok(a) ? launch_wgrad_image(a, (hipStream_t)stream) : wgrad_dispatch(a, (hipStream_t)stream);
    t_capture = nullptr;
    if (splits_out) *splits_out = a.splits;
    if (out_args) *out_args = a;
    return rc;
}

extern "C" int mh_conv2d_wgrad(const mh_conv_desc* d, const float* in, const float* dout, int32_t dout_ld,
                               float* dw, float* db, void* stream) {
    return wgrad_entry(d, in, dout, dout_ld, dw, db, nullptr, 0, 0, nullptr, stream);
}

extern "C" int mh_conv2d_wgrad_partial(const mh_conv_desc* d, const float* in, const float* dout, int32_t dout_ld,
                                       float* ws, int32_t* splits, float* db, void* stream) {
    MH_REQUIRE(splits, MH_ERR_ARG, "mh_conv2d_wgrad_partial: splits must not be null");
    if (!ws) return wgrad_entry(d, in, dout, dout_ld, nullptr, nullptr, nullptr, 0, 1, splits, stream);   // query
    MH_REQUIRE(*splits > 0, MH_ERR_ARG, "mh_conv2d_wgrad_partial: *splits must come from a query call (ws = NULL)");
    int used = 0;
    const int want = *splits;
    int rc = wgrad_entry(d, in, dout, dout_ld, nullptr, nullptr, nullptr, want, 1, &used, stream);
    if (rc) return rc;
    MH_REQUIRE(used == want, MH_ERR_ARG, "mh_conv2d_wgrad_partial: split count %d does not match this geometry (%d)", want, used);
    return wgrad_entry(d, in, dout, dout_ld, nullptr, db, ws, want, 0, nullptr, stream);
}

// Several layers' partial filter gradients in ONE launch (they are independent: different workspaces, different biases).  A step of the
// engines issues the filter gradients of a whole pyramid level / estimator as one batch; at 1/16-1/64 resolution each of those launches
// is a handful of workgroups that costs its dispatch latency, not its work.  Exact-fp32 layers and leftovers go out one by one.
extern "C" int mh_conv2d_wgrad_partial_group(const mh_wgrad_item* items, int32_t n, void* stream) {
    MH_REQUIRE(items && n > 0, MH_ERR_ARG, "mh_conv2d_wgrad_partial_group: empty item list");
    WgradGroup G;            // ~1.5 KB: built on the host, passed by value as the kernel argument
    G.n = 0; G.blk0[0] = 0;
    size_t lds = 0;
    int first = -1;
    auto single = [&](const mh_wgrad_item& it) -> int {
        int32_t sp = it.splits;
        return mh_conv2d_wgrad_partial(&it.d, it.in, it.dout, it.dout_ld, it.ws, &sp, it.db, stream);
    };
    auto flush = [&]() -> int {
        int rc = 0;
        if (G.n == 1) rc = single(items[first]);
        else if (G.n > 1) {
            mh_note_kernel("wgrad_group_kernel layers %d grid %d lds %d", G.n, G.blk0[G.n], (int)lds);
            hipLaunchKernelGGL(wgrad_group_kernel, dim3(G.blk0[G.n]), dim3(256), lds, (hipStream_t)stream, G);
            rc = mh_check_launch("wgrad_group");
        }
        G.n = 0; G.blk0[0] = 0; lds = 0; first = -1;
        return rc;
    };
    int batch_max_m = 0;
    for (int i = 0; i < n; ++i) if (items[i].group_max_m > batch_max_m) batch_max_m = items[i].group_max_m;
    for (int i = 0; i < n; ++i) {
        const mh_wgrad_item& it = items[i];
        MH_REQUIRE(it.ws && it.splits > 0, MH_ERR_ARG, "mh_conv2d_wgrad_partial_group: item %d: ws / splits must come from a query call", i);
        int used = 0;
        if (int rc = wgrad_entry(&it.d, it.in, it.dout, it.dout_ld, nullptr, nullptr, nullptr, it.splits, 1, &used, stream)) return rc;
        MH_REQUIRE(used == it.splits, MH_ERR_ARG, "mh_conv2d_wgrad_partial_group: item %d: split count %d does not match this geometry (%d)", i, it.splits, used);
        WgradArgs a; WgradCapture c{-1, 0, 0};
        if (int rc = wgrad_entry(&it.d, it.in, it.dout, it.dout_ld, nullptr, it.db, it.ws, it.splits, 0, nullptr, stream, &a, &c)) return rc;
        // grouped: layers whose own launch is dispatch-latency bound (<= max_m reduction pixels: 1/16 resolution and below).  The big layers
        // fill the chip alone, and as one long grid they only coarsen the interleaving with the input-gradient chain (measured: +1.5 %
        // step time with everything grouped); 1- and 2-wave tile shapes would idle most of a 256-thread workgroup.
        // (cap re-swept at the end of round 2, experiments #35: 16384: 1.918 ms | 8192: 1.903 | 4096: 1.899 | 2048: 1.905)
        const int max_m = batch_max_m > 0 ? batch_max_m : 4096;
        const bool narrow = (c.cfg == 8 || c.cfg == 10 || c.cfg == 11);
        if (c.cfg < 0 || narrow || a.M > max_m || c.lds > (size_t)MH_WG_GROUP_LDS || c.nblocks <= 0) {
            if (int rc = single(it)) return rc;
            continue;
        }
        if (G.n == 0) first = i;
        G.cfg[G.n] = c.cfg; G.nblk[G.n] = c.nblocks; G.a[G.n] = a;
        G.blk0[G.n + 1] = G.blk0[G.n] + ((c.nblocks + 7) & ~7);
        if (c.lds > lds) lds = c.lds;
        if (++G.n == MH_WG_GROUP_MAX) { if (int rc = flush()) return rc; }
    }
    return flush();
}

// ---- reduction of the per-split partial filter gradients (one launch for every layer of a step) --------
namespace {
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const mh_wgrad_seg* __restrict__ segs, int nseg) {
    // block -> segment: segs[].blk0 is the exclusive prefix of the segments' block counts
    const int lo = mh_find_seg(segs, nseg, (int)blockIdx.x);
    const mh_wgrad_seg sg = segs[lo];
    if (sg.size <= 1024 && sg.splits >= 32 && (sg.size & 3) == 0) {
        // a small gradient with many splits (the 3-channel image layer: 432 values x 167 splits) is ONE block here.  Round 3 walked the splits of 64
        // elements at a time, 16 split groups deep: seven passes of ~11 DEPENDENT loads = 27 us at the very end of the step (device time stamps,
        // round 4).  Now every float4 column of the gradient gets 256 / columns split groups at once and a thread keeps 8 loads in flight:
        // 108 columns x 2 groups x 84 splits = 11 round trips.
        __shared__ float4 part[256];
        const int ncol = sg.size >> 2;                                  // <= 256 float4 columns
        const int G = 256 / ncol;                                       // split groups (>= 1)
        const int col = threadIdx.x % ncol, grp = threadIdx.x / ncol;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (grp < G) {
            const float* src = sg.ws + col * 4;
            float4 acc[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            int sp = grp;
            for (; sp + 15 * G < sg.splits; sp += 16 * G) {         // 16 loads in flight (256 partials of the image layer: 8 rounds instead of 16)
                float4 v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const float4*>(src + (int64_t)(sp + u * G) * sg.size);
#pragma unroll
                for (int u = 0; u < 16; ++u) { acc[u & 7].x += v[u].x; acc[u & 7].y += v[u].y; acc[u & 7].z += v[u].z; acc[u & 7].w += v[u].w; }
            }
            for (; sp + 7 * G < sg.splits; sp += 8 * G) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(src + (int64_t)(sp + u * G) * sg.size);
#pragma unroll
                for (int u = 0; u < 8; ++u) { acc[u].x += v[u].x; acc[u].y += v[u].y; acc[u].z += v[u].z; acc[u].w += v[u].w; }
            }
            for (; sp < sg.splits; sp += G) {
                const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)sp * sg.size);
                acc[0].x += v.x; acc[0].y += v.y; acc[0].z += v.z; acc[0].w += v.w;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { t.x += acc[u].x; t.y += acc[u].y; t.z += acc[u].z; t.w += acc[u].w; }
        }
        part[threadIdx.x] = t;
        __syncthreads();
        if (threadIdx.x < ncol) {
            float4 u = part[threadIdx.x];
            for (int g2 = 1; g2 < G; ++g2) { const float4 v = part[g2 * ncol + threadIdx.x]; u.x += v.x; u.y += v.y; u.z += v.z; u.w += v.w; }
            float* d = sg.dst + threadIdx.x * 4;
            if (sg.accumulate) { d[0] += u.x; d[1] += u.y; d[2] += u.z; d[3] += u.w; }
            else { d[0] = u.x; d[1] = u.y; d[2] = u.z; d[3] = u.w; }
        }
        return;
    }
    const int e0 = ((int)blockIdx.x - sg.blk0) * 1024 + threadIdx.x * 4;
    if (e0 >= sg.size) return;
    if ((sg.size & 3) == 0) {       // every split slice 16-byte aligned (ws is): vector path
        const float* src = sg.ws + e0;
        // 4 independent accumulators = 4 loads in flight per lane (a single dependent chain ran at 2.6 TB/s); with many splits 16 at a time first:
        // a 64 - 170-split layer of the step's LAST batch was 16 - 40 dependent rounds of four loads, all of it exposed behind the join (round 4)
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f), t1 = t, t2 = t, t3 = t;
        int s = 0;
        for (; s + 16 <= sg.splits; s += 16) {
            float4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const float4*>(src + (int64_t)(s + u) * sg.size);
#pragma unroll
            for (int u = 0; u < 16; u += 4) {
                t.x += v[u].x; t.y += v[u].y; t.z += v[u].z; t.w += v[u].w;
                t1.x += v[u + 1].x; t1.y += v[u + 1].y; t1.z += v[u + 1].z; t1.w += v[u + 1].w;
                t2.x += v[u + 2].x; t2.y += v[u + 2].y; t2.z += v[u + 2].z; t2.w += v[u + 2].w;
                t3.x += v[u + 3].x; t3.y += v[u + 3].y; t3.z += v[u + 3].z; t3.w += v[u + 3].w;
            }
        }
        for (; s + 4 <= sg.splits; s += 4) {
            const float4 v0 = *reinterpret_cast<const float4*>(src + (int64_t)(s + 0) * sg.size);
            const float4 v1 = *reinterpret_cast<const float4*>(src + (int64_t)(s + 1) * sg.size);
            const float4 v2 = *reinterpret_cast<const float4*>(src + (int64_t)(s + 2) * sg.size);
            const float4 v3 = *reinterpret_cast<const float4*>(src + (int64_t)(s + 3) * sg.size);
            t.x += v0.x; t.y += v0.y; t.z += v0.z; t.w += v0.w;
            t1.x += v1.x; t1.y += v1.y; t1.z += v1.z; t1.w += v1.w;
            t2.x += v2.x; t2.y += v2.y; t2.z += v2.z; t2.w += v2.w;
            t3.x += v3.x; t3.y += v3.y; t3.z += v3.z; t3.w += v3.w;
        }
        for (; s < sg.splits; ++s) {
            const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)s * sg.size);
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
        t.x += (t1.x + t2.x) + t3.x; t.y += (t1.y + t2.y) + t3.y; t.z += (t1.z + t2.z) + t3.z; t.w += (t1.w + t2.w) + t3.w;
        float* d = sg.dst + e0;
        if (sg.accumulate) { d[0] += t.x; d[1] += t.y; d[2] += t.z; d[3] += t.w; }
        else { d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w; }
    } else {
        for (int e = e0; e < min(e0 + 4, sg.size); ++e) {
            float t = 0.f;
            for (int s = 0; s < sg.splits; ++s) t += sg.ws[(int64_t)s * sg.size + e];
            if (sg.accumulate) sg.dst[e] += t; else sg.dst[e] = t;
        }
    }
}
}  // namespace

extern "C" int mh_wgrad_reduce(const mh_wgrad_seg* segs_device, int32_t nseg, int32_t nblocks, void* stream) {
    MH_REQUIRE(segs_device && nseg > 0 && nblocks > 0, MH_ERR_ARG, "mh_wgrad_reduce: empty segment table");
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, segs_device, nseg);
    return mh_check_launch("wgrad_reduce");
}

// this translation unit's copy of the deterministic-accumulation table (mh_common.h)
extern "C" int mh_det_sync_wgrad(const void* t) { return mh_det_upload(*reinterpret_cast<const mh_det_table*>(t)); }
extern "C" int mh_det_ovf_wgrad(void) { return mh_det_overflow_take(); }      // this translation unit's saturation flag of the deterministic twin (mh_common.h)
