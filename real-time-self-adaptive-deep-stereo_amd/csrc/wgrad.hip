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

static int g_wgrad_target_wgs = 0;
// tuning hook (microbenchmarks): number of workgroups the pixel split aims for (0 = heuristic)
static std::atomic<int> g_wgrad_target_pct{0};     // mh_tune_wgrad_target_pct: scale of the pixel-split workgroup targets while a plan is recorded (0 = default)
extern "C" int mh_tune_wgrad_target_pct(int pct) { g_wgrad_target_pct = pct > 0 ? pct : 0; return 0; }
static int g_wgrad_plain = 0;
static int g_wgrad_tile64 = 0;
static int g_wgrad_w8 = 0;
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
                    if (plain) *d = acc[i][j][r]; else atomicAdd(d, acc[i][j][r]);
                }
            }
        }
    if (do_bias && n0 + tid < p.N) atomicAdd(p.db + n0 + tid, bsum);
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
                    if (plain) *d = acc[i][j][r]; else atomicAdd(d, acc[i][j][r]);
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
            atomicAdd(p.db + n0 + tid, t);
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
    static bool attr_done = false;
    if (!attr_done) {
        if (lds_max > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_bf16_kernel<WM, WN, MT, NT>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
            if (e != hipSuccess) { mh_set_error("wgrad_bf16: hipFuncSetAttribute(%d B LDS): %s", (int)lds_max, hipGetErrorString(e)); return (int)e; }
        }
        attr_done = true;
    }
    if (a.M < 0) return 0;
    a.ktiles = a.flat ? 1 : mh_cdiv(a.K, BK);
    a.ntiles = mh_cdiv(a.N, BN);
    const int base = (a.flat ? mh_cdiv(a.taps, BK / 4) : a.taps) * a.ktiles * a.ntiles;
    constexpr int units = WM * WN * MT * NT;
    // workgroup targets of the pixel split: 256 / 512 / 1024 by tile size (round 1: 384 / 768 / 1536, tuned while the filter gradients overlapped the
    // input-gradient chain; in the deferred one-lane step they mostly run alone and fewer splits = less partial-sum traffic: 1.961 -> 1.912 ms at 2/3,
    // 1.911 at 1/2, 1.964 at 0.4, 2.03 at 1/3; experiments #34).  MH_WGRAD_TARGET_PCT scales them (A/B hook).
    static const int env_scale0 = []() { const char* e = getenv("MH_WGRAD_TARGET_PCT"); return e ? atoi(e) : 100; }();
    const int tuned = g_wgrad_target_pct.load(std::memory_order_relaxed);
    const int env_scale = tuned > 0 ? tuned : env_scale0;
    // (layers with more than 65536 reduction pixels -- several streams batched through one model, DispNet's / the pyramid's full-size layers -- are
    //  throughput bound and keep the round-1 targets: B = 4 batched 834 vs 796 pairs/s)
    const int base_t = a.M > 65536 ? (units >= 32 ? 384 : (units >= 8 ? 768 : 1536)) : (units >= 32 ? 256 : (units >= 8 ? 512 : 1024));
    const int target = g_wgrad_target_wgs > 0 ? g_wgrad_target_wgs : base_t * env_scale / 100;
    int splits = a.forced_splits > 0 ? a.forced_splits : mh_cdiv(target, base);
    int maxs = mh_cdiv(a.M, PT * 2);
    static const int cap = []() { const char* e = getenv("MH_WGRAD_MAXSPLITS"); return e ? atoi(e) : 192; }();   // A/B hook
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
    static bool attr_done = false;
    if (!attr_done) {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<WM, WN, MT, NT, PT, VEC>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) { mh_set_error("wgrad: hipFuncSetAttribute(%d B LDS): %s", (int)lds, hipGetErrorString(e)); return (int)e; }
        }
        attr_done = true;
    }
    if (a.M < 0) return 0;               // mh_init(): attribute set-up only
    a.ktiles = mh_cdiv(a.K, BK);
    a.ntiles = mh_cdiv(a.N, BN);
    const int base = a.taps * a.ktiles * a.ntiles;
    // enough pixel splits for ~3 workgroups per CU, but keep >= 4 reduction tiles per split
    // measured (profiles/r01_microbench.txt): big dW tiles want ~1.5 workgroups per CU, tiny ones (a
    // few MFMAs per reduction tile) need many more to hide their latency
    constexpr int units = WM * WN * MT * NT;
    const int target = g_wgrad_target_wgs > 0 ? g_wgrad_target_wgs : (units >= 32 ? 384 : (units >= 8 ? 768 : 1536));
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
        if (p.ws) dst[e] = v; else atomicAdd(dst + e, v);
    }
    if (p.db && tid == 0) {
        float v = 0.f;
        for (int i = 0; i < 256; ++i) v += bred[i];
        atomicAdd(p.db, v);
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

// ---- "taps" kernel: all nine taps of a stride-1 3x3 layer in ONE workgroup, operands through the LDS transposing read --------------------------
// The tiled kernel above gives every tap its own workgroup: each of them pulls the same dz pixels (and a shifted copy of the same input pixels)
// out of L2 as fp32, converts them, transposes them in registers into [channel][pixel] LDS tiles.  Here a workgroup owns dW[9 taps][32 input
// channels][128 output channels] (144 accumulator registers per lane, 4 MFMA waves as 2 x 2: wave (wr, wc) = input channels 16 wr..+15, all taps,
// output channels 64 wc..+63) and walks its share of the reduction in segments of 32 consecutive pixels of one image row (of one dilation sub-lattice):
//   * per segment the dz pixels (32 x 128) and ONE new input row (34 x 32: the 3-row halo patch lives in a ring of 8 row slots, a vertical run
//     of segments re-uses two of its three rows) are loaded once, rounded to bf16 and stored pixel-major, channels contiguous -- as they sit in memory;
//   * ds_read_b64_tr_b16 (mh_lds_read_tr16) turns 4 pixel rows x 16 channels into the MFMA operand order, so a tap is just a row offset into the
//     patch: 9 A fragments + 4 B fragments feed 36 MFMAs per wave and segment; global loads per flop drop ~6x against the tiled kernel;
//   * the reduction index inside a segment is permuted (pixel 16 r + 4 lq + j for lane quad lq, read r, element j -- the same for both operands, so
//     the sum is unchanged): one read then covers 16 consecutive pixel rows of a [pixel][16 channels] tile.
// Partial sums go to the split workspace like the tiled kernel's (or, without one, fp32 atomics).  Opt-in: MH_WGRAD_TAPS=1 / mh_tune_wgrad_taps.
//
// State at the end of round 2 (profiles/r02_microbench_wgrad_taps.txt): bit-exact agreement of the emulator model of the transposing read with the
// MI355X, results within 3e-7 of the tiled kernel's; stand-alone 28.1 us + 8.1 us of split reduction against 34.9 + 4.4 for 3x3 128->128 at 96x320
// (61.6 + 8.3 against 84.4 + 4.1 for four images), but the whole step is 4 % SLOWER with it (1.984 vs 1.899 ms): one workgroup per CU with 144
// accumulators per lane means 64 splits = 37.7 MB of partial sums per layer (17.1 MB tiled) -- ~10 us of the 28 are the partial-sum stores, and the
// reduction doubles --, and the walk itself runs at ~0.7 us per segment where the MFMA work is 0.24 us.  What that is NOT (each tried, see the
// experiment log #37-#41): memory latency (4 segments in flight in the loader waves: no change), LDS bank conflicts of the transposing read
// ([pixel][channels + pad] rows vs the guide's conflict-free 32-byte-row tiles: no change), the loaders' instruction count (250 -> 110 per segment:
// -6 %).  With loads, MFMAs and stores all switched off the barrier-coupled skeleton alone still takes 0.59 us per segment; next: counters.
// LDS images: [16-channel tile][pixel][16 channels] -- 32-byte rows, so the 16 rows one transposing read touches (4 lane groups x 4 pixels) are 512
// contiguous bytes, the layout the guide measures conflict free for ds_read_b64_tr_b16.
constexpr int WT_PW = 34, WT_NSLOT = 8;
constexpr int WT_SUBA = WT_PW * 16, WT_SUBB = 32 * 16;                  // halfs per (patch row slot, channel tile) / per dz channel tile
constexpr int WT_PATCH_HALFS = WT_NSLOT * 2 * WT_SUBA, WT_DZ_HALFS = 8 * WT_SUBB;
constexpr size_t WT_LDS = (size_t)(WT_PATCH_HALFS + 2 * WT_DZ_HALFS) * 2;      // 33 792 B (>= the 16 KB the bias reduction re-uses)

// The workgroup is SPECIALISED: waves 0-3 only read fragments and issue MFMAs, waves 4-7 only load -- WT_NST segments ahead, in registers (they hold no
// accumulators, so they have the room; 5 or 6 stages spill at 512 threads = 256 registers per lane) --, round to bf16 and fill the LDS stage the
// consumers read next; one barrier per segment couples the two halves.  (The first version loaded one segment ahead from the MFMA waves themselves:
// 35 us; the specialisation alone did not change that, see above.)
// FULL: N is a multiple of 128 -- every consumer wave owns four existing column tiles and the MFMA stream carries no conditions.
constexpr int WT_NST = 4;                                              // register stages of the loader waves = segments in flight
template <bool FULL>
__global__ __launch_bounds__(512) void wgrad_taps_kernel(WgradArgs p) {
    HIP_DYNAMIC_SHARED(float, smem)
    unsigned short* const Pa = reinterpret_cast<unsigned short*>(smem);
    unsigned short* const Bz = Pa + WT_PATCH_HALFS;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (scalar: the per-wave conditions below become scalar branches)
    const bool producer = wave >= 4;
    const int d = p.dil;
    int bid = mh_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int tk = bid % p.ktiles; bid /= p.ktiles;              // the k-tiles of a split are neighbours on one XCD: dz comes out of its L2
    const int tn = bid % p.ntiles; bid /= p.ntiles;
    const int split = bid;
    const int k0 = tk * 32, n0 = tn * 128;
    const int Kr = (p.K + 3) & ~3;
    const int Hl = (p.Ho + d - 1) / d, Wl = (p.Wo + d - 1) / d, nsx = (Wl + 31) >> 5;
    const int S = p.B * d * d * nsx * Hl;                         // segments: (b, cy, cx, 32-column strip, lattice row), lattice row fastest
    const int sbeg = split * p.chunk, send = min(S, sbeg + p.chunk);
    const int nseg = send - sbeg;
    const int nsegp = (nseg + WT_NST - 1) / WT_NST * WT_NST;
    const int dbg = p.dbg_plain_store >> 4;                       // timing experiments only (mh_tune_wgrad_taps)
    const bool do_bias = (p.db != nullptr) && tk == 0;
    float4 bsum4[4];                                              // loader lanes: sums of their four dz pieces (one channel group each)
#pragma unroll
    for (int u = 0; u < 4; ++u) bsum4[u] = make_float4(0.f, 0.f, 0.f, 0.f);

    if (producer) {
        // ================================ loader waves ================================
        const int pt = tid - 256;
        const __amdgpu_buffer_rsrc_t rs_in = mh_make_rsrc(p.in, p.in_bytes);
        const __amdgpu_buffer_rsrc_t rs_dz = mh_make_rsrc(p.dz, p.dz_bytes);
        float4 rz[WT_NST][4], rp[WT_NST][4];
        // A wave64 VALU instruction occupies its SIMD for 4 cycles, and the loaders share the SIMDs with the MFMA waves: the first version of this path
        // decoded every segment with integer divisions (~570 VALU instructions per segment: 2300 clk against 576 clk of MFMA work per segment).  Now
        // the segment cursor (b, cy, cx, strip, lattice row) advances with carries in scalar registers, every per-lane quantity is computed once, and
        // an address is one v_add of a lane constant and a scalar base.
        // dz: a loader wave and u pick (channel tile, pixel half); inside, 4 lanes cover the tile's 16 channels of one pixel and 16 pixels follow:
        // the wave's ds_write_b64 is 512 contiguous bytes of the [tile][pixel][16] image
        const int zl = pt & 63, zw = pt >> 6;
        const int zpx = zl >> 2, zcl = zl & 3;
        int prr[4], ppj[4], pconst[4], plds[4];
        bool pk_ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = pt + 256 * u;
            prr[u] = q / (WT_PW * 8);
            const int rem = q - prr[u] * (WT_PW * 8);
            ppj[u] = rem >> 3;
            const int c4 = rem & 7;
            pk_ok[u] = k0 + c4 * 4 < Kr;
            pconst[u] = ((prr[u] * d * p.Wi + ppj[u] * d) * p.in_ld + c4 * 4) * 4;
            plds[u] = (c4 >> 2) * WT_SUBA + ppj[u] * 16 + (c4 & 3) * 4;
        }
        // cursor of the next segment to issue (scalar) + what is constant along its vertical run (same image, sub-lattice, strip; lattice row runs):
        // lane masks of the columns / channels that exist, the byte offsets of lattice row 0.  A single wave issues at most one instruction every ~4
        // cycles whatever its kind, so the loaders' budget per segment is ~140 instructions (576 clk of MFMA work): per segment only the row-dependent
        // scalars are recomputed (measured with everything else switched off: 250 instructions per segment = 1400 clk, the whole kernel's pace).
        int cj = 0, c_ly, c_sx, c_cx, c_cy, c_b;
        {
            const int s = sbeg;
            c_ly = s % Hl; int t = s / Hl;
            c_sx = t % nsx; t /= nsx;
            c_cx = t % d; t /= d;
            c_cy = t % d; c_b = t / d;
        }
        const int zstepY = d * p.Wo * p.dz_ld * 4, pstepY = d * p.Wi * p.in_ld * 4;
        bool zok[4], pok[4];
        int zvo[4];                                               // dz: lane part of the byte offset per load
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cmb = zw * 4 + u, nt = cmb >> 1, px = (cmb & 1) * 16 + zpx;        // (wave, u) -> channel tile 0..7, pixel half
            zvo[u] = (d * px * p.dz_ld + nt * 16 + zcl * 4) * 4;
        }
        int r_hlc = 0, r_zrow0 = 0, r_prow0 = 0;
        auto new_run = [&]() {                                    // (divisions: once per vertical run)
            const int wlc = (p.Wo - c_cx + d - 1) / d;            // columns / rows of this sub-lattice
            r_hlc = (p.Ho - c_cy + d - 1) / d;
            const int wrem = wlc - c_sx * 32;                     // valid pixels of the strip (may exceed 32)
            const int plo = 1 - c_sx * 32, phi = wrem + 1;        // valid patch-column range [plo, phi)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int cmb = zw * 4 + u, nt = cmb >> 1, px = (cmb & 1) * 16 + zpx;
                zok[u] = (n0 + nt * 16 + zcl * 4 < p.N) & (px < wrem) & !(dbg & 4);          // (& not &&: no short-circuit branches on lane conditions)
                pok[u] = pk_ok[u] & (ppj[u] >= plo) & (ppj[u] < phi) & !(dbg & 4);
            }
            r_zrow0 = (((c_b * p.Ho + c_cy) * p.Wo + c_cx + d * c_sx * 32) * p.dz_ld + n0) * 4;
            r_prow0 = (((c_b * p.Hi + c_cy) * p.Wi + c_cx + d * (c_sx * 32 - 1)) * p.in_ld + k0) * 4;
        };
        new_run();
        auto issue = [&](float4 (&z)[4], float4 (&q4)[4]) {      // loads of the cursor's segment (past the end: everything out of range = zeros), then advance
            const bool live = cj < nseg;
            const bool fresh = (cj == 0) || (c_ly == 0);          // first segment of the workgroup or of a vertical run: three patch rows, else one
            const bool zrow_ok = live && c_ly < r_hlc;
            const int zbase = r_zrow0 + c_ly * zstepY;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                z[u] = mh_buf_load4(rs_dz, (zok[u] & zrow_ok) ? zvo[u] + zbase : MH_OOB);
            const int lr0 = fresh ? c_ly - 1 : c_ly + 1;
            int rhi = fresh ? 3 : 1;                              // valid patch-row range [rlo, rhi) of this load
            if (r_hlc - lr0 < rhi) rhi = r_hlc - lr0;
            if (!live) rhi = 0;
            const int rlo = lr0 < 0 ? -lr0 : 0;
            const int pbase = r_prow0 + lr0 * pstepY;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                q4[u] = mh_buf_load4(rs_in, (pok[u] & (prr[u] >= rlo) & (prr[u] < rhi)) ? pconst[u] + pbase : MH_OOB);
            ++cj;
            if (++c_ly == Hl) {
                c_ly = 0;
                if (++c_sx == nsx) {
                    c_sx = 0;
                    if (++c_cx == d) {
                        c_cx = 0;
                        if (++c_cy == d) { c_cy = 0; ++c_b; }
                    }
                }
                new_run();
            }
        };
        int s_ly = sbeg % Hl;                                     // lattice row of the next segment to store
        int s_j = 0;
        auto store = [&](const float4 (&z)[4], const float4 (&q4)[4], int slot0) {      // next segment -> dz buffer s_j & 1, patch slots slot0..
            const int rows = ((s_j == 0) || (s_ly == 0)) ? 3 : 1;
            unsigned short* const Bb = Bz + (s_j & 1) * WT_DZ_HALFS + zpx * 16 + zcl * 4;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int cmb = zw * 4 + u, nt = cmb >> 1, ph = cmb & 1;
                *reinterpret_cast<uint2*>(Bb + nt * WT_SUBB + ph * 256) = make_uint2(mh_pack_bf16(z[u].x, z[u].y), mh_pack_bf16(z[u].z, z[u].w));
                if (do_bias) { bsum4[u].x += z[u].x; bsum4[u].y += z[u].y; bsum4[u].z += z[u].z; bsum4[u].w += z[u].w; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (prr[u] < rows)
                    *reinterpret_cast<uint2*>(Pa + ((slot0 + prr[u]) & (WT_NSLOT - 1)) * (2 * WT_SUBA) + plds[u]) =
                        make_uint2(mh_pack_bf16(q4[u].x, q4[u].y), mh_pack_bf16(q4[u].z, q4[u].w));
            ++s_j;
            if (++s_ly == Hl) s_ly = 0;
        };
        // prologue: segments 0 .. NST-1 in flight, segment 0 into LDS, its stage re-used for segment NST
#pragma unroll
        for (int u = 0; u < WT_NST; ++u) issue(rz[u], rp[u]);
        store(rz[0], rp[0], 0);
        issue(rz[0], rp[0]);
        __syncthreads();
        int base = 0;                                             // ring slot of lattice row ly - 1 of segment i
        // (the walk is padded to a multiple of WT_NST segments on both sides of the workgroup: no condition around the loads, so hipcc's waitcnt pass
        //  keeps the counted vmcnt(N) waits -- with a tail condition it drained to vmcnt(0) at the top of every WT_NST-th segment; loads past the end
        //  are out-of-range = zeros, and the zeros go to a stage nobody reads)
        for (int i0 = 0; i0 < nsegp; i0 += WT_NST) {
#pragma unroll
            for (int u = 0; u < WT_NST; ++u) {
                const int st = (u + 1) % WT_NST;                  // stage of segment i + 1 (static after unrolling)
                const bool run_start = s_ly == 0;                 // segment i + 1 opens a vertical run
                store(rz[st], rp[st], base + 3);                  // slots base+3 .. base+5: never one of the three being read
                issue(rz[st], rp[st]);
                base = (base + (run_start ? 3 : 1)) & (WT_NSLOT - 1);
                __syncthreads();
            }
        }
    } else {
        // ================================ MFMA waves ================================
        const int wr = wave >> 1, wc = wave & 1;
        const int li = lane & 15, lq = lane >> 4;
        const bool kvalid = k0 + wr * 16 < p.K;
        int njw = (p.N - n0 - wc * 64 + 15) >> 4;
        njw = njw < 0 ? 0 : (njw > 4 ? 4 : njw);
        const bool active = kvalid && njw > 0;
        f32x4 acc[9][4];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // lane-constant parts of the transposing-read addresses: pixel rows of the two reads, 4-channel piece
        const int lrow = (4 * lq + (li >> 2)) * 16 + 4 * (li & 3);       // read 0: pixel rows 4 lq + (li >> 2); read 1: 16 rows (256 halfs) further
        __syncthreads();                                          // (the loaders' prologue)
        int base = 0;
        int n_ly = sbeg % Hl;                                     // lattice row of segment i + 1
        if (++n_ly == Hl) n_ly = 0;
        for (int i = 0; i < nsegp; ++i) {
            if (active && i < nseg && !(dbg & 1)) {
                // one segment: 4 B fragments (this wave's output-channel tiles), then per tap one A fragment and 4 MFMAs; without FULL the column
                // tiles past N are skipped (wave-uniform scalar branches)
                const unsigned short* const Bb = Bz + (i & 1) * WT_DZ_HALFS + wc * 4 * WT_SUBB + lrow;
                u32x4 bf[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (FULL || j < njw) {
                        const uint2 b0 = mh_lds_read_tr16(Bb + j * WT_SUBB), b1 = mh_lds_read_tr16(Bb + j * WT_SUBB + 256);
                        bf[j] = (u32x4){b0.x, b0.y, b1.x, b1.y};
                    }
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int ky = t / 3, kx = t - ky * 3;
                    const unsigned short* const Ab = Pa + (((base + ky) & (WT_NSLOT - 1)) * 2 + wr) * WT_SUBA + kx * 16 + lrow;
                    const uint2 a0 = mh_lds_read_tr16(Ab), a1 = mh_lds_read_tr16(Ab + 256);
                    const u32x4 af = (u32x4){a0.x, a0.y, a1.x, a1.y};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (FULL || j < njw) acc[t][j] = mh_mfma_bf16(af, bf[j], acc[t][j]);
                }
            }
            base = (base + (n_ly == 0 ? 3 : 1)) & (WT_NSLOT - 1);
            if (++n_ly == Hl) n_ly = 0;
            __syncthreads();
        }
        float* const dwb = p.ws ? p.ws + (int64_t)split * ((int64_t)9 * p.K * p.N) : p.dw;
        const bool plain = (p.ws != nullptr) || p.dbg_plain_store;
        if (kvalid && !(dbg & 8)) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = k0 + wr * 16 + lq * 4 + r;
                    if (k >= p.K) continue;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int n = n0 + wc * 64 + j * 16 + li;
                        if ((FULL || j < njw) && n < p.N) {
                            float* dst = dwb + ((int64_t)t * p.K + k) * p.N + n;
                            if (plain) *dst = acc[t][j][r]; else atomicAdd(dst, acc[t][j][r]);
                        }
                    }
                }
        }
    }
    if (do_bias) {                                                // uniform per workgroup.  32 loader lanes per 4-channel group -> LDS -> one atomic per output channel
        float* red = smem;                                        // [32][128]; every wave is past the last barrier of the walk
        if (producer) {
            const int pt = tid - 256, zl = pt & 63, zw = pt >> 6;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int cmb = zw * 4 + u;
                *reinterpret_cast<float4*>(red + ((cmb & 1) * 16 + (zl >> 2)) * 128 + (cmb >> 1) * 16 + (zl & 3) * 4) = bsum4[u];
            }
        }
        __syncthreads();
        if (tid < 128 && n0 + tid < p.N) {
            float t = 0.f;
#pragma unroll 8
            for (int r = 0; r < 32; ++r) t += red[r * 128 + tid];
            atomicAdd(p.db + n0 + tid, t);
        }
    }
}

static std::atomic<int> g_wgrad_taps_flags{0};
static std::atomic<int> g_wgrad_taps_launches{0};
static std::atomic<int> g_wgrad_taps{-1};                          // -1: environment (MH_WGRAD_TAPS, default off), 0 / 1: mh_tune_wgrad_taps
static bool wgrad_taps_ok(const WgradArgs& a) {
    static const int env_on = []() { const char* e = getenv("MH_WGRAD_TAPS"); return e ? atoi(e) : 0; }();
    // (16384: the step A/B of round 2 ran with 4096, which also sent the 48x160 layers here -- 240 segments = 4 per workgroup against ~15 us of
    //  per-workgroup prologue + 147 KB of partial sums.  With the floor at 16384 the step is still 3.5 % slower: 1.959 vs 1.893 ms.)
    static const int env_minm = []() { const char* e = getenv("MH_WGRAD_TAPS_MINM"); return e ? atoi(e) : 16384; }();
    const int t = g_wgrad_taps.load(std::memory_order_relaxed);
    if (!(t >= 0 ? t : env_on)) return false;
    const int min_m = (g_wgrad_taps_flags.load(std::memory_order_relaxed) & 0x100) ? 0 : env_minm;       // mh_tune_wgrad_taps(1 + 16 * 0x100): every size (tests)
    return a.bf16 && !a.flat && a.taps == 9 && a.kh == 3 && a.kw == 3 && a.stride == 1 && a.pad_t == a.dil && a.pad_l == a.dil &&
           a.Hi == a.Ho && a.Wi == a.Wo && a.vecA && a.vecB && a.K >= 32 && a.N > 64 && a.M > min_m;     // (N <= 64 would idle the second wave column: tiled kernels)
}
static int launch_wgrad_taps(WgradArgs& a, hipStream_t s) {
    a.ktiles = mh_cdiv(a.K, 32);
    a.ntiles = mh_cdiv(a.N, 128);
    const int d = a.dil;
    const int Hl = mh_cdiv(a.Ho, d), nsx = mh_cdiv(mh_cdiv(a.Wo, d), 32);
    const int S = a.B * d * d * nsx * Hl;
    const int base = a.ktiles * a.ntiles;
    const int target = g_wgrad_target_wgs > 0 ? g_wgrad_target_wgs : 256;         // one workgroup per CU: the partial sums are 4 bytes x every accumulator in flight
    int splits = a.forced_splits > 0 ? a.forced_splits : mh_cdiv(target, base);
    int maxs = S / 8;                                                            // >= 8 segments per workgroup
    if (maxs > 192) maxs = 192;
    if (splits > maxs) splits = maxs;
    if (splits < 1) splits = 1;
    const int chunk = mh_cdiv(S, splits);
    a.splits = mh_cdiv(S, chunk);
    a.chunk = chunk;
    if (a.query) return 0;
    if (t_capture) { t_capture->cfg = -1; return 0; }            // never part of a grouped launch
    a.dbg_plain_store |= (g_wgrad_taps_flags.load(std::memory_order_relaxed) & 0xff) << 4;
    g_wgrad_taps_launches.fetch_add(1, std::memory_order_relaxed);
    mh_note_kernel("wgrad_taps_kernel K=%d N=%d dil=%d segments %d splits %d grid %d", a.K, a.N, d, S, a.splits, base * a.splits);
    if (a.N % 128 == 0) hipLaunchKernelGGL(wgrad_taps_kernel<true>, dim3(base * a.splits), dim3(512), WT_LDS, s, a);
    else hipLaunchKernelGGL(wgrad_taps_kernel<false>, dim3(base * a.splits), dim3(512), WT_LDS, s, a);
    return mh_check_launch("wgrad_taps");
}

static int wgrad_dispatch(WgradArgs& a, hipStream_t s) {
    const int K = a.flat ? a.taps * 4 : a.K, N = a.N;        // flat: the dW tile rows are (tap, channel) pairs
    const bool all = a.M < 0;
    int rc = 0;
    if (!all && wgrad_taps_ok(a)) return launch_wgrad_taps(a, s);
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
    static const int w8_on = []() { const char* e = getenv("MH_WGRAD_W8"); return e ? atoi(e) : 1; }();       // A/B hook
    MH_WG(K > 64 && N > 64 && a.M > 4096 && a.bf16 && w8_on && g_wgrad_w8 != 3, 4, 2, 2, 4, 32)
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

// on >= 16: 1 + 16 * bits -- timing experiments (WRONG results): bit 2 no global loads, bit 3 no partial-sum stores, bit 0: skip the MFMA walk
// (conditions inside the MFMA stream were tried and are useless: they make hipcc shuffle the 144 accumulators at every merge, +14 us)
// (+ 16 * 0x100: no lower bound on the reduction pixels -- tests).  Returns the number of all-taps launches since the previous call.
extern "C" int mh_tune_wgrad_taps(int on) {
    g_wgrad_taps = on < 0 ? -1 : (on ? 1 : 0);
    g_wgrad_taps_flags = on >= 16 ? on >> 4 : 0;
    return g_wgrad_taps_launches.exchange(0, std::memory_order_relaxed);
}

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
    static const int flat_on = []() { const char* e = getenv("MH_WGRAD_FLAT"); return e ? atoi(e) : 1; }();      // A/B hook
    // measured: the 7x7 image layer of DispNet 210 -> ~60 us per tower, but MADNet's 3x3 one is 1 % slower flat (9 taps only
    // re-read dz 9x from L2, and the flat tables cost more than they save) -> many-tap layers only; MH_WGRAD_FLAT=2 forces it
    a.flat = (flat_on && a.bf16 && a.vecA && a.vecB && d->K <= 4 && (a.taps >= 16 || (flat_on == 2 && a.taps > 1))) ? 1 : 0;
    a.ws = ws; a.forced_splits = forced_splits; a.query = query;
    {
        const int64_t inb = (((int64_t)d->B * d->Hi * d->Wi - 1) * d->in_ld + (int64_t)((d->K + 3) / 4) * 4) * 4;
        const int64_t dzb = (((int64_t)a.M - 1) * dout_ld + d->N) * 4;
        MH_REQUIRE(inb < (1ll << 31) - 64 && dzb < (1ll << 31) - 64, MH_ERR_UNSUPPORTED, "mh_conv2d_wgrad: tensors must be < 2 GiB");
        a.in_bytes = (unsigned)inb; a.dz_bytes = (unsigned)dzb;
    }
    t_capture = cap;
    const int rc = wgrad_n1_ok(a) ? launch_wgrad_n1(a, (hipStream_t)stream) : wgrad_dispatch(a, (hipStream_t)stream);
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
    static const int group_on = []() { const char* e = getenv("MH_WGRAD_GROUP"); return e ? atoi(e) : 1; }();      // A/B hook
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
        static const int env_max_m = []() { const char* e = getenv("MH_WGRAD_GROUP_MAXM"); return e ? atoi(e) : 0; }();
        const int max_m = env_max_m > 0 ? env_max_m : (batch_max_m > 0 ? batch_max_m : 4096);
        const bool narrow = (c.cfg == 8 || c.cfg == 10 || c.cfg == 11);
        if (!group_on || c.cfg < 0 || narrow || a.M > max_m || c.lds > (size_t)MH_WG_GROUP_LDS || c.nblocks <= 0) {
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
    int lo = 0, hi = nseg - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (segs[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const mh_wgrad_seg sg = segs[lo];
    const int e0 = ((int)blockIdx.x - sg.blk0) * 1024 + threadIdx.x * 4;
    if (e0 >= sg.size) return;
    if ((sg.size & 3) == 0) {       // every split slice 16-byte aligned (ws is): vector path
        const float* src = sg.ws + e0;
        // 4 independent accumulators = 4 loads in flight per lane (a single dependent chain ran at 2.6 TB/s)
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f), t1 = t, t2 = t, t3 = t;
        int s = 0;
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
