// conv_direct.hip -- EXPERIMENT, OFF BY DEFAULT (measured slower, see the end of this header) -- LDS-free, barrier-free bf16-MFMA kernel
// for the SMALL stride-1 layers (the estimator chains of the coarse
// pyramid levels and the stride-1 pyramid layers: Nets/MadNet.py:73-120,173-249 via Nets/sharedLayers.py:54-63) and their input
// gradients.
//
// Why: at batch 1 these layers have 120 .. 7680 output pixels.  The tiled implicit-GEMM kernel (conv.hip) spends 10-17 us on each of
// them whatever their size (profiles/r02_bench_eager_serial_kernel_stats_mixed.csv): its 32x32 tiles run 16 waves per workgroup
// that are ISSUE bound -- ~200 instructions per wave and 64-k step (address arithmetic, fp32->bf16 conversion, swizzled LDS stores,
// barrier, LDS reads) for 2 MFMAs -- and a deeper prefetch made them slower (profiles/r02_experiments.txt #4).  Here every wave owns a
// (16*MT pixels) x (16*NT channels) output tile and feeds its MFMAs straight from global memory:
//   * A operand (v_mfma_f32_16x16x32_bf16: lane (i, q) holds k = 8q .. 8q+7 of row i) = 8 consecutive channels of ONE pixel
//     = two 16-byte loads (NHWC), converted to bf16 in registers;
//   * B operand = 8 consecutive k of ONE output column: contiguous in memory when the weights are stored k-fastest -- the input
//     gradient reads HWIO weights [tap][cin][cout] with k = cout (contiguous as stored); the forward pass needs [tap][cout][cin],
//     i.e. a transposed copy of the filter bank (`wt`, written once per step by mh_transpose_weights);
//   * no LDS, no barrier, no inter-wave dependency: ~25 instructions per 32-k step, every load independent (the waves of a CU
//     re-read each other's operands from L1 / L2: at most 2.6 MB of activations and 0.6 MB of weights per layer).
// Precision: bf16 operands (code 1) or split-bf16 (code 2: hi/lo of both operands built in registers, 3 MFMAs per product).
// The epilogue is the one of conv_igemm_kernel (bias + leaky, or accumulate + leaky-gradient mask with a channel range).
//
// MEASURED on the MI355X (profiles/r02_experiments.txt #12, in situ, serial rocprofv3 trace): 22-26 us per launch on average against
// 10-13 us for the tiled kernel it was meant to replace; whole step 2.46 -> 3.05-3.14 ms (mixed), 2.24 -> 2.99 ms (bf16).  A four-stage
// software pipeline of the operand loads changed nothing, so it is not the load round trip: every wave re-reads operands its
// neighbours also read, 16 bytes per lane from 16 different cache lines per instruction -- ~96 line accesses per 32-k step and wave for
// 2-6 MFMAs -- and the L1 (one line per clock) becomes the bound.  Sharing operands through LDS is what the tiled kernel is FOR.
// The kernel stays as an opt-in (mh_tune_conv_direct(2) / MH_CONV_DIRECT=2: parity-tested) and as the record of the experiment.
#include "conv_args.h"
#include <stdlib.h>
#include <atomic>

namespace {

struct DirectGeo { int mtiles, ntiles, kchunks, nwaves; };

// 8 consecutive fp32 (two 16-byte loads) -> one MFMA operand fragment (bf16, or hi + lo planes for split-bf16).  K is a multiple
// of 8, so a lane's group of 8 channels is either entirely inside the channel range or entirely outside (out-of-range offset = zeros).
template <bool X3>
__device__ __forceinline__ void to_frag(const float4& v0, const float4& v1, u32x4& hi, u32x4& lo) {
    if constexpr (X3) {
        unsigned h0, h1, h2, h3, l0, l1, l2, l3;
        mh_split_bf16x2(v0.x, v0.y, h0, l0); mh_split_bf16x2(v0.z, v0.w, h1, l1);
        mh_split_bf16x2(v1.x, v1.y, h2, l2); mh_split_bf16x2(v1.z, v1.w, h3, l3);
        hi = (u32x4){h0, h1, h2, h3}; lo = (u32x4){l0, l1, l2, l3};
    } else {
        hi = (u32x4){mh_pack_bf16(v0.x, v0.y), mh_pack_bf16(v0.z, v0.w), mh_pack_bf16(v1.x, v1.y), mh_pack_bf16(v1.z, v1.w)};
    }
}

// DGRAD = false: y[p][n] = sum_{tap,k} x[p + tap][k] * wt[tap][n][k]        (wt = transposed filter bank, K = Cin,  N = Cout)
// DGRAD = true : dx[p][n] = sum_{tap,k} dz[p - tap][k] * w[tap][n][k]        (w  = HWIO as stored,        K = Cout, N = Cin)
template <int MT, int NT, bool DGRAD, bool X3>
__global__ __launch_bounds__(256) void conv_direct_kernel(ConvArgs p, DirectGeo g) {
    const int lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
    const int wv = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wv >= g.nwaves) return;                               // (no barrier in this kernel: whole waves may leave)
    const int tn = wv % g.ntiles, tm = wv / g.ntiles;
    const int m0 = tm * (16 * MT), n0 = tn * (16 * NT);
    const __amdgpu_buffer_rsrc_t rs_in = mh_make_rsrc(p.in, p.in_bytes);
    const __amdgpu_buffer_rsrc_t rs_w = mh_make_rsrc(p.w, p.w_bytes);

    // A rows of this lane: pixel m0 + 16*i + li
    int a_y[MT], a_x[MT], a_img[MT];
    bool a_ok[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = m0 + 16 * i + li;
        a_ok[i] = m < p.M;
        const int mm = a_ok[i] ? m : 0;
        const int ox = mm % p.Wo, t2 = mm / p.Wo;
        a_y[i] = t2 % p.Ho; a_x[i] = ox; a_img[i] = (t2 / p.Ho) * p.Hi * p.Wi;
    }
    // B columns of this lane: channel n0 + 16*j + li ; byte offset of (tap 0, k 0) of that column
    int b_off[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n0 + 16 * j + li;
        b_off[j] = n < p.N ? n * p.K * 4 : MH_OOB;
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // The K walk = taps x 32-channel chunks, flattened, with the operand loads of step s + D issued while step s is multiplied
    // (D = 4 register stages of raw fp32): without it every step waited for its own loads -- a dependent L2 round trip per 32 k,
    // 22-26 us per layer (profiles/r02_experiments.txt #12).  Loads past the walk get out-of-range offsets (zeros, no traffic)
    // and their MFMAs add nothing, so the loop needs no tail handling.
    constexpr int D = 4;
    const int tap_stride = p.N * p.K * 4;                       // bytes per tap of the [tap][N][K] weight operand
    const int nsteps = p.taps * g.kchunks;
    float4 ra[D][MT][2], rb[D][NT][2];
    int i_tap = 0, i_c = 0;                                     // (tap, chunk) of the next step to ISSUE
    int a_base[MT];
    auto new_tap = [&]() {
        const int ky = i_tap / p.kw, kx = i_tap - ky * p.kw;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int iy = DGRAD ? a_y[i] + p.pad_t - ky * p.dil : a_y[i] * p.stride - p.pad_t + ky * p.dil;
            const int ix = DGRAD ? a_x[i] + p.pad_l - kx * p.dil : a_x[i] * p.stride - p.pad_l + kx * p.dil;
            const bool ok = i_tap < p.taps && a_ok[i] && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            a_base[i] = ok ? ((a_img[i] + iy * p.Wi + ix) * p.in_ld + lq * 8) * 4 : MH_OOB;
        }
    };
    new_tap();
    auto issue = [&](float4 (&a)[MT][2], float4 (&b)[NT][2]) {
        const bool kin = i_tap < p.taps && i_c * 32 + lq * 8 < p.K;          // this lane's 8 channels exist (K % 8 == 0)
        const int wofs = i_tap * tap_stride + lq * 32 + i_c * 128;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int off = (a_base[i] != MH_OOB && kin) ? a_base[i] + i_c * 128 : MH_OOB;
            a[i][0] = mh_buf_load4(rs_in, off);
            a[i][1] = mh_buf_load4(rs_in, off == MH_OOB ? MH_OOB : off + 16);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int off = (b_off[j] != MH_OOB && kin) ? b_off[j] + wofs : MH_OOB;
            b[j][0] = mh_buf_load4(rs_w, off);
            b[j][1] = mh_buf_load4(rs_w, off == MH_OOB ? MH_OOB : off + 16);
        }
        if (++i_c == g.kchunks) { i_c = 0; ++i_tap; new_tap(); }
    };
    auto consume = [&](const float4 (&a)[MT][2], const float4 (&b)[NT][2]) {
        u32x4 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) to_frag<X3>(a[i][0], a[i][1], ah[i], al[i]);
#pragma unroll
        for (int j = 0; j < NT; ++j) to_frag<X3>(b[j][0], b[j][1], bh[j], bl[j]);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if constexpr (X3) {
                    acc[i][j] = mh_mfma_bf16(al[i], bh[j], acc[i][j]);
                    acc[i][j] = mh_mfma_bf16(ah[i], bl[j], acc[i][j]);
                }
                acc[i][j] = mh_mfma_bf16(ah[i], bh[j], acc[i][j]);
            }
    };
#pragma unroll
    for (int d = 0; d < D; ++d) issue(ra[d], rb[d]);
    for (int s0 = 0; s0 < nsteps; s0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            float4 ca[MT][2], cb[NT][2];
#pragma unroll
            for (int i = 0; i < MT; ++i) { ca[i][0] = ra[d][i][0]; ca[i][1] = ra[d][i][1]; }
#pragma unroll
            for (int j = 0; j < NT; ++j) { cb[j][0] = rb[d][j][0]; cb[j][1] = rb[d][j][1]; }
            issue(ra[d], rb[d]);                                  // step s0 + d + D into the stage just drained
            consume(ca, cb);                                      // step s0 + d
        }
    }
    // epilogue: acc[i][j][r] = row (pixel) m0 + 16 i + 4 lq + r, column n0 + 16 j + li
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 16 * i + 4 * lq + r;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = n0 + 16 * j + li;
                if (n >= p.N) continue;
                float v = acc[i][j][r];
                if (p.bias) v += p.bias[n];
                if (p.alpha != 1.0f) v = v > 0.f ? v : p.alpha * v;
                float* dst = p.out + (int64_t)m * p.out_ld + n;
                if (p.accumulate) v += *dst;
                if (p.mask_ref && n >= p.mask_c0 && n < p.mask_c1) v *= (p.mask_ref[(int64_t)m * p.mask_ld + n] > 0.f) ? 1.0f : p.mask_alpha;
                *dst = v;
            }
        }
}

// 0 = off (default: measured slower), 1 = size heuristic, 2 = forced whenever eligible (tests); process-wide tuning hook
std::atomic<int> g_direct_mode{-2};
std::atomic<int> g_direct_launches{0};
int direct_mode() {
    int m = g_direct_mode.load(std::memory_order_relaxed);
    if (m == -2) { const char* e = getenv("MH_CONV_DIRECT"); m = e ? atoi(e) : 0; g_direct_mode.store(m, std::memory_order_relaxed); }
    return m;
}

template <int MT, int NT, bool DGRAD, bool X3>
int launch_direct(ConvArgs& a, hipStream_t s) {
    DirectGeo g;
    g.mtiles = mh_cdiv(a.M, 16 * MT); g.ntiles = mh_cdiv(a.N, 16 * NT);
    g.kchunks = mh_cdiv(a.K, 32); g.nwaves = g.mtiles * g.ntiles;
    ++g_direct_launches;
    mh_note_kernel("conv_direct_kernel<%d,%d,%s,%s> wave tile %dx%d waves %d", MT, NT, DGRAD ? "dgrad" : "fwd", X3 ? "bf16x3" : "bf16", 16 * MT, 16 * NT, g.nwaves);
    hipLaunchKernelGGL((conv_direct_kernel<MT, NT, DGRAD, X3>), dim3(mh_cdiv(g.nwaves, 4)), dim3(256), 0, s, a, g);
    return mh_check_launch("conv_direct");
}

template <bool DGRAD, bool X3>
int launch_direct_tile(ConvArgs& a, hipStream_t s) {
    // wave tile: 32 x 32 when that still gives every SIMD a wave (>= 1024 tiles), else 16 x 32 (more, smaller waves)
    const int64_t big = (int64_t)mh_cdiv(a.M, 32) * mh_cdiv(a.N, 32);
    if (big >= 1024) return launch_direct<2, 2, DGRAD, X3>(a, s);
    return launch_direct<1, 2, DGRAD, X3>(a, s);
}

}  // namespace

extern "C" int mh_tune_conv_direct(int mode) {
    g_direct_mode = mode < 0 ? 0 : mode;
    return g_direct_launches.exchange(0);
}

// Eligible: stride 1, same spatial size in and out, K a multiple of 8 with 16-byte rows, bf16 or split-bf16 arithmetic, and a
// k-fastest weight operand: the input gradient always has one (HWIO as stored); the forward pass when the caller passed `wt`.
bool mh_conv_direct_ok(const ConvArgs& a, const float* wt) {
    const int mode = direct_mode();
    if (mode == 0) return false;
    if (!(a.bf16 || a.x3)) return false;
    if (a.mode == 1 && (a.stride != 1 || a.Hi != a.Ho || a.Wi != a.Wo || a.ncls != 0)) return false;      // (the forward pass takes any stride)
    if ((a.K & 7) != 0 || !a.vecA || (a.in_ld & 3) != 0) return false;
    if (a.mode == 0 && !wt) return false;
    if (a.mode == 1 && a.x3) return false;                      // gradients never run split-bf16
    if (a.N < 16) return false;
    if (mode == 2) return true;
    // heuristic: the layers the tiled kernel runs on its latency-bound 32x32 / 32x64 tiles
    return (int64_t)a.M <= 8192 && (int64_t)a.taps * a.K >= 256;
}

int mh_conv_direct_launch(ConvArgs& a, const float* wt, hipStream_t s) {
    if (a.mode == 0) {
        a.w = wt;                                               // [tap][N][K], same byte count as the HWIO bank
        return a.x3 ? launch_direct_tile<false, true>(a, s) : launch_direct_tile<false, false>(a, s);
    }
    return launch_direct_tile<true, false>(a, s);
}

// ---- transposed filter banks for the forward pass: wt[tap][n][k] = w[tap][k][n], every layer of a table in one launch -----------
namespace {
struct TransposeSeg { const float* src; float* dst; int taps, K, N, blk0; };
__global__ __launch_bounds__(256) void transpose_weights_kernel(const TransposeSeg* __restrict__ segs, int nseg) {
    int lo = 0, hi = nseg - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (segs[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const TransposeSeg sg = segs[lo];
    const int e = ((int)blockIdx.x - sg.blk0) * 256 + threadIdx.x;       // destination element: (tap, n, k), k fastest
    const int per_tap = sg.K * sg.N;
    if (e >= sg.taps * per_tap) return;
    const int tap = e / per_tap, r = e - tap * per_tap;
    const int n = r / sg.K, k = r - n * sg.K;
    sg.dst[e] = sg.src[tap * per_tap + k * sg.N + n];
}
}  // namespace

extern "C" int mh_transpose_weights(const mh_transpose_seg* segs_device, int32_t nseg, int32_t nblocks, void* stream) {
    MH_REQUIRE(segs_device && nseg > 0 && nblocks > 0, MH_ERR_ARG, "mh_transpose_weights: empty segment table");
    static_assert(sizeof(TransposeSeg) == sizeof(mh_transpose_seg), "segment layout");
    hipLaunchKernelGGL(transpose_weights_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const TransposeSeg*>(segs_device), nseg);
    return mh_check_launch("transpose_weights");
}
