// conv_rows.hip -- row-streaming 3x3 conv kernel for the thin layers at 1/2 resolution: MADNet pyramid conv1 (3 -> 16, stride 2, forward) and conv2
// (16 -> 16 at 192x640 x 2 towers, forward and input gradient); Nets/MadNet.py:56-66, sharedLayers.py:54-92.
//
// Those layers are 0.3-1.1 GFLOP over 31-39 MB of activations: 6-8 us of HBM time, no MFMA time to speak of -- and 22-31 us on the tiled kernels, whose
// workgroups re-gather every input pixel nine times through L2 and pay a stage / barrier / epilogue round trip per 128-pixel tile.
// Here nothing is staged and nothing is shared:
//   * a WAVE owns a 32-pixel column strip and walks down R output rows.  Roles are swapped against the other kernels: the FILTER BANK is the A
//     operand (v_mfma_f32_32x32x16_bf16: 32 output channels x 16 input channels of one tap), converted once per wave and kept in registers for
//     all 9 taps (36 VGPRs, 72 with the lo plane of the split-bf16 forward mode); the PIXELS are the B operand -- lane l holds 8 channels of pixel
//     l & 31, i.e. 32 contiguous bytes of the NHWC row, loaded straight from global memory by two 16-byte buffer loads (out-of-range columns and
//     rows read zeros: no border code);
//   * stride 1: an input row is loaded ONCE per wave as three column-shifted fragments (the shifts hit L1) and feeds the three output rows it
//     belongs to: three accumulators rotate through the walk (static register indices: the row loop is unrolled), the loads run two rows ahead of
//     the 9 (27) MFMAs.  Stride 2: two new input rows per output row, the third stays in registers as the next row's first;
//   * with the swap the accumulator layout is pixel-per-lane, 4 CONSECUTIVE CHANNELS per register quad: bias / leaky / accumulate / mask epilogue on
//     float4s, 16-byte stores, bf16 shadow store for the streamed filter gradient -- no LDS transpose;
//   * every load and store of the walk is UNCONDITIONAL (out-of-range offset into a buffer descriptor instead of a branch): a memory access behind a
//     branch costs hipcc its count of what is in flight -- loaded registers become phi copies behind s_waitcnt vmcnt(0), a join takes the smaller
//     store count -- and the walk then waits for its own stores at every row (22 -> 15-18 us, profiles/r03_experiments.txt #13).
// LDS is used once, to hand the fp32 filter bank to the waves in fragment order.  Same arithmetic as the other kernels of the same precision code:
// operands rounded to bf16 (precision 1) or split into hi + lo bf16 with 3 MFMAs per product (precision 2, forward), fp32 accumulation; another
// summation order.
#include "conv_args.h"
#include <stdlib.h>
#include <atomic>

namespace {

template <bool X3>
__device__ __forceinline__ void rows_cvt(const float4& a, const float4& b, u32x4& hi, u32x4& lo) {
    if (X3) {
        unsigned h0, h1, h2, h3, l0, l1, l2, l3;
        mh_split_bf16x2(a.x, a.y, h0, l0); mh_split_bf16x2(a.z, a.w, h1, l1);
        mh_split_bf16x2(b.x, b.y, h2, l2); mh_split_bf16x2(b.z, b.w, h3, l3);
        hi = (u32x4){h0, h1, h2, h3}; lo = (u32x4){l0, l1, l2, l3};
    } else {
        hi = (u32x4){mh_pack_bf16(a.x, a.y), mh_pack_bf16(a.z, a.w), mh_pack_bf16(b.x, b.y), mh_pack_bf16(b.z, b.w)};
    }
}

// p.mode 0: out = conv(in, w) ; p.mode 1: out = conv2d_backprop_input (stride 1: the same walk with the taps flipped and the bank transposed).
// S = 2: the forward pass of the down-sampling layers (conv1 3 -> 16, conv3 16 -> 32): an output row takes input rows 2y - pad .. + 2, two new ones
// per step (the third is the next step's first and stays in registers), lane pixels two apart.
// EPI: the epilogue reads the old output (accumulate) and / or a leaky-gradient mask; those loads are issued BEFORE the next row's prefetch so that
// waiting for them does not drain it.
// NQ: channel quads per lane that exist (2: <= 16 output channels, 4: <= 32) -- compile time, so that no all-out-of-range store is issued.
// SH: the input AND the leaky mask come from bf16 shadows (input gradient whose predecessor wrote the shadow of dz, activation shadow from the
// forward pass): one 16-byte load per fragment instead of two + a conversion, 8 mask bytes instead of 16.
template <int S, bool X3, bool EPI, int NQ, bool SH = false>
__global__ __launch_bounds__(256) void conv_rows_kernel(ConvArgs p, int R, int strips, int rblocks, unsigned mulK, unsigned mulN) {
    // filter bank -> LDS in A-fragment order, zero padded: sA[tap][output channel m < 32][input channel kk < 16].  forward: HWIO w[t][kk][m] ;
    // input gradient: taps flipped and the bank transposed, w[8 - t][m][kk] (m = input channel of the forward conv)
    __shared__ __attribute__((aligned(16))) float sA[9 * 32 * 16];
    __shared__ __attribute__((aligned(16))) float sBias[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int unit = blockIdx.x * 4 + wave;                    // (image, row block, strip), strips fastest
    const bool active = unit < p.B * rblocks * strips;          // (a spare wave of the last workgroup: its loads read zeros, it leaves after the barrier)
    const int strip = unit % strips;
    const int t2 = unit / strips;
    const int rb = t2 % rblocks, b = t2 / rblocks;
    const int x0 = strip * 32, r0 = rb * R;
    const int r1 = min(r0 + R, p.Ho);
    const int lp = lane & 31, lh = lane >> 5;

    const __amdgpu_buffer_rsrc_t rs_in = SH ? mh_make_rsrc(p.in_shadow, p.in_shadow_bytes) : mh_make_rsrc(p.in, p.in_bytes);
    const int sh_ld = (p.K + 31) & ~31;              // (SH) pixel stride of the input's shadow, halfs
    // byte offset of (row, pixel S * (x0 + lp) + dx - pad, channel 8 * lh) ; K <= 8: the upper half-wave has no channels
    const bool kok = 8 * lh < p.K;
    auto row_off = [&](int r, int dx) -> int {
        const int x = S * (x0 + lp) + dx - p.pad_l;
        const bool ok = active && kok && (unsigned)r < (unsigned)p.Hi && (unsigned)x < (unsigned)p.Wi;
        if (SH) return ok ? (((b * p.Hi + r) * p.Wi + x) * sh_ld + 8 * lh) * 2 : MH_OOB;
        return ok ? (((b * p.Hi + r) * p.Wi + x) * p.in_ld + 8 * lh) * 4 : MH_OOB;
    };
    // K % 4 != 0 (the image layer: 3 channels in a 4-float pixel): whatever sits in the padding lanes of the last group must not meet the MFMA
    const int ktail = p.K & 3;
    auto load_row = [&](float4 (&dst)[3][2], int r) {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int o = row_off(r, dx);
            dst[dx][0] = mh_buf_load4(rs_in, o);
            if (!SH) dst[dx][1] = mh_buf_load4(rs_in, (o == MH_OOB || 8 * lh + 4 >= p.K) ? MH_OOB : o + 16);
        }
    };
    auto cvt_row = [&](float4 (&src)[3][2], u32x4* Bh, u32x4* Bl) {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            if (SH) { Bh[dx] = __builtin_bit_cast(u32x4, src[dx][0]); Bl[dx] = (u32x4){0u, 0u, 0u, 0u}; continue; }      // 8 bf16 as they lie in the shadow
            if (ktail) {            // (wave-uniform; K < 8 here: only the first group carries channels)
                float4& v = src[dx][0];
                if (ktail < 2) v.y = 0.f;
                if (ktail < 3) v.z = 0.f;
                v.w = 0.f;
            }
            u32x4 lo = {0u, 0u, 0u, 0u};
            rows_cvt<X3>(src[dx][0], src[dx][1], Bh[dx], lo);
            Bl[dx] = lo;
        }
    };
    float4 raw[3][2];
    // the first rows are on their way while the filter bank is staged
    float4 rawb[3][2];
    if constexpr (S == 1) { load_row(raw, r0 - 1); load_row(rawb, r0); }
    else load_row(raw, 2 * r0 - p.pad_t);

    for (int i = tid; i < 9 * 32 * 16 / 4; i += 256) reinterpret_cast<float4*>(sA)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    // coalesced read of the bank as it lies in memory, scattered into fragment order (mode 0: source index (t*K + kk)*N + m ; mode 1: source tap
    // 8 - t, index (t'*N + m)*K + kk): 9 independent loads per thread, one wait
    const int nwt = 9 * p.K * p.N;                    // <= 4608
    {
        const __amdgpu_buffer_rsrc_t rs_w = mh_make_rsrc(p.w, p.w_bytes);
        float wv[18];
#pragma unroll
        for (int u = 0; u < 18; ++u) { const int j = tid + 256 * u; wv[u] = mh_buf_load1(rs_w, j < nwt ? j * 4 : MH_OOB); }
#pragma unroll
        for (int u = 0; u < 18; ++u) {
            const int j = tid + 256 * u;
            // j / K, j / N by multiplication (mul = ceil(2^32 / d) from the host: exact for j < 2^16)
            int t, m, kk;
            if (p.mode == 0) { const int q = (int)__umulhi((unsigned)j, mulN); m = j - q * p.N; t = (int)__umulhi((unsigned)q, mulK); kk = q - t * p.K; }
            else { const int q = (int)__umulhi((unsigned)j, mulK); kk = j - q * p.K; const int tt = (int)__umulhi((unsigned)q, mulN); m = q - tt * p.N; t = 8 - tt; }
            if (j < nwt) sA[(t * 32 + m) * 16 + kk] = wv[u];
        }
    }
    if (tid < 32) sBias[tid] = (p.bias && tid < p.N) ? p.bias[tid] : 0.f;
    __syncthreads();
    if (!active) return;

    // ---- A fragments: row m = output channel (lane & 31), k = 8 * (lane >> 5) + j = input channel of the tap: 32 contiguous bytes of sA -------
    u32x4 Ah[9], Al[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const float4 v0 = *reinterpret_cast<const float4*>(sA + (t * 32 + lp) * 16 + 8 * lh);
        const float4 v1 = *reinterpret_cast<const float4*>(sA + (t * 32 + lp) * 16 + 8 * lh + 4);
        u32x4 lo = {0u, 0u, 0u, 0u};
        rows_cvt<X3>(v0, v1, Ah[t], lo);
        Al[t] = lo;
    }

    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto mac = [&](f32x16& acc, int dy, const u32x4* Bh, const u32x4* Bl) {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int t = dy * 3 + dx;
            acc = mh_mfma_bf16_32(Ah[t], Bh[dx], acc);
            if (X3) {
                acc = mh_mfma_bf16_32(Ah[t], Bl[dx], acc);
                acc = mh_mfma_bf16_32(Al[t], Bh[dx], acc);
            }
        }
    };
    // epilogue of output row y: lane = pixel x0 + lp, register quad q = channels 8q + 4 lh .. +3
    // Every access goes through a descriptor with an out-of-range offset for the lanes / channel groups that do not exist: straight-line code,
    // so the compiler can count the outstanding stores and the next row's loads stay in flight across the epilogue.
    const __amdgpu_buffer_rsrc_t rs_out = mh_make_rsrc(p.out, p.out_bytes);
    const __amdgpu_buffer_rsrc_t rs_mask = mh_make_rsrc(p.mask_ref, p.mask_ref ? p.mask_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rs_msh = mh_make_rsrc(SH ? (const void*)p.mask_shadow : (const void*)p.out, SH ? p.mask_shadow_bytes : 0u);
    const int msh_ld = (p.N + 31) & ~31;
    const __amdgpu_buffer_rsrc_t rs_sh = mh_make_rsrc(p.shadow, p.shadow ? (unsigned)p.M * (unsigned)p.shadow_ld * 2u : 0u);
    float4 oldv[NQ], mkv[NQ];
    auto epi_load = [&](int y) {
        if (!EPI) return;
        const int x = x0 + lp;
        const int m = (b * p.Ho + y) * p.Wo + x;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int n = 8 * q + 4 * lh;
            const bool ok = y >= 0 && x < p.Wo && n < p.N;
            oldv[q] = mh_buf_load4(rs_out, (ok && p.accumulate) ? (m * p.out_ld + n) * 4 : MH_OOB);
            if (SH) {        // sign of the activation's bf16 shadow (pixel stride = N rounded up to 32 halfs)
                const u32x2 m2 = __builtin_amdgcn_raw_buffer_load_b64(rs_msh, ok ? (m * msh_ld + n) * 2 : MH_OOB, 0, 0);
                mkv[q] = make_float4(__builtin_bit_cast(float, m2[0] << 16), __builtin_bit_cast(float, m2[0] & 0xffff0000u),
                                     __builtin_bit_cast(float, m2[1] << 16), __builtin_bit_cast(float, m2[1] & 0xffff0000u));
            } else
            mkv[q] = mh_buf_load4(rs_mask, ok ? (m * p.mask_ld + n) * 4 : MH_OOB);
        }
    };
    auto store_row = [&](const f32x16& acc, int y) {
        const int x = x0 + lp;
        const int m = (b * p.Ho + y) * p.Wo + x;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int n = 8 * q + 4 * lh;
            const bool ok = y >= 0 && x < p.Wo && n < p.N;
            float4 v = make_float4(acc[4 * q + 0], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
            const float4 bv = *reinterpret_cast<const float4*>(sBias + n);
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            if (p.alpha != 1.0f) {
                v.x = v.x > 0.f ? v.x : p.alpha * v.x; v.y = v.y > 0.f ? v.y : p.alpha * v.y;
                v.z = v.z > 0.f ? v.z : p.alpha * v.z; v.w = v.w > 0.f ? v.w : p.alpha * v.w;
            }
            if (EPI) {
                v.x += oldv[q].x; v.y += oldv[q].y; v.z += oldv[q].z; v.w += oldv[q].w;          // (zeros without accumulate)
                if (p.mask_ref) {
                    const float4 mk = mkv[q];
                    v.x *= (mk.x > 0.f || n + 0 < p.mask_c0 || n + 0 >= p.mask_c1) ? 1.0f : p.mask_alpha;
                    v.y *= (mk.y > 0.f || n + 1 < p.mask_c0 || n + 1 >= p.mask_c1) ? 1.0f : p.mask_alpha;
                    v.z *= (mk.z > 0.f || n + 2 < p.mask_c0 || n + 2 >= p.mask_c1) ? 1.0f : p.mask_alpha;
                    v.w *= (mk.w > 0.f || n + 3 < p.mask_c0 || n + 3 >= p.mask_c1) ? 1.0f : p.mask_alpha;
                }
            }
            mh_buf_store4(rs_out, ok ? (m * p.out_ld + n) * 4 : MH_OOB, v);
            mh_buf_store2(rs_sh, ok ? (m * p.shadow_ld + n) * 2 : MH_OOB, mh_pack_bf16(v.x, v.y), mh_pack_bf16(v.z, v.w));
        }
    };

    if constexpr (S == 1) {
        // ---- the walk: input rows r0 - 1 .. r1 (pad 1); input row r is tap row 0 of output row r + 1, 1 of r, 2 of r - 1 ---------------------
        // one step: convert the loaded row, start the next row's loads, 3 x 3 (x 3) MFMAs, store the output row the step completed
        // the loads run TWO rows ahead (two register buffers, alternating): one row of MFMAs is too short to cover the HBM latency
        auto step = [&](f32x16& aNew, f32x16& aMid, f32x16& aOld, float4 (&buf)[3][2], int r) {
            u32x4 Bh[3], Bl[3];
            cvt_row(buf, Bh, Bl);
            if (EPI) __builtin_amdgcn_sched_barrier(0);      // the row's loads are consumed BEFORE anything new is issued (else the wait covers the new loads too)
            // (unconditional on purpose -- a row index of -1 reads zeros: loads under a branch turn their destination registers into phi copies
            //  that hipcc resolves with moves behind an s_waitcnt vmcnt(0), i.e. no prefetch at all)
            epi_load(r - 1 >= r0 ? r - 1 : -1);
            load_row(buf, r + 2 <= r1 ? r + 2 : -1);
            if (r + 1 < r1) { aNew = zero16; mac(aNew, 0, Bh, Bl); }
            if (r >= r0 && r < r1) mac(aMid, 1, Bh, Bl);
            if (r - 1 >= r0) mac(aOld, 2, Bh, Bl);
            store_row(aOld, r - 1 >= r0 ? r - 1 : -1);          // (row -1: every offset out of range.  Unconditional for the same reason as the loads: behind
                                                                //  a branch join hipcc falls back to the smaller of the two outstanding-store counts)
        };
        f32x16 a0 = zero16, a1 = zero16, a2 = zero16;
        // output row y lives in a[(y - r0) % 3]: at input row r (i = r - r0 + 1): new = y = r + 1 -> (i) % 3, mid -> (i - 1) % 3, old -> (i - 2) % 3;
        // row buffer i % 2
        for (int r = r0 - 1; r <= r1; r += 6) {
            step(a0, a2, a1, raw, r);                    // i = 0: new -> a0 ; mid = row r0 - 1 (outside) ; old = outside
            if (r + 1 <= r1) step(a1, a0, a2, rawb, r + 1);
            if (r + 2 <= r1) step(a2, a1, a0, raw, r + 2);
            if (r + 3 <= r1) step(a0, a2, a1, rawb, r + 3);
            if (r + 4 <= r1) step(a1, a0, a2, raw, r + 4);
            if (r + 5 <= r1) step(a2, a1, a0, rawb, r + 5);
        }
    } else {
        // ---- stride 2: output row y <- input rows 2y - pad_t + {0, 1, 2} ---------------------------------------------------------------------
        u32x4 Ph[3], Pl[3], Qh[3], Ql[3], Th[3], Tl[3];
        cvt_row(raw, Ph, Pl);
        load_row(raw, 2 * r0 - p.pad_t + 1);
        load_row(rawb, 2 * r0 - p.pad_t + 2);
        for (int y = r0; y < r1; ++y) {
            cvt_row(raw, Qh, Ql);
            cvt_row(rawb, Th, Tl);
            if (EPI) __builtin_amdgcn_sched_barrier(0);
            epi_load(y);
            load_row(raw, y + 1 < r1 ? 2 * y + 3 - p.pad_t : -1);
            load_row(rawb, y + 1 < r1 ? 2 * y + 4 - p.pad_t : -1);
            f32x16 acc = zero16;
            mac(acc, 0, Ph, Pl); mac(acc, 1, Qh, Ql); mac(acc, 2, Th, Tl);
            store_row(acc, y);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) { Ph[dx] = Th[dx]; Pl[dx] = Tl[dx]; }
        }
    }
}

// minimum output pixels (B * H * W) of a layer for this kernel; 0 = off.  Default 65536: the 1/2-resolution layers (the thin tiles of the patch /
// bank kernels keep the 32 -> 32 layers at 1/4 resolution)
std::atomic<int> g_rows_minpix{-1};
int rows_minpix() {
    const int v = g_rows_minpix.load(std::memory_order_relaxed);
    return v < 0 ? 65536 : v;
}

}  // namespace

// 3x3 "SAME" layers with <= 16 input and <= 32 output channels (a multiple of 8), many pixels, bf16 or split-bf16 arithmetic: stride 1 (forward and
// input gradient), stride 2 (forward)
bool mh_conv_rows_ok(const ConvArgs& a) {
    const int minpix = rows_minpix();
    if (minpix <= 0) return false;
    if (!(a.bf16 || (a.x3 && a.mode == 0))) return false;
    if (!(a.kh == 3 && a.kw == 3 && a.dil == 1)) return false;
    const bool s1 = a.stride == 1 && a.pad_t == 1 && a.pad_l == 1 && a.Hi == a.Ho && a.Wi == a.Wo && a.K % 4 == 0;
    const bool s2 = a.stride == 2 && a.mode == 0 && a.pad_t >= 0 && a.pad_t <= 1 && a.pad_l >= 0 && a.pad_l <= 1 &&
                    a.Ho == (a.Hi + 1) / 2 && a.Wo == (a.Wi + 1) / 2 && (a.K % 4 == 0 || a.K < 4);
    if (!(s1 || s2)) return false;
    if (a.ncls != 0 || a.K < 2 || a.K > 16 || a.N > 32 || a.N % 8 != 0 || !a.vecA || !a.vecC) return false;
    if (a.x3 && (a.accumulate || a.mask_ref)) return false;        // (the split-bf16 instances have no registers left for the pre-loaded epilogue operands)
    if ((int64_t)a.B * a.Ho * a.Wo * ((a.N + 31) / 32 * 32) * 2 >= (1ll << 31) - 64) return false;       // 32-bit offsets into the shadow
    return (int64_t)a.B * a.Ho * a.Wo >= minpix;
}

int mh_conv_rows_launch(ConvArgs& a, hipStream_t s) {
    // units = strips x row blocks x images; aim at ~2 waves per SIMD (2048 waves) with at least 4 rows per wave (stride 1: the 2 halo rows are
    // re-loaded; stride 2: one of the 2R + 1).  Measured at 192x640x2 (profiles/r03_experiments.txt #13): R = 3 / 4 / 6 / 8 -> 22.5 / 18.0 / 20.8 / 20.6 us
    const int strips = mh_cdiv(a.Wo, 32);
    int R = 4;
    while ((int64_t)a.B * strips * mh_cdiv(a.Ho, R) > 2048 && R < 16) ++R;
    if (a.stride == 2) while (R > 1 && (int64_t)a.B * strips * mh_cdiv(a.Ho, R) < 1024) --R;       // (one shared row per block: short blocks are cheap there)
    const int rblocks = mh_cdiv(a.Ho, R);
    const int units = a.B * strips * rblocks;
    const int grid = mh_cdiv(units, 4);
    const bool epi = a.accumulate || a.mask_ref;
    const unsigned mulK = (unsigned)(((1ull << 32) + a.K - 1) / a.K), mulN = (unsigned)(((1ull << 32) + a.N - 1) / a.N);
#define MH_ROWS(Sv, X3v, EPIv)                                                                                                          \
    { if (a.N <= 16) hipLaunchKernelGGL((conv_rows_kernel<Sv, X3v, EPIv, 2>), dim3(grid), dim3(256), 0, s, a, R, strips, rblocks, mulK, mulN);       \
      else hipLaunchKernelGGL((conv_rows_kernel<Sv, X3v, EPIv, 4>), dim3(grid), dim3(256), 0, s, a, R, strips, rblocks, mulK, mulN); }
    const bool sh = a.stride == 1 && !a.x3 && epi && a.in_shadow && a.mask_shadow && a.mask_ref;
    if (sh) {
        if (a.N <= 16) hipLaunchKernelGGL((conv_rows_kernel<1, false, true, 2, true>), dim3(grid), dim3(256), 0, s, a, R, strips, rblocks, mulK, mulN);
        else hipLaunchKernelGGL((conv_rows_kernel<1, false, true, 4, true>), dim3(grid), dim3(256), 0, s, a, R, strips, rblocks, mulK, mulN);
    } else
    if (a.stride == 1) {
        if (a.x3) MH_ROWS(1, true, false)
        else if (epi) MH_ROWS(1, false, true)
        else MH_ROWS(1, false, false)
    } else {
        if (a.x3) MH_ROWS(2, true, false)
        else if (epi) MH_ROWS(2, false, true)
        else MH_ROWS(2, false, false)
    }
#undef MH_ROWS
    mh_note_kernel("conv_rows_kernel<%s,%s,s%d%s> R=%d grid %d", a.mode == 1 ? "dgrad" : "fwd", a.x3 ? "bf16x3" : "bf16", a.stride, sh ? ",shadows" : "", R, grid);
    return mh_check_launch("conv_rows");
}

extern "C" int mh_tune_conv_rows(int min_pixels) { return g_rows_minpix.exchange(min_pixels < 0 ? -1 : min_pixels); }

// ---- the image layer, forward, straight from the frames (round 5: mh_conv_image_fwd) -----------------------------------------------------------
// MADNet's conv1 (3 -> 16, 3x3, stride 2, Nets/MadNet.py:56-60) on the reflect-padded pair (Stereo_net._preprocess_inputs -> pad_image): 0.2 GFLOP over
// 11 MB of pixels and 16 + 16 MB of results.  As two launches -- mh_pad_reflect writing the padded copy (11 us), then the row-streaming kernel above with 3 of
// its 16 MFMA reduction lanes alive (18 us) -- it was 29 us at the head of the forward chain, where nothing overlaps it.  Here a THREAD owns an output pixel:
// its 27 inputs are loaded from the ORIGINAL frames through the reflection (the padded copy is not read: it is only needed by this layer's filter gradient,
// at the other end of the step, and is written on a side lane), the bank is read through the scalar cache (uniform addresses: s_load, operands of v_fmac),
// 432 exact-fp32 FMAs, leaky, four 16-byte stores + the bf16 shadow.  HBM bound (8 - 10 us).  Exact fp32: no rounding of the frames at all.
struct ImageConvArgs {
    const float* raw; const float* w; const float* bias; float* out; unsigned short* shadow;
    int NB, H0, W0, Hp, Wp, rpt, rpl, Ho, Wo, stride, pad_t, pad_l, out_ld, shadow_ld, M;
    float div, sub, alpha;
    unsigned raw_bytes;
};

namespace {
__global__ __launch_bounds__(256) void conv_image_fwd_kernel(ImageConvArgs p) {
    constexpr int C = 3, N = 16;
    const int m = blockIdx.x * 256 + threadIdx.x;
    const bool live = m < p.M;
    const int mm = live ? m : 0;
    const int ox = mm % p.Wo;
    const int t2 = mm / p.Wo;
    const int oy = t2 % p.Ho, b = t2 / p.Ho;
    const __amdgpu_buffer_rsrc_t rs = mh_make_rsrc(p.raw, p.raw_bytes);
    // all 27 loads are UNCONDITIONAL (a tap outside the padded frame = an out-of-range offset = 0): a load behind a branch costs hipcc its count of what is in
    // flight, and the 27 requests become 27 round trips (the first build of this kernel: s_cbranch_execz + s_waitcnt vmcnt(0) around every one of them)
    float xv[9][C];
    unsigned inmask = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int ky = t / 3, kx = t - ky * 3;
        const int py = oy * p.stride + ky - p.pad_t, px = ox * p.stride + kx - p.pad_l;          // in the padded frame: outside it the SAME padding's zeros
        const bool in = live && (unsigned)py < (unsigned)p.Hp && (unsigned)px < (unsigned)p.Wp;
        int sy = py - p.rpt, sx = px - p.rpl;                                                    // mh_pad_reflect's mapping
        sy = sy < 0 ? -sy : (sy >= p.H0 ? 2 * (p.H0 - 1) - sy : sy);
        sx = sx < 0 ? -sx : (sx >= p.W0 ? 2 * (p.W0 - 1) - sx : sx);
        int off = (((b * p.H0 + sy) * p.W0 + sx) * C) * 4;
        MH_KEEP_VGPR(off);
        off = in ? off : MH_OOB;
        inmask |= in ? (1u << t) : 0u;
#pragma unroll
        for (int c = 0; c < C; ++c) xv[t][c] = mh_buf_load1(rs, off == MH_OOB ? MH_OOB : off + 4 * c);
    }
    if (p.div != 1.0f || p.sub != 0.f) {             // (uniform; MADNet feeds the frames as they are, DispNet-style preprocessing divides and shifts)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < C; ++c) xv[t][c] = ((inmask >> t) & 1u) ? (xv[t][c] / p.div - p.sub) : 0.f;
    }
    float acc[N];
#pragma unroll
    for (int n = 0; n < N; ++n) acc[n] = p.bias ? MH_CONST_F32_PTR(p.bias)[n] : 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const MH_CONST_F32* wr = MH_CONST_F32_PTR(p.w) + (t * C + c) * N;           // (uniform address, constant address space: s_load, operands of v_fmac)
#pragma unroll
            for (int n = 0; n < N; ++n) acc[n] += xv[t][c] * wr[n];
        }
#pragma unroll
    for (int n = 0; n < N; ++n) acc[n] = (p.alpha == 1.0f || acc[n] > 0.f) ? acc[n] : p.alpha * acc[n];
    // The results leave through LDS: a thread holds the 64 bytes of ITS pixel, so a 16-byte store instruction would touch 64 lines with a quarter line each.
    // Transposed, consecutive lanes write consecutive 16 bytes: four fully coalesced 1 KB stores per wave (and two for the shadow).
    __shared__ __attribute__((aligned(16))) float so[256 * 20];           // [thread][16 + 4 pad]: 80-byte rows keep the float4 writes of 8 neighbours on distinct banks
    const int tid = threadIdx.x;
#pragma unroll
    for (int q = 0; q < N / 4; ++q) *reinterpret_cast<float4*>(so + tid * 20 + 4 * q) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    __syncthreads();
    const int m0 = blockIdx.x * 256;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = j * 256 + tid;                 // (pixel of the workgroup, channel quad): consecutive lanes = consecutive 16 bytes of out
        const int pl = e >> 2, q = e & 3;
        const float4 v = *reinterpret_cast<const float4*>(so + pl * 20 + 4 * q);
        const int mo = m0 + pl;
        if (mo < p.M) *reinterpret_cast<float4*>(p.out + (int64_t)mo * p.out_ld + 4 * q) = v;
    }
    if (p.shadow) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int e = j * 256 + tid;             // (pixel, channel octet)
            const int pl = e >> 1, q = e & 1;
            const float4 a = *reinterpret_cast<const float4*>(so + pl * 20 + 8 * q), b2 = *reinterpret_cast<const float4*>(so + pl * 20 + 8 * q + 4);
            const int mo = m0 + pl;
            if (mo < p.M)
                *reinterpret_cast<u32x4*>(p.shadow + (int64_t)mo * p.shadow_ld + 8 * q) =
                    (u32x4){mh_pack_bf16(a.x, a.y), mh_pack_bf16(a.z, a.w), mh_pack_bf16(b2.x, b2.y), mh_pack_bf16(b2.z, b2.w)};
        }
    }
}
}  // namespace

extern "C" int mh_conv_image_ok(int32_t C, int32_t N, int32_t kh, int32_t kw, int32_t stride) {
    return (C == 3 && N == 16 && kh == 3 && kw == 3 && (stride == 1 || stride == 2)) ? 1 : 0;
}

extern "C" int mh_conv_image_fwd(const float* frames, int32_t NB, int32_t H0, int32_t W0, int32_t C, int32_t Hp, int32_t Wp, int32_t reflect_t, int32_t reflect_l,
                                 float div, float sub, const float* w, const float* bias, int32_t N, int32_t stride, int32_t pad_t, int32_t pad_l, float alpha,
                                 float* out, int32_t out_ld, void* shadow, int32_t shadow_ld, void* stream) {
    MH_REQUIRE(frames && w && out, MH_ERR_ARG, "mh_conv_image_fwd: null argument");
    MH_REQUIRE(mh_conv_image_ok(C, N, 3, 3, stride) == 1, MH_ERR_UNSUPPORTED, "mh_conv_image_fwd: serves 3 -> 16 channels, 3x3, stride 1 / 2 (got %d -> %d, stride %d)", C, N, stride);
    MH_REQUIRE(NB > 0 && H0 > 1 && W0 > 1 && Hp >= H0 && Wp >= W0 && reflect_t >= 0 && reflect_l >= 0 && reflect_t < H0 && reflect_l < W0 && Hp - H0 - reflect_t < H0 &&
               Wp - W0 - reflect_l < W0 && Hp - H0 - reflect_t >= 0 && Wp - W0 - reflect_l >= 0, MH_ERR_ARG, "mh_conv_image_fwd: bad frame / padding geometry");
    MH_REQUIRE(div != 0.f && pad_t >= 0 && pad_t <= 1 && pad_l >= 0 && pad_l <= 1, MH_ERR_ARG, "mh_conv_image_fwd: div must be non-zero, SAME padding offsets 0 / 1");
    MH_REQUIRE(out_ld >= N && out_ld % 4 == 0 && mh_aligned16(out) && mh_aligned16(w) && (!shadow || (shadow_ld >= N && shadow_ld % 8 == 0 && mh_aligned16(shadow))), MH_ERR_ALIGN,
               "mh_conv_image_fwd: 16-byte rows required");
    const int Ho = (Hp + stride - 1) / stride, Wo = (Wp + stride - 1) / stride;
    const int64_t M = (int64_t)NB * Ho * Wo, rb = (int64_t)NB * H0 * W0 * C * 4;
    MH_REQUIRE(M < (1ll << 31) - 256 && rb < (1ll << 31) - 64, MH_ERR_UNSUPPORTED, "mh_conv_image_fwd: tensors must be < 2 GiB");
    ImageConvArgs a;
    a.raw = frames; a.w = w; a.bias = bias; a.out = out; a.shadow = (unsigned short*)shadow;
    a.NB = NB; a.H0 = H0; a.W0 = W0; a.Hp = Hp; a.Wp = Wp; a.rpt = reflect_t; a.rpl = reflect_l; a.Ho = Ho; a.Wo = Wo; a.stride = stride; a.pad_t = pad_t; a.pad_l = pad_l;
    a.out_ld = out_ld; a.shadow_ld = shadow_ld; a.M = (int)M; a.div = div; a.sub = sub; a.alpha = alpha; a.raw_bytes = (unsigned)rb;
    hipLaunchKernelGGL(conv_image_fwd_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    mh_note_kernel("conv_image_fwd_kernel<3,16,s%d> grid %d", stride, (int)((M + 255) / 256));
    return mh_check_launch("conv_image_fwd");
}
