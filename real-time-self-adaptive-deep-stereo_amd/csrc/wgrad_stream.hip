// wgrad_stream.hip -- filter gradients of the stride-1 3x3 (dilated) layers from bf16 NHWC "shadows" of the activations / gradient maps,
// every layer of a backward batch in ONE launch.  Gradient of tf.nn.conv2d / atrous_conv2d + bias_add (Nets/sharedLayers.py:54-77) as TF's
// Conv2DBackpropFilter / BiasAddGrad compute it for Stereo_Online_Adaptation.py:126-128's minimize().
//
//   dW[ky][kx][k][n] = sum_{b,y,x} X[b][y + (ky-1) d][x + (kx-1) d][k] * dZ[b][y][x][n]        (zero outside the image)
//
// Why a third filter-gradient kernel (profiles/r03_pmc_wgrad_tiled_vs_taps.txt): the tiled kernel (wgrad.hip) gives every tap its own workgroup, so
// each of them pulls the same fp32 pixels out of L2, rounds them to bf16 and transposes them through registers into [channel][pixel] LDS tiles --
// 3.9 M VALU + 0.6 M LDS instructions (49 % of the LDS cycles bank conflicts) around 8.9 M MFMA cycles for the 128->128 layer, MFMA busy 31 % of the
// wave cycles, and 17 MB of per-split partial sums per layer.  Here
//   * the operands arrive as bf16 NHWC shadows (mh_shadow_cast, or the producing kernel's epilogue): half the bytes, no conversion pass;
//   * a WAVE owns dW[9 taps][32 input channels][32 output channels] (144 accumulator registers, v_mfma_f32_32x32x16_bf16) and walks a vertical
//     run of 32-pixel row segments: per step it needs the dz segment (32 pixels x 32 channels) and ONE new input row (32 + 2d pixels x 32 channels;
//     the three rows of the 3x3 window live in a ring).  Both are copied straight from global memory into the wave's own LDS ring by LDS DMA
//     (buffer_load_dwordx4 ... lds): no loader waves, no conversion, no LDS store instructions, and NO BARRIER in the walk -- a wave waits on its
//     own vmcnt only;
//   * ds_read_b64_tr_b16 turns [pixel][channel] rows into the MFMA operand order, so a tap is a byte offset into the ring: 20 transposing reads
//     feed 9 MFMAs; a 32-lane half reads 4 whole 64-byte pixel rows = all 64 banks once (conflict free without padding or swizzle);
//   * the 8 (or 4) waves of a workgroup work on DIFFERENT pixel runs of the same tile and add their accumulators through LDS at the end, so a
//     workgroup writes ONE partial tile: the split workspace shrinks ~8x against one partial per wave;
//   * one grid serves every layer of a batch (table in device memory): the chip's workgroups are divided over (layer, tile, pixel split) in
//     proportion to the work, each wave streams tens of rows instead of a handful, and the dispatch latency is paid once per batch.
// Partial tiles go to ws[split][9][K][N] (the layout mh_wgrad_reduce sums), or straight to dw when a layer has a single split; the bias partial sums of a
// layer with several splits follow as ws[splits * 9*K*N + split * N + n] (fixed-order reduction instead of atomics: bit-identical replays).
#include "mh_common.h"
#include <stdlib.h>
#include <atomic>

namespace {

constexpr int WS_FLAG = 0x40000000;        // "out of range" marker inside buffer offsets: shadows are < 1 GiB, so flag + anything stays out of range
constexpr int WS_ZSLOT = 2048;             // dz ring slot: 32 pixels x 32 channels x 2 bytes

struct WsGeo { int nw, wave_bytes; };

// ---- shadow cast: fp32 NHWC (any channel stride) -> bf16 NHWC with the channel count padded to a multiple of 32 (pad = 0) -----------------
__global__ __launch_bounds__(256) void shadow_cast_kernel(const mh_shadow_seg* __restrict__ segs, int nseg) {
    const int lo = mh_find_seg(segs, nseg, (int)blockIdx.x);
    const mh_shadow_seg sg = segs[lo];
    const int g8 = sg.dst_ld >> 3;                                   // 8-channel groups per pixel
    const int64_t item = (int64_t)((int)blockIdx.x - sg.blk0) * 256 + threadIdx.x;
    if (item >= sg.npix * g8) return;
    const int64_t pix = item / g8;
    const int c0 = (int)(item - pix * g8) * 8;
    const float* s = sg.src + pix * sg.src_ld + c0;
    float v[8];
    if (c0 + 8 <= sg.C && (sg.src_ld & 3) == 0 && ((uintptr_t)sg.src & 15) == 0) {
        const float4 a = *reinterpret_cast<const float4*>(s), b = *reinterpret_cast<const float4*>(s + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (c0 + e < sg.C) ? s[e] : 0.f;
    }
    u32x4 o;
    o[0] = mh_pack_bf16(v[0], v[1]); o[1] = mh_pack_bf16(v[2], v[3]); o[2] = mh_pack_bf16(v[4], v[5]); o[3] = mh_pack_bf16(v[6], v[7]);
    *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(sg.dst) + pix * sg.dst_ld + c0) = o;
}

__global__ __launch_bounds__(256) void shadow_cast_one_kernel(mh_shadow_seg sg) {
    const int g8 = sg.dst_ld >> 3;
    const int64_t item = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (item >= sg.npix * g8) return;
    const int64_t pix = item / g8;
    const int c0 = (int)(item - pix * g8) * 8;
    const float* s = sg.src + pix * sg.src_ld + c0;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (c0 + e < sg.C) ? s[e] : 0.f;
    u32x4 o;
    o[0] = mh_pack_bf16(v[0], v[1]); o[1] = mh_pack_bf16(v[2], v[3]); o[2] = mh_pack_bf16(v[4], v[5]); o[3] = mh_pack_bf16(v[6], v[7]);
    *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(sg.dst) + pix * sg.dst_ld + c0) = o;
}

// ---- the streaming kernel ---------------------------------------------------------------------------------------------------------------
// NXG: LDS-DMA instructions per input row slot (slot = NXG x 16 pixels >= 32 + 2 d: 3 for d <= 8, 4 for d = 16); D: prefetch distance in steps.
// S: stride of the forward convolution (1, or 2 = the pyramid's down-sampling layers: 'SAME' on even sizes pads only behind, so output pixel
// (y, x) reads input rows 2y .. 2y + 2 and columns 2x .. 2x + 2 -- two new input rows of 65 pixels per step, pixel stride 2 in the reads).
template <int NXG, int D, int S>
__device__ __forceinline__ void wgrad_stream_body(const mh_wgs_layer& L, const int id, float* const smem_f) {
    constexpr int XSLOT = NXG * 1024, RX = S * D + 3, RZ = D + 1, G = S * NXG + 2;
    constexpr int WAVE_BYTES = RX * XSLOT + RZ * WS_ZSLOT;
    unsigned char* const smem = reinterpret_cast<unsigned char*>(smem_f);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NW = (int)blockDim.x >> 6;
    const int tiles = L.ktiles * L.ntiles;
    const int tile = id % tiles, split = id / tiles;
    const int kt = tile / L.ntiles, nt = tile - kt * L.ntiles;
    const int k0 = kt * 32, n0 = nt * 32;
    const int d = L.dil;
    const int nsx = (L.W + 31) >> 5, Hl = (L.H + d - 1) / d;
    const int Sg = L.B * d * nsx * Hl;                                  // row segments: (b, cy, 32-column strip, lattice row), lattice row fastest
    const int WT = L.splits * NW, wv = split * NW + wave;
    const int sq = Sg / WT, sr = Sg - sq * WT;
    int s = wv * sq + (wv < sr ? wv : sr);
    const int s1 = s + sq + (wv < sr ? 1 : 0);

    const int Hx = S * L.H, Wx = S * L.W;                                // input image size (the layer record carries the OUTPUT size)
    const mh_dma_src rs_x = mh_make_dma_src(L.x, (unsigned)((int64_t)L.B * Hx * Wx * L.x_ld * 2));
    const mh_dma_src rs_z = mh_make_dma_src(L.dz, (unsigned)((int64_t)L.B * L.H * L.W * L.dz_ld * 2));
    unsigned char* const xw = smem + wave * WAVE_BYTES;                 // this wave's input-row ring, then its dz ring
    unsigned char* const zw = xw + RX * XSLOT;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;
    const bool do_bias = (L.db != nullptr) && kt == 0;

    // transposing-read lane constants (bytes inside a slot): pixel 8 (l >> 5) + ((l & 15) >> 2) of the 16-pixel half-step, 4-channel piece
    // 4 ((l >> 4) & 1) + (l & 3); the second read of a fragment is 4 pixels (256 bytes) further, the second half-step 16 pixels (1024 bytes)
    const int lpx = 8 * (lane >> 5) + ((lane & 15) >> 2), lch = (4 * ((lane >> 4) & 1) + (lane & 3)) * 8;
    const int lrd = lpx * 64 + lch;                                     // dz slots (and input slots at stride 1)
    const int lrx = S * lpx * 64 + lch;                                 // input slots: pixel stride S
    const int kxb = d * 64;                                             // byte step of one tap column

    while (s < s1) {
        // ---- one vertical run: rows r0 .. r0 + n - 1 of column c -----------------------------------------------------------------------
        const int c = s / Hl, r0 = s - c * Hl;
        int n = Hl - r0;
        if (n > s1 - s) n = s1 - s;
        const int sx = c % nsx;
        const int t2 = c / nsx;
        const int cy = t2 % d, b = t2 / d;
        const int x0 = sx * 32;
        int xl[NXG], zl[2];
#pragma unroll
        for (int g = 0; g < NXG; ++g) {
            const int cc = g * 64 + lane, p = cc >> 2, qq = cc & 3;
            const int x = (S == 1) ? x0 - d + p : 2 * x0 + p;            // stride 2: input columns 2 x0 .. 2 x0 + 64
            const bool ok = (p < (S == 1 ? 32 + 2 * d : 65)) && x >= 0 && x < Wx;
            xl[g] = ok ? (x * L.x_ld + k0 + qq * 8) * 2 : WS_FLAG;
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int cc = g * 64 + lane, p = cc >> 2, qq = cc & 3;
            const int x = x0 + p;
            zl[g] = (x < L.W) ? (x * L.dz_ld + n0 + qq * 8) * 2 : WS_FLAG;
        }
        const int xrow = Wx * L.x_ld * 2, zrow = L.W * L.dz_ld * 2;
        auto issue_x = [&](int rr, bool live, int slot) {                // input row rr of this column (lattice row at stride 1, image row at stride 2; outside = zeros)
            const int y = (S == 1) ? cy + d * rr : rr;
            const int base = (live && rr >= 0 && y < Hx) ? (b * Hx + y) * xrow : WS_FLAG;
#pragma unroll
            for (int g = 0; g < NXG; ++g) mh_glds16(rs_x, xw + slot * XSLOT + g * 1024, xl[g] + base);
        };
        auto issue_z = [&](int rr, bool live, int slot) {
            const int y = cy + d * rr;
            const int base = (live && y < L.H) ? (b * L.H + y) * zrow : WS_FLAG;
#pragma unroll
            for (int g = 0; g < 2; ++g) mh_glds16(rs_z, zw + slot * WS_ZSLOT + g * 1024, zl[g] + base);
        };
        // prologue: stride 1: rows r0 - 1, r0, then the load groups of steps 0 .. D - 1 (group j = input row r0 + j + 1 and dz row r0 + j);
        // stride 2: row 2 r0, then groups j = input rows 2 (r0 + j) + 1, 2 (r0 + j) + 2 and dz row r0 + j
        if (S == 1) {
            issue_x(r0 - 1, true, 0);
            issue_x(r0, true, 1);
        } else issue_x(2 * r0, true, 0);
#pragma unroll
        for (int j = 0; j < D; ++j) {
            if (S == 1) issue_x(r0 + j + 1, j < n, (j + 2) % RX);
            else {
                issue_x(2 * (r0 + j) + 1, j < n, (2 * j + 1) % RX);
                issue_x(2 * (r0 + j) + 2, j < n, (2 * j + 2) % RX);
            }
            issue_z(r0 + j, j < n, j % RZ);
        }
        int xs = 0;                                                      // ring slot of the first input row of step i
        int zs = 0;                                                      // ring slot of dz row (r0 + i)
        int xl_next = (S == 1 ? D + 2 : 2 * D + 1) % RX, zl_next = D % RZ;        // slots of the next load group
        for (int i = 0; i < n; ++i) {
            // every ds_read of step i - 1 has returned (its MFMAs consumed them): the slots the next group overwrites are free
            if (S == 1) issue_x(r0 + i + D + 1, i + D < n, xl_next);
            else {
                issue_x(2 * (r0 + i + D) + 1, i + D < n, xl_next);
                if (++xl_next == RX) xl_next = 0;
                issue_x(2 * (r0 + i + D) + 2, i + D < n, xl_next);
            }
            issue_z(r0 + i + D, i + D < n, zl_next);
            if (++xl_next == RX) xl_next = 0;
            if (++zl_next == RZ) zl_next = 0;
            MH_WAIT_VMCNT(D * G);                                        // all but the newest D groups have landed: step i's rows are in LDS
            const unsigned short* const zb = reinterpret_cast<const unsigned short*>(zw + zs * WS_ZSLOT + lrd);
            u32x4 bf[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint2 b0 = mh_lds_read_tr16(zb + h * 512), b1 = mh_lds_read_tr16(zb + h * 512 + 128);
                bf[h] = (u32x4){b0.x, b0.y, b1.x, b1.y};
            }
            if (do_bias) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        bsum += __builtin_bit_cast(float, bf[h][e] << 16) + __builtin_bit_cast(float, bf[h][e] & 0xffff0000u);
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                int sl = xs + ky;
                if (sl >= RX) sl -= RX;
                const unsigned char* const xrow_b = xw + sl * XSLOT + lrx;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const unsigned short* const ab = reinterpret_cast<const unsigned short*>(xrow_b + kx * kxb);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint2 a0 = mh_lds_read_tr16(ab + h * (S * 512)), a1 = mh_lds_read_tr16(ab + h * (S * 512) + S * 128);
                        const u32x4 af = (u32x4){a0.x, a0.y, a1.x, a1.y};
                        acc[ky * 3 + kx] = mh_mfma_bf16_32(af, bf[h], acc[ky * 3 + kx]);
                    }
                }
            }
            xs += S;
            if (xs >= RX) xs -= RX;
            if (++zs == RZ) zs = 0;
        }
        MH_WAIT_VMCNT(0);                                                // the trailing (out-of-range) groups must not land in the next run's rows
        s += n;
    }

    // ---- the workgroup's waves add their tiles through LDS (three taps at a time), one partial tile leaves the workgroup ------------------
    __syncthreads();
    float* const red = smem_f;                                           // [NW][3 taps][4 register quads][64 lanes][4]
    float* const bred = smem_f + NW * 3 * 1024;                          // [NW][64] (behind the tile buffer: NW * 12 KB + NW * 256 B <= NW * 16 KB)
    float* const dst = L.ws + (int64_t)split * ((int64_t)9 * L.K * L.N);
    for (int t3 = 0; t3 < 3; ++t3) {
#pragma unroll
        for (int tt = 0; tt < 3; ++tt)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const f32x16& a = acc[t3 * 3 + tt];
                *reinterpret_cast<f32x4*>(red + (((wave * 3 + tt) * 4 + q4) * 64 + lane) * 4) = (f32x4){a[4 * q4], a[4 * q4 + 1], a[4 * q4 + 2], a[4 * q4 + 3]};
            }
        if (t3 == 0) bred[wave * 64 + lane] = bsum;
        __syncthreads();
        for (int it = tid; it < 768; it += (int)blockDim.x) {            // item = (tap of the three, register quad, lane)
            const int tt = it >> 8, q4 = (it >> 6) & 3, ln = it & 63;
            f32x4 v = *reinterpret_cast<const f32x4*>(red + ((tt * 4 + q4) * 64 + ln) * 4);
            for (int w = 1; w < NW; ++w) {
                const f32x4 u = *reinterpret_cast<const f32x4*>(red + (((w * 3 + tt) * 4 + q4) * 64 + ln) * 4);
                v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
            }
            const int nn = n0 + (ln & 31);
            const int kb = k0 + 8 * q4 + 4 * (ln >> 5);                  // C/D layout: row = (r & 3) + 8 (r >> 2) + 4 (l >> 5), col = l & 31
            if (nn < L.N) {
                float* const o = dst + ((int64_t)(t3 * 3 + tt) * L.K + kb) * L.N + nn;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (kb + e < L.K) o[(int64_t)e * L.N] = v[e];
            }
        }
        if (t3 == 0 && do_bias && tid < 32 && n0 + tid < L.N) {
            float t = 0.f;
            for (int w = 0; w < NW; ++w) t += bred[w * 64 + tid] + bred[w * 64 + 32 + tid];
            // round 6: with more than one pixel split the bias partial sums go BEHIND the filter partials, ws[splits][9][K][N] | [splits][N], plain stores summed in
            // split order by the batch's mh_wgrad_reduce (one more segment) -- a float atomic per workgroup and channel made the last bits of every bias gradient
            // depend on the arrival order (scripts/exp/det_probe.py).  A single split is a single addend: the atomic onto the zeroed db is order-free.
            if (L.splits > 1) L.ws[(int64_t)L.splits * ((int64_t)9 * L.K * L.N) + (int64_t)split * L.N + n0 + tid] = t;
            else mh_atomic_add(L.db + n0 + tid, t);
        }
        __syncthreads();
    }
}

__device__ __forceinline__ int wgrad_stream_layer_of(const mh_wgs_layer* __restrict__ tab, int nlayers, int bid) {
    int li = 0;
    for (int q = 1; q < nlayers; ++q)
        if (tab[q].blk0 <= bid) li = q;
    return __builtin_amdgcn_readfirstlane(li);
}

template <int NXG, int D, int S>
__global__ __launch_bounds__(512) void wgrad_stream_kernel(const mh_wgs_layer* __restrict__ tab, int nlayers) {
    HIP_DYNAMIC_SHARED(float, smem_f)
    const int bid = mh_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const mh_wgs_layer L = tab[wgrad_stream_layer_of(tab, nlayers, bid)];
    wgrad_stream_body<NXG, D, S>(L, bid - L.blk0, smem_f);
}

// a table that mixes stride-1 (dilation <= 8) and stride-2 layers -- a pyramid batch: two down-sampling layers and the two layers behind them --
// in ONE grid: the workgroup takes the instance of its layer's stride (wave-uniform), the LDS is sized for the larger ring
__global__ __launch_bounds__(512) void wgrad_stream_mixed_kernel(const mh_wgs_layer* __restrict__ tab, int nlayers) {
    HIP_DYNAMIC_SHARED(float, smem_f)
    const int bid = mh_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const mh_wgs_layer L = tab[wgrad_stream_layer_of(tab, nlayers, bid)];
    if (L.stride == 2) wgrad_stream_body<5, 1, 2>(L, bid - L.blk0, smem_f);
    else wgrad_stream_body<3, 1, 1>(L, bid - L.blk0, smem_f);
}

struct StreamCfg { int nxg, dist; };
template <int NXG, int D, int S = 1>
static int stream_launch(const mh_wgs_layer* tab, int nlayers, int nblocks, int nw, hipStream_t s, bool attr_only) {
    constexpr int WAVE_BYTES = (S * D + 3) * NXG * 1024 + (D + 1) * WS_ZSLOT;
    static std::atomic<uint64_t> attr_done{0};
    const uint64_t attr_dev = mh_device_bit();
    if (!(attr_done.load(std::memory_order_relaxed) & attr_dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_stream_kernel<NXG, D, S>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { mh_set_error("wgrad_stream: hipFuncSetAttribute(160 KB LDS): %s", hipGetErrorString(e)); return (int)e; }
        attr_done.fetch_or(attr_dev);
    }
    if (attr_only) return 0;
    const size_t lds = (size_t)nw * WAVE_BYTES;
    MH_REQUIRE(lds <= 160 * 1024, MH_ERR_UNSUPPORTED, "mh_wgrad_stream: %d waves x %d B of LDS rings exceed 160 KB", nw, WAVE_BYTES);
    mh_note_kernel("wgrad_stream_kernel<%d,%d,s%d> layers %d grid %d x %d waves lds %d", NXG, D, S, nlayers, nblocks, nw, (int)lds);
    hipLaunchKernelGGL((wgrad_stream_kernel<NXG, D, S>), dim3(nblocks), dim3(64 * nw), lds, s, tab, nlayers);
    return mh_check_launch("wgrad_stream");
}

static int stream_launch_mixed(const mh_wgs_layer* tab, int nlayers, int nblocks, int nw, hipStream_t s, bool attr_only) {
    constexpr int WAVE_BYTES = 5 * 5 * 1024 + 2 * WS_ZSLOT;            // the stride-2 instance's rings (the stride-1 ones need 16 KB)
    static std::atomic<uint64_t> attr_done{0};
    const uint64_t attr_dev = mh_device_bit();
    if (!(attr_done.load(std::memory_order_relaxed) & attr_dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_stream_mixed_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { mh_set_error("wgrad_stream: hipFuncSetAttribute(160 KB LDS): %s", hipGetErrorString(e)); return (int)e; }
        attr_done.fetch_or(attr_dev);
    }
    if (attr_only) return 0;
    const size_t lds = (size_t)nw * WAVE_BYTES;
    MH_REQUIRE(lds <= 160 * 1024, MH_ERR_UNSUPPORTED, "mh_wgrad_stream: %d waves x %d B of LDS rings exceed 160 KB", nw, WAVE_BYTES);
    mh_note_kernel("wgrad_stream_mixed_kernel layers %d grid %d x %d waves lds %d", nlayers, nblocks, nw, (int)lds);
    hipLaunchKernelGGL(wgrad_stream_mixed_kernel, dim3(nblocks), dim3(64 * nw), lds, s, tab, nlayers);
    return mh_check_launch("wgrad_stream_mixed");
}

static std::atomic<int> g_stream_dist{0};      // mh_tune_wgrad_stream: prefetch distance (0 = default)

}  // namespace

// one tensor, arguments by value (the fallback of mh_conv2d_sh for kernel families whose epilogue does not write the shadow)
int mh_shadow_cast_one(const float* src, int src_ld, int C, void* dst, int dst_ld, int64_t npix, hipStream_t s) {
    mh_shadow_seg sg;
    sg.src = src; sg.dst = dst; sg.npix = npix; sg.C = C; sg.src_ld = src_ld; sg.dst_ld = dst_ld; sg.blk0 = 0;
    hipLaunchKernelGGL(shadow_cast_one_kernel, dim3((unsigned)((npix * (dst_ld / 8) + 255) / 256)), dim3(256), 0, s, sg);
    return mh_check_launch("shadow_cast_one");
}

int mh_wgrad_stream_init() {
    if (int rc = stream_launch<3, 1>(nullptr, 0, 0, 0, nullptr, true)) return rc;
    if (int rc = stream_launch<3, 2>(nullptr, 0, 0, 0, nullptr, true)) return rc;
    if (int rc = stream_launch<4, 1>(nullptr, 0, 0, 0, nullptr, true)) return rc;
    if (int rc = stream_launch<5, 1, 2>(nullptr, 0, 0, 0, nullptr, true)) return rc;
    if (int rc = stream_launch_mixed(nullptr, 0, 0, 0, nullptr, true)) return rc;
    return 0;
}

extern "C" int mh_tune_wgrad_stream(int dist) { g_stream_dist = dist > 0 ? dist : 0; return 0; }

extern "C" int mh_shadow_cast(const mh_shadow_seg* segs_device, int32_t nseg, int32_t nblocks, void* stream) {
    MH_REQUIRE(segs_device && nseg > 0 && nblocks > 0, MH_ERR_ARG, "mh_shadow_cast: empty segment table");
    hipLaunchKernelGGL(shadow_cast_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, segs_device, nseg);
    return mh_check_launch("shadow_cast");
}

// Host-side planner: divides `target_wgs` workgroups over the (layer, tile, pixel split) space in proportion to the rows each has to stream.
extern "C" int mh_wgrad_stream_plan(mh_wgs_layer* layers, int32_t n, int32_t target_wgs, int32_t nwaves, int32_t* nblocks_out) {
    MH_REQUIRE(layers && n > 0 && n <= 64 && nblocks_out, MH_ERR_ARG, "mh_wgrad_stream_plan: 1 .. 64 layers");
    MH_REQUIRE(nwaves >= 1 && nwaves <= 8, MH_ERR_ARG, "mh_wgrad_stream_plan: 1 .. 8 waves per workgroup");
    if (target_wgs <= 0) target_wgs = 256;
    int64_t maxseg = 1;
    for (int i = 0; i < n; ++i) {
        mh_wgs_layer& L = layers[i];
        MH_REQUIRE(L.B > 0 && L.H > 0 && L.W > 0 && L.K > 0 && L.N > 0 && L.dil >= 1 && L.dil <= 16, MH_ERR_ARG, "mh_wgrad_stream_plan: layer %d: bad geometry", i);
        if (L.stride == 0) L.stride = 1;
        MH_REQUIRE(L.stride == 1 || (L.stride == 2 && L.dil == 1), MH_ERR_UNSUPPORTED, "mh_wgrad_stream_plan: layer %d: stride 1 (any dilation) or stride 2 (dilation 1)", i);
        MH_REQUIRE(L.stride == layers[0].stride || L.dil <= 8, MH_ERR_UNSUPPORTED, "mh_wgrad_stream_plan: a table that mixes strides takes dilations <= 8");
        L.ktiles = mh_cdiv(L.K, 32); L.ntiles = mh_cdiv(L.N, 32);
        MH_REQUIRE(L.x_ld >= L.ktiles * 32 && L.dz_ld >= L.ntiles * 32 && L.x_ld % 8 == 0 && L.dz_ld % 8 == 0, MH_ERR_ARG,
                   "mh_wgrad_stream_plan: layer %d: shadow strides must cover the channel count rounded up to 32", i);
        MH_REQUIRE((int64_t)L.B * L.H * L.W * L.stride * L.stride * L.x_ld * 2 < (int64_t)WS_FLAG && (int64_t)L.B * L.H * L.W * L.dz_ld * 2 < (int64_t)WS_FLAG, MH_ERR_UNSUPPORTED,
                   "mh_wgrad_stream_plan: layer %d: shadows must be < 1 GiB", i);
        const int64_t seg = (int64_t)L.B * L.dil * mh_cdiv(L.W, 32) * mh_cdiv(L.H, L.dil);
        if (seg > maxseg) maxseg = seg;
    }
    // smallest rows-per-workgroup sigma whose split counts fit the target (>= 2 rows per wave: below that the 3-row prologue dominates)
    auto blocks_for = [&](int64_t sigma) {
        int64_t t = 0;
        for (int i = 0; i < n; ++i) {
            const mh_wgs_layer& L = layers[i];
            const int64_t seg = (int64_t)L.B * L.dil * mh_cdiv(L.W, 32) * mh_cdiv(L.H, L.dil);
            t += (int64_t)L.ktiles * L.ntiles * mh_cdiv(seg, sigma);
        }
        return t;
    };
    int64_t lo = 2 * nwaves, hi = maxseg > lo ? maxseg : lo;
    if (blocks_for(lo) > target_wgs) {
        while (lo < hi) {
            const int64_t mid = (lo + hi) / 2;
            if (blocks_for(mid) <= target_wgs) hi = mid; else lo = mid + 1;
        }
    }
    int blk = 0;
    for (int i = 0; i < n; ++i) {
        mh_wgs_layer& L = layers[i];
        const int64_t seg = (int64_t)L.B * L.dil * mh_cdiv(L.W, 32) * mh_cdiv(L.H, L.dil);
        L.splits = mh_cdiv(seg, lo);
        L.blk0 = blk;
        blk += L.ktiles * L.ntiles * L.splits;
    }
    *nblocks_out = blk;
    return 0;
}

extern "C" int mh_wgrad_stream(const mh_wgs_layer* layers_device, int32_t nlayers, int32_t nblocks, int32_t nwaves, int32_t max_dil, void* stream) {
    MH_REQUIRE(layers_device && nlayers > 0 && nblocks > 0, MH_ERR_ARG, "mh_wgrad_stream: empty layer table");
    MH_REQUIRE(nwaves >= 1 && nwaves <= 8, MH_ERR_ARG, "mh_wgrad_stream: 1 .. 8 waves per workgroup");
    hipStream_t s = (hipStream_t)stream;
    if (max_dil == -2) return stream_launch<5, 1, 2>(layers_device, nlayers, nblocks, nwaves, s, false);      // a table of stride-2 layers
    if (max_dil == -3) return stream_launch_mixed(layers_device, nlayers, nblocks, nwaves, s, false);          // stride-1 (dilation <= 8) and stride-2 layers
    MH_REQUIRE(max_dil >= 1 && max_dil <= 16, MH_ERR_UNSUPPORTED, "mh_wgrad_stream: dilation 1 .. 16 (or -2: a table of stride-2 layers, -3: mixed strides)");
    if (max_dil > 8) return stream_launch<4, 1>(layers_device, nlayers, nblocks, nwaves, s, false);
    int dist = g_stream_dist.load(std::memory_order_relaxed);
    if (dist <= 0) dist = nwaves <= 6 ? 2 : 1;                          // the deepest ring that fits 160 KB
    if (dist >= 2 && nwaves <= 6) return stream_launch<3, 2>(layers_device, nlayers, nblocks, nwaves, s, false);
    return stream_launch<3, 1>(layers_device, nlayers, nblocks, nwaves, s, false);
}

// this translation unit's copy of the deterministic-accumulation table (mh_common.h)
extern "C" int mh_det_sync_wgrad_stream(const void* t) { return mh_det_upload(*reinterpret_cast<const mh_det_table*>(t)); }
extern "C" int mh_det_ovf_wgrad_stream(void) { return mh_det_overflow_take(); }      // this translation unit's saturation flag of the deterministic twin (mh_common.h)
