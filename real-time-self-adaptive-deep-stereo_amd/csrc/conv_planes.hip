// conv_planes.hip -- the plane kernels (round 4).  Three kernels over activations kept as bf16 NHWC planes:
//   conv_planes_kernel        stride-1 'SAME' 3x3 (dilated) layers, whole reduction in one LDS patch: split-bf16 forward (PL = 2) and, with one plane and a
//                             mirrored / transposed bank, their input gradients and plain-bf16 forward layers (PL = 1)
//   conv_planes_ck_kernel     the same walk over reductions of more than 128 channels (DispNet): K-chunked, a loader wave streams 64-channel patch chunks
//                             through three LDS buffers
//   conv_planes_s2bwd_kernel  input gradient of the stride-2 3x3 layers: four parity classes = four small convolutions over one dz patch
// What follows describes the first; the other two have their own headers further down.
// -- split-bf16 ("bf16x3") forward pass of the stride-1 "SAME" 3x3 (dilated) layers from PRE-SPLIT operands.
// What is computed: tf.nn.conv2d / atrous_conv2d + bias_add + leaky (Nets/sharedLayers.py:54-77) for the estimator / context / pyramid layers
// of Nets/MadNet.py:73-171,173-249 -- the same three-MFMA products (lo*hi + hi*lo + hi*hi, fp32 accumulate) as conv_bank_kernel<..., X3>.
//
// Why a new kernel (profiles/r03_pmc_roofline.json): conv_bank_kernel stages fp32 activations through VGPRs and splits them
// into hi / lo bf16 in EVERY consumer workgroup (6.7 M VALU for 1.7 M MFMAs), feeds each 32x32 wave tile from 1 KB bank fragments that serve only
// 4 MFMAs (the K walk runs at the L2 -> L1 rate of the fragment loads, ~45 B/clk/CU) and writes fp32 + a bf16 shadow.  Here
//   * an activation lives in HBM as TWO bf16 NHWC planes, hi = bf16(x) and lo = bf16(x - hi) (pixel stride = channels rounded up to 32, padding
//     zero): the hi plane IS the shadow the streamed filter gradient and the input gradients already read; the producer's epilogue splits each
//     element once, the consumer never converts;
//   * the patch (tile + one-lattice-pixel halo, both planes) goes global -> LDS by LDS DMA (buffer_load_dwordx4 ... lds): no VGPR round trip, no
//     VALU, no ds_write; the LDS image is [patch row][patch column][2*K16 + 1 chunks of 16 B] -- the odd chunk count (one pad chunk per pixel, written
//     as zeros by out-of-range lanes) makes every ds_read_b128 lane group of the v_mfma_f32_32x32x16_bf16 A operand hit 16 distinct bank quads;
//   * a wave owns MBW stacked 32-pixel M-blocks x 32 output columns (v_mfma_f32_32x32x16_bf16, 16 accumulator registers per block): one 1 KB bank
//     fragment per plane feeds 3 * MBW MFMAs of 32 cycles -- at MBW = 4 the fragment stream is ~21 B/clk/CU;
//   * the weights come from a fragment bank in the 32x32x16 register image (mh_pack_weights, trans = 2) straight into registers, three steps ahead;
//     the K walk (9 taps x K16 steps of 16 channels) is fully unrolled: no barrier, no address arithmetic (every LDS offset is an immediate);
//   * the epilogue transposes the tile through LDS and stores 16 bytes per lane: hi plane, lo plane and (only where a non-plane consumer exists) fp32.
#include "conv_args.h"
#include <stdlib.h>
#include <algorithm>
#include <atomic>

// Bank layout rule (host and kernels agree by this function alone): reductions over more than 128 channels -- except 97..112 and 193..208, the
// split-bf16 iconv layers of DispNet, which have whole-K instances -- are K-chunked in chunks of MH_PLANES_KC16 * 16 channels.
#define MH_PLANES_KC16 4
#ifndef MH_PLANES_WHOLE_MAX
#define MH_PLANES_WHOLE_MAX 8
#endif
extern "C" int mh_planes_kc16(int32_t K) {
    const int k16 = (K + 15) / 16;
    return (k16 <= MH_PLANES_WHOLE_MAX || k16 == 13) ? 0 : MH_PLANES_KC16;
}

// walk step t = (tap t / K16, 16-channel step t % K16) -> its fragment's index in the bank: the bank is tap-major for whole-K layouts and chunk-major
// ([chunk of MH_PLANES_KC16 steps][tap][step in chunk], the reduction padded to whole chunks) beyond 128 channels (mh_planes_kc16) -- the stride-2 kernels walk
// tap-major either way (their patch holds the whole reduction) and pick the fragments where they lie.  Compile-time: the walks are fully unrolled.
template <int K16, int TAPS>
__host__ __device__ constexpr int planes_bank_step(int t) {
    constexpr bool chunked = !(K16 <= MH_PLANES_WHOLE_MAX || K16 == 13);
    const int tap = t / K16, s = t % K16;
    return chunked ? ((s / MH_PLANES_KC16) * TAPS + tap) * MH_PLANES_KC16 + (s % MH_PLANES_KC16) : t;
}

namespace {

struct PlanesArgs {
    const unsigned short* in_hi; const unsigned short* in_lo;     // bf16 planes [B][H][W][in_pld]
    const void* wb;                                               // fragment bank, 32x32x16 image: [(tap * K16 + s)][32-column tile][plane][lane][8 bf16]
    const float* bias;
    const unsigned short* mask_hi; int mask_pld; float mask_alpha;  // != null: out *= (mask > 0 ? 1 : mask_alpha), mask = a bf16 plane [pixel][mask_pld] (sign test: the
                                                                   // fused gradient of tf.maximum(alpha x, x) in an input-gradient launch)
    int mask_c0, mask_c1;                                          // ... for output columns in [mask_c0, mask_c1) only (a concat's member; whole row: 0, INT_MAX)
    int nchunks;                                                   // K-chunked instances: chunks of K16 * 16 reduction channels (1 for the whole-K instances)
    int Hin, Win;                                                  // stride-2 kernels: size of the INPUT of the walk (dz of the input gradient; x of the stride-2 forward); H, W = size of the result
    int pad_t, pad_l;                                              // stride-2 forward: TF 'SAME' padding in front ((k - 2) / 2 on even sizes)
    int acc_out;                                                   // 1: the fp32 result is ADDED to what `out` holds before the mask (an input gradient that is not the first contribution: (old + new) * mask)
    float* out; unsigned short* out_hi; unsigned short* out_lo;   // any of them may be null
    unsigned in_bytes, wb_bytes, out_bytes, outp_bytes;
    int in_pld, out_ld, out_pld;
    int B, H, W, K, N, dil;
    float alpha;
    int tiles_y, tiles_x, ntiles_n, nwg;
    mh_tile_decode dec;                                           // magic multipliers of the workgroup -> tile decode (mh_common.h)
    int dbg;                                                      // timing experiments (scripts/microbench.py phases): 1 = skip the K walk, 2 = skip the staging, 16 = no epilogue, 32 = no stores
};

// MC: columns of a 32-pixel M-block (32: one row of 32 lattice pixels; 16: two rows of 16).  WM x WN waves; a wave owns MBW M-blocks x 32 columns.
// K16: 16-channel steps per tap (K rounded up to 16).
template <int MC, int WM, int WN, int MBW, int K16, int PL = 2>
struct PlanesGeo {
    static constexpr int MR = 32 / MC;                 // rows of an M-block
    static constexpr int MBW_ = MBW;
    static constexpr int NW = WM * WN, NTH = NW * 64;
    static constexpr int TR = WM * MBW * MR;           // tile rows (lattice)
    static constexpr int BM = WM * MBW * 32, BN = WN * 32;
    static constexpr int PR = TR + 2, PC = MC + 2;     // patch rows / columns
    static constexpr int NCK = 2 * K16;                // data chunks (16 B = 8 channels) per patch pixel and plane
    static constexpr int NCK1 = NCK + 1;               // + one pad chunk: odd => conflict-free ds_read_b128 groups
    // two-row M-blocks: the second row must start a multiple of 16 chunks after the first (its 8 lanes of a 16-lane read group take the bank quads the
    // first row's 8 lanes leave free)
    static constexpr int ROWP = MR == 1 ? PC * NCK1 : ((PC * NCK1 + 15) / 16) * 16;
    static constexpr int PLANE_BLKS = (PR * ROWP * 16 + 1023) / 1024;
    static constexpr int PLANE_BYTES = PLANE_BLKS * 1024;
    static constexpr int CS = BN + 8;                  // (= 8 mod 16: the two half-waves of an accumulator write -- rows 4 apart -- hit disjoint bank halves)
    static constexpr int LDS_TILES = PL * PLANE_BYTES, LDS_CS = BM * CS * 4;
    static constexpr int LDS = LDS_TILES > LDS_CS ? LDS_TILES : LDS_CS;
};

// What the epilogue needs from memory, requested BEFORE the K walk (round 4: the eight conditional bias loads behind the walk's last barrier and the
// mask loads inside the store loop each exposed a MALL / HBM round trip per launch -- ~1 us of a 20 - 30 us layer, 33 launches per step): the thread's
// 8 bias values (two 16-byte buffer loads, out-of-range columns read 0) and, in the one-plane form, the mask chunks of every pixel row it will store.
template <class G, int PL>
struct PlanesEpiPre {
    static constexpr int ITER = (G::BM * (G::BN / 8) + G::NTH - 1) / G::NTH;
    f32x4 b0, b1;
    u32x4 mk[PL == 1 ? ITER : 1];
};
template <class G, int PL>
__device__ __forceinline__ void planes_epilogue_prefetch(const PlanesArgs& p, int tid, int n0, int y00, int x00, int b, int d, PlanesEpiPre<G, PL>& pre) {
    constexpr int MR = G::MR, NTH = G::NTH, BM = G::BM, C8 = G::BN / 8, RP = NTH / C8;
    const int n = n0 + (tid % C8) * 8;
    const __amdgpu_buffer_rsrc_t rs_b = mh_make_rsrc(p.bias ? (const void*)p.bias : (const void*)p.in_hi, p.bias ? (unsigned)(p.N * 4) : 0u);
    pre.b0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, n * 4, 0, 0));
    pre.b1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, n * 4 + 16, 0, 0));
    if constexpr (PL == 1) {
        if (p.mask_hi && tid < NTH) {
            const __amdgpu_buffer_rsrc_t rs_mk = mh_make_rsrc((const void*)p.mask_hi, (unsigned)((int64_t)p.B * p.H * p.W * p.mask_pld * 2));
#pragma unroll
            for (int it = 0; it < PlanesEpiPre<G, PL>::ITER; ++it) {
                const int m = tid / C8 + it * RP;
                const int blk = m >> 5, w = m & 31;
                const int row = MR == 1 ? blk : blk * 2 + (w >> 4), colp = MR == 1 ? w : (w & 15);
                const int y = y00 + row * d, x = x00 + colp * d;
                const bool ok = m < BM && y < p.H && x < p.W && n < p.N;
                pre.mk[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_mk, ok ? (((b * p.H + y) * p.W + x) * p.mask_pld + n) * 2 : MH_OOB, 0, 0);
            }
        }
    }
}

// the epilogue of both kernels: accumulators -> LDS [pixel][column], then 8 consecutive columns per lane: bias, leaky, mask, split, 16-byte stores.
// `active`: the calling thread belongs to a compute wave (the K-chunked kernel's loader wave only keeps the barrier company); NTH = compute threads.
template <class G, int PL>
__device__ __forceinline__ void planes_epilogue(const PlanesArgs& p, float* smem_all, const f32x16 (&acc)[G::MBW_], int tid, bool active, int wm, int wn, int lane,
                                                int n0, int y00, int x00, int b, int d, const PlanesEpiPre<G, PL>& pre) {
    constexpr int MR = G::MR, NTH = G::NTH, BM = G::BM, BN = G::BN, CS = G::CS, MBW = G::MBW_;
    // ---- epilogue: accumulators -> LDS [pixel][column], then 8 consecutive columns per lane: bias, leaky, split, 16-byte stores ----------------
    float* const Cs = smem_all;
    if (p.dbg & 16) return;                   // timing experiment (mh_tune_conv_planes bit 12): no epilogue at all
    if (active) {
        const int col = wn * 32 + (lane & 31);
#pragma unroll
        for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Cs[((wm * MBW + mb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * CS + col] = acc[mb][r];
    }
    __syncthreads();
    if (!active) return;
    constexpr int C8 = BN / 8;
    static_assert(NTH % C8 == 0, "a thread keeps its 8-column group");
    constexpr int RP = NTH / C8;
    const int c8 = tid % C8;
    const int n = n0 + c8 * 8;
    const float bv[8] = {pre.b0[0], pre.b0[1], pre.b0[2], pre.b0[3], pre.b1[0], pre.b1[1], pre.b1[2], pre.b1[3]};
    const __amdgpu_buffer_rsrc_t rs_o = mh_make_rsrc(p.out ? p.out : (float*)p.out_hi, p.out ? p.out_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rs_oh = mh_make_rsrc(p.out_hi ? p.out_hi : (unsigned short*)p.out, p.out_hi ? p.outp_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rs_ol = mh_make_rsrc(p.out_lo ? p.out_lo : (unsigned short*)p.out, p.out_lo ? p.outp_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rs_mk = mh_make_rsrc(p.mask_hi ? (const void*)p.mask_hi : (const void*)p.in_hi, p.mask_hi ? (unsigned)((int64_t)p.B * p.H * p.W * p.mask_pld * 2) : 0u);
#pragma unroll
    for (int it = 0; it < PlanesEpiPre<G, PL>::ITER; ++it) {
        const int m = tid / C8 + it * RP;
        if (m >= BM) break;
        const int blk = m >> 5, w = m & 31;
        const int row = MR == 1 ? blk : blk * 2 + (w >> 4), colp = MR == 1 ? w : (w & 15);
        const int y = y00 + row * d, x = x00 + colp * d;
        const bool ok = y < p.H && x < p.W && n < p.N;
        const int pix = (b * p.H + y) * p.W + x;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(&Cs[m * CS + c8 * 8]), v1 = *reinterpret_cast<const f32x4*>(&Cs[m * CS + c8 * 8 + 4]);
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[e] += bv[e];
            if (p.alpha != 1.0f) v[e] = v[e] > 0.f ? v[e] : p.alpha * v[e];
        }
        if (p.acc_out) {        // (uniform) earlier contributions to this gradient map: read where the result goes
            const int ofa = ok ? (pix * p.out_ld + n) * 4 : MH_OOB;
            const f32x4 o0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_o, ofa, 0, 0));
            const f32x4 o1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_o, ofa == MH_OOB ? MH_OOB : ofa + 16, 0, 0));
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += o0[e]; v[4 + e] += o1[e]; }
        }
        if (p.mask_hi) {        // 8 bf16 of the activation's hi plane: bf16 keeps sign and zero, the test is that of the fp32 tensor
            u32x4 mq;
            if constexpr (PL == 1) mq = pre.mk[it];
            else mq = __builtin_amdgcn_raw_buffer_load_b128(rs_mk, ok ? (pix * p.mask_pld + n) * 2 : MH_OOB, 0, 0);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float mk = __builtin_bit_cast(float, (e & 1) ? (mq[e >> 1] & 0xffff0000u) : (mq[e >> 1] << 16));
                v[e] *= (mk > 0.f || n + e < p.mask_c0 || n + e >= p.mask_c1) ? 1.0f : p.mask_alpha;
            }
        }
        unsigned hh[4], ll[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) mh_split_bf16x2(v[2 * e], v[2 * e + 1], hh[e], ll[e]);
        const bool st = ok && !(p.dbg & 32);      // timing experiment (bit 13): the whole epilogue but its stores
        const int op = st ? (pix * p.out_pld + n) * 2 : MH_OOB;
        const int of = st ? (pix * p.out_ld + n) * 4 : MH_OOB;
        const u32x4 qh = {hh[0], hh[1], hh[2], hh[3]}, ql = {ll[0], ll[1], ll[2], ll[3]};
        const u32x4 f0 = __builtin_bit_cast(u32x4, make_float4(v[0], v[1], v[2], v[3])), f1 = __builtin_bit_cast(u32x4, make_float4(v[4], v[5], v[6], v[7]));
        const int of1 = of == MH_OOB ? MH_OOB : of + 16;
        // (write-through and non-temporal stores were measured: +11 / +5 us per step, profiles/r04_experiments.txt #6)
        __builtin_amdgcn_raw_buffer_store_b128(qh, rs_oh, op, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(ql, rs_ol, op, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(f0, rs_o, of, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(f1, rs_o, of1, 0, 0);
    }
}

// PL = 2: split-bf16 (hi + lo planes of both operands, three MFMAs per product: the forward layers).  PL = 1: plain bf16 from the hi plane and a one-plane
// bank (one MFMA per product): the INPUT GRADIENTS of the same layers -- a 'SAME' 3x3 input gradient is this forward walk over dz with the taps mirrored
// and the bank transposed, which mh_pack_weights (trans = 3) bakes into the bank, so the kernel does not know the difference.
template <int MC, int WM, int WN, int MBW, int K16, int PL>
__global__ __launch_bounds__(WM * WN * 64) void conv_planes_kernel(PlanesArgs p) {
    using G = PlanesGeo<MC, WM, WN, MBW, K16, PL>;
    constexpr int MR = G::MR, NW = G::NW, NTH = G::NTH, TR = G::TR, BM = G::BM, BN = G::BN, PR = G::PR, PC = G::PC;
    constexpr int NCK = G::NCK, NCK1 = G::NCK1, ROWP = G::ROWP, PLANE_BLKS = G::PLANE_BLKS, PLANE_BYTES = G::PLANE_BYTES, CS = G::CS;
    HIP_DYNAMIC_SHARED(float, smem_all)
    unsigned char* const smem = reinterpret_cast<unsigned char*>(smem_all);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int d = p.dil;

    if (p.dbg & 128) mh_setprio<2>();        // experiment (mh_tune_conv_planes bit 15): the main lane's waves ahead of co-resident filter-gradient waves at the issue arbiters
    const int lin = mh_xcd_remap(blockIdx.x, p.nwg);
    int tile_n, ttx, tty, cx, cy, b;
    mh_decode_tile(lin, p.dec, tile_n, ttx, tty, cx, cy, b);
    const int n0 = tile_n * BN;
    const int y00 = cy + d * (tty * TR), x00 = cx + d * (ttx * MC);       // image position of tile pixel (0, 0)
    PlanesEpiPre<G, PL> pre;
    planes_epilogue_prefetch<G, PL>(p, tid, n0, y00, x00, b, d, pre);

    // ---- weight fragments: ring of NSTB steps, PF steps ahead (ordinary loads: hipcc counts them) -----------------------------------
    // weight-fragment ring: in the replayed step the banks come from MALL / HBM (everything else the step touches has passed through the L2 since the
    // previous replay), so three steps in flight (1150 cycles at MBW = 4, PL = 2) do not cover the latency: eight (r04: step 1.525 -> 1.481 ms; twelve: the same)
#ifndef MH_PLANES_RING2
#define MH_PLANES_RING2 8
#endif
#ifndef MH_PLANES_RING1
#define MH_PLANES_RING1 8
#endif
    constexpr int T = 9 * K16, NSTB = PL == 2 ? MH_PLANES_RING2 : MH_PLANES_RING1, PF = NSTB - 1;
    const __amdgpu_buffer_rsrc_t rs_b = mh_make_rsrc(p.wb, p.wb_bytes);
    const int nt32 = (p.N + 31) >> 5;
    const int nt = tile_n * WN + wn;
    const int voff_b = nt < nt32 ? nt * (PL * 1024) + lane * 16 : MH_OOB;
    const int step_b = nt32 * (PL * 1024);
    u32x4 fb[NSTB][PL];
    auto issue_b = [&](int t, int slot) {
        if (t < T) {
            fb[slot][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, voff_b, t * step_b, 0);
            if constexpr (PL == 2) fb[slot][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, voff_b, t * step_b + 1024, 0);
        }
    };
#pragma unroll
    for (int t = 0; t < PF; ++t) issue_b(t, t % NSTB);

    // ---- stage the patch: LDS DMA, both planes; chunk g of a plane <- (patch row, patch column, channel chunk) ------------------------------
    if (!(p.dbg & 2)) {
        const mh_dma_src rs_h = mh_make_dma_src(p.in_hi, p.in_bytes), rs_l = mh_make_dma_src(p.in_lo, p.in_bytes);
        const int pix_b = p.in_pld * 2;
        for (int i = wave; i < PLANE_BLKS; i += NW) {
            const int g = i * 64 + lane;
            const int pr = g / ROWP, rem = g - pr * ROWP;
            const int pc = rem / NCK1, c = rem - pc * NCK1;
            const int iy = y00 + (pr - 1) * d, ix = x00 + (pc - 1) * d;
            const bool ok = pr < PR && pc < PC && c < NCK && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const int off = ok ? ((b * p.H + iy) * p.W + ix) * pix_b + c * 16 : MH_OOB;
            mh_glds16(rs_h, smem + i * 1024, off);
            if constexpr (PL == 2) mh_glds16(rs_l, smem + PLANE_BYTES + i * 1024, off);
        }
    }
    MH_WAIT_VMCNT(0);
    __syncthreads();

    f32x16 acc[MBW];
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;

    // ---- K walk: step t = (tap t / K16, channels 16 (t % K16) .. +15); A fragments one step ahead (two register sets) ---------------------
    {
        const int j = lane & 31, kg = lane >> 5;
        const int lr = MR == 1 ? 0 : (j >> 4), lc = MR == 1 ? j : (j & 15);
        const unsigned char* const a_h = smem + (((wm * MBW * MR + lr) * ROWP + lc * NCK1 + kg) * 16);
        const unsigned char* const a_l = a_h + PLANE_BYTES;
        u32x4 fa[2][MBW][PL];
        auto issue_a = [&](int t, int set) {
            if (t < T) {
                const int tap = t / K16, s = t - tap * K16;
                const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
                for (int mb = 0; mb < MBW; ++mb) {
                    const int imm = (((mb * MR + ky) * ROWP + kx * NCK1 + 2 * s) * 16);
                    fa[set][mb][0] = *reinterpret_cast<const u32x4*>(a_h + imm);
                    if constexpr (PL == 2) fa[set][mb][1] = *reinterpret_cast<const u32x4*>(a_l + imm);
                }
            }
        };
        issue_a(0, 0);
        if (!(p.dbg & 1)) {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int sa = t & 1, sb = t % NSTB;
                // lo(A)*hi(B), hi(A)*lo(B), hi(A)*hi(B) -- term outermost: consecutive MFMAs hit different accumulators
#pragma unroll
                for (int term = (PL == 2 ? 0 : 2); term < 3; ++term) {
#pragma unroll
                    for (int mb = 0; mb < MBW; ++mb) {
                        acc[mb] = mh_mfma_bf16_32(fa[sa][mb][term == 0 ? PL - 1 : 0], fb[sb][term == 1 ? PL - 1 : 0], acc[mb]);
                        if (term == (PL == 2 ? 0 : 2) && mb == 0) issue_a(t + 1, sa ^ 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                issue_b(t + PF, (t + PF) % NSTB);          // slot of step t - 1: free
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    __syncthreads();                                 // every wave is done with the patch: the accumulator tile goes over it

    planes_epilogue<G, PL>(p, smem_all, acc, tid, true, wm, wn, lane, n0, y00, x00, b, d, pre);
}

// ---- staggered variant (round 6): the 128 x 128 tile as TWO wave groups of 64 pixels each, out of phase ------------------------------------------------
// One tile per CU means staging -> walk -> epilogue of a launch cannot overlap across tiles, and two co-resident 64-pixel workgroups run IN phase (r04 #5).  Here the
// two halves are wave groups of ONE workgroup (2 x WN waves, two per SIMD) that share the staged patch and part company behind the staging barrier:
//   * group 0 raises its issue priority (s_setprio) for the walk: it takes the MFMA pipe whenever it has an MFMA ready, group 1 fills the gaps -- group 0 ends its walk
//     first, and its epilogue (VALU, LDS, stores) runs in the shadow of group 1's remaining MFMAs; what stays exposed is ONE half-tile epilogue;
//   * no barrier behind the staging: the epilogue transposes per WAVE through a private 32 x 32 LDS block (wave-local: LDS operations of a wave execute in order), one
//     M-block at a time, and stores the wave's own 32 columns (64-byte runs per pixel in the planes, 128 in the fp32 map); the patch is never overwritten.
// Cost: each half walks the whole bank, so the weight-fragment stream per MFMA doubles (MBW = 2: ~43 B/clk/CU).
template <int MC, int WN, int MBW, int K16, int PL>
struct PlanesStgGeo : PlanesGeo<MC, 2, WN, MBW, K16, PL> {
    using B = PlanesGeo<MC, 2, WN, MBW, K16, PL>;
    static constexpr int TW = 40;                                   // floats per pixel row of a wave's transposition block (= 8 mod 16: the half-waves of an accumulator write hit disjoint bank halves)
    static constexpr int WBUF = 32 * TW * 4;                        // bytes per wave
    static constexpr int LDS_STG = B::LDS_TILES + B::NW * WBUF;
};

template <class G, int PL>
struct PlanesStgPre {
    f32x4 b0, b1;
    u32x4 mk[PL == 1 ? 2 * G::MBW_ : 1];
};

// pixel p (0 .. 31) of M-block `blk` of the tile -> image position
template <class G>
__device__ __forceinline__ void planes_stg_pixel(int blk, int w, int y00, int x00, int d, int& y, int& x) {
    constexpr int MR = G::MR;
    const int row = MR == 1 ? blk : blk * 2 + (w >> 4), colp = MR == 1 ? w : (w & 15);
    y = y00 + row * d; x = x00 + colp * d;
}

template <class G, int PL>
__device__ __forceinline__ void planes_stg_prefetch(const PlanesArgs& p, int wm, int wn, int lane, int n0, int y00, int x00, int b, int d, PlanesStgPre<G, PL>& pre) {
    constexpr int MBW = G::MBW_;
    const int n = n0 + wn * 32 + (lane & 3) * 8;
    const __amdgpu_buffer_rsrc_t rs_b = mh_make_rsrc(p.bias ? (const void*)p.bias : (const void*)p.in_hi, p.bias ? (unsigned)(p.N * 4) : 0u);
    pre.b0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, n * 4, 0, 0));
    pre.b1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, n * 4 + 16, 0, 0));
    if constexpr (PL == 1) {
        if (p.mask_hi) {
            const __amdgpu_buffer_rsrc_t rs_mk = mh_make_rsrc((const void*)p.mask_hi, (unsigned)((int64_t)p.B * p.H * p.W * p.mask_pld * 2));
#pragma unroll
            for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) {
                    int y, x;
                    planes_stg_pixel<G>(wm * MBW + mb, ps * 16 + (lane >> 2), y00, x00, d, y, x);
                    const bool ok = y < p.H && x < p.W && n < p.N;
                    pre.mk[mb * 2 + ps] = __builtin_amdgcn_raw_buffer_load_b128(rs_mk, ok ? (((b * p.H + y) * p.W + x) * p.mask_pld + n) * 2 : MH_OOB, 0, 0);
                }
        }
    }
}

// a wave's own epilogue: per M-block, accumulators -> the wave's LDS block [pixel][column] -> 8 consecutive columns of 16 pixels per pass: bias, leaky, mask, split, 16-byte stores
template <class G, int PL>
__device__ __forceinline__ void planes_stg_epilogue(const PlanesArgs& p, float* wb, const f32x16 (&acc)[G::MBW_], int wm, int wn, int lane, int n0, int y00, int x00, int b, int d,
                                                    const PlanesStgPre<G, PL>& pre) {
    constexpr int MBW = G::MBW_, TW = G::TW;
    if (p.dbg & 16) return;
    const int c8 = lane & 3;
    const int n = n0 + wn * 32 + c8 * 8;
    const float bv[8] = {pre.b0[0], pre.b0[1], pre.b0[2], pre.b0[3], pre.b1[0], pre.b1[1], pre.b1[2], pre.b1[3]};
    const __amdgpu_buffer_rsrc_t rs_o = mh_make_rsrc(p.out ? p.out : (float*)p.out_hi, p.out ? p.out_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rs_oh = mh_make_rsrc(p.out_hi ? p.out_hi : (unsigned short*)p.out, p.out_hi ? p.outp_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rs_ol = mh_make_rsrc(p.out_lo ? p.out_lo : (unsigned short*)p.out, p.out_lo ? p.outp_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rs_mk = mh_make_rsrc(p.mask_hi ? (const void*)p.mask_hi : (const void*)p.in_hi, p.mask_hi ? (unsigned)((int64_t)p.B * p.H * p.W * p.mask_pld * 2) : 0u);
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb) {
        mh_wave_sync();                            // the previous block's reads are behind us
#pragma unroll
        for (int r = 0; r < 16; ++r) wb[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * TW + (lane & 31)] = acc[mb][r];
        mh_wave_sync();
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int w = ps * 16 + (lane >> 2);
            int y, x;
            planes_stg_pixel<G>(wm * MBW + mb, w, y00, x00, d, y, x);
            const bool ok = y < p.H && x < p.W && n < p.N;
            const int pix = (b * p.H + y) * p.W + x;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(&wb[w * TW + c8 * 8]), v1 = *reinterpret_cast<const f32x4*>(&wb[w * TW + c8 * 8 + 4]);
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] += bv[e];
                if (p.alpha != 1.0f) v[e] = v[e] > 0.f ? v[e] : p.alpha * v[e];
            }
            if (p.acc_out) {
                const int ofa = ok ? (pix * p.out_ld + n) * 4 : MH_OOB;
                const f32x4 o0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_o, ofa, 0, 0));
                const f32x4 o1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_o, ofa == MH_OOB ? MH_OOB : ofa + 16, 0, 0));
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += o0[e]; v[4 + e] += o1[e]; }
            }
            if (p.mask_hi) {
                u32x4 mq;
                if constexpr (PL == 1) mq = pre.mk[mb * 2 + ps];
                else mq = __builtin_amdgcn_raw_buffer_load_b128(rs_mk, ok ? (pix * p.mask_pld + n) * 2 : MH_OOB, 0, 0);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float mk = __builtin_bit_cast(float, (e & 1) ? (mq[e >> 1] & 0xffff0000u) : (mq[e >> 1] << 16));
                    v[e] *= (mk > 0.f || n + e < p.mask_c0 || n + e >= p.mask_c1) ? 1.0f : p.mask_alpha;
                }
            }
            unsigned hh[4], ll[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) mh_split_bf16x2(v[2 * e], v[2 * e + 1], hh[e], ll[e]);
            const bool st = ok && !(p.dbg & 32);
            const int op = st ? (pix * p.out_pld + n) * 2 : MH_OOB;
            const int of = st ? (pix * p.out_ld + n) * 4 : MH_OOB;
            const u32x4 qh = {hh[0], hh[1], hh[2], hh[3]}, ql = {ll[0], ll[1], ll[2], ll[3]};
            const u32x4 f0 = __builtin_bit_cast(u32x4, make_float4(v[0], v[1], v[2], v[3])), f1 = __builtin_bit_cast(u32x4, make_float4(v[4], v[5], v[6], v[7]));
            const int of1 = of == MH_OOB ? MH_OOB : of + 16;
            __builtin_amdgcn_raw_buffer_store_b128(qh, rs_oh, op, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(ql, rs_ol, op, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(f0, rs_o, of, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(f1, rs_o, of1, 0, 0);
        }
    }
}

template <int MC, int WN, int MBW, int K16, int PL>
__global__ __launch_bounds__(2 * WN * 64) void conv_planes_kernel_stg(PlanesArgs p) {
    using G = PlanesStgGeo<MC, WN, MBW, K16, PL>;
    constexpr int MR = G::MR, NW = G::NW, TR = G::TR, BN = G::BN, PR = G::PR, PC = G::PC;
    constexpr int NCK = G::NCK, NCK1 = G::NCK1, ROWP = G::ROWP, PLANE_BLKS = G::PLANE_BLKS, PLANE_BYTES = G::PLANE_BYTES;
    HIP_DYNAMIC_SHARED(float, smem_all)
    unsigned char* const smem = reinterpret_cast<unsigned char*>(smem_all);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int d = p.dil;

    const int lin = mh_xcd_remap(blockIdx.x, p.nwg);
    int tile_n, ttx, tty, cx, cy, b;
    mh_decode_tile(lin, p.dec, tile_n, ttx, tty, cx, cy, b);
    const int n0 = tile_n * BN;
    const int y00 = cy + d * (tty * TR), x00 = cx + d * (ttx * MC);
    PlanesStgPre<G, PL> pre;
    planes_stg_prefetch<G, PL>(p, wm, wn, lane, n0, y00, x00, b, d, pre);

    constexpr int T = 9 * K16, NSTB = PL == 2 ? MH_PLANES_RING2 : MH_PLANES_RING1, PF = NSTB - 1;
    const __amdgpu_buffer_rsrc_t rs_b = mh_make_rsrc(p.wb, p.wb_bytes);
    const int nt32 = (p.N + 31) >> 5;
    const int nt = tile_n * WN + wn;
    const int voff_b = nt < nt32 ? nt * (PL * 1024) + lane * 16 : MH_OOB;
    const int step_b = nt32 * (PL * 1024);
    u32x4 fb[NSTB][PL];
    auto issue_b = [&](int t, int slot) {
        if (t < T) {
            fb[slot][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, voff_b, t * step_b, 0);
            if constexpr (PL == 2) fb[slot][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, voff_b, t * step_b + 1024, 0);
        }
    };
#pragma unroll
    for (int t = 0; t < PF; ++t) issue_b(t, t % NSTB);

    if (!(p.dbg & 2)) {
        const mh_dma_src rs_h = mh_make_dma_src(p.in_hi, p.in_bytes), rs_l = mh_make_dma_src(p.in_lo, p.in_bytes);
        const int pix_b = p.in_pld * 2;
        for (int i = wave; i < PLANE_BLKS; i += NW) {
            const int g = i * 64 + lane;
            const int pr = g / ROWP, rem = g - pr * ROWP;
            const int pc = rem / NCK1, c = rem - pc * NCK1;
            const int iy = y00 + (pr - 1) * d, ix = x00 + (pc - 1) * d;
            const bool ok = pr < PR && pc < PC && c < NCK && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const int off = ok ? ((b * p.H + iy) * p.W + ix) * pix_b + c * 16 : MH_OOB;
            mh_glds16(rs_h, smem + i * 1024, off);
            if constexpr (PL == 2) mh_glds16(rs_l, smem + PLANE_BYTES + i * 1024, off);
        }
    }
    MH_WAIT_VMCNT(0);
    __syncthreads();                                 // the only barrier of the kernel
    if (wm == 0 && !(p.dbg & 64)) mh_setprio<3>();

    f32x16 acc[MBW];
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
    {
        const int j = lane & 31, kg = lane >> 5;
        const int lr = MR == 1 ? 0 : (j >> 4), lc = MR == 1 ? j : (j & 15);
        const unsigned char* const a_h = smem + (((wm * MBW * MR + lr) * ROWP + lc * NCK1 + kg) * 16);
        const unsigned char* const a_l = a_h + PLANE_BYTES;
        u32x4 fa[2][MBW][PL];
        auto issue_a = [&](int t, int set) {
            if (t < T) {
                const int tap = t / K16, s = t - tap * K16;
                const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
                for (int mb = 0; mb < MBW; ++mb) {
                    const int imm = (((mb * MR + ky) * ROWP + kx * NCK1 + 2 * s) * 16);
                    fa[set][mb][0] = *reinterpret_cast<const u32x4*>(a_h + imm);
                    if constexpr (PL == 2) fa[set][mb][1] = *reinterpret_cast<const u32x4*>(a_l + imm);
                }
            }
        };
        issue_a(0, 0);
        if (!(p.dbg & 1)) {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int sa = t & 1, sb = t % NSTB;
#pragma unroll
                for (int term = (PL == 2 ? 0 : 2); term < 3; ++term) {
#pragma unroll
                    for (int mb = 0; mb < MBW; ++mb) {
                        acc[mb] = mh_mfma_bf16_32(fa[sa][mb][term == 0 ? PL - 1 : 0], fb[sb][term == 1 ? PL - 1 : 0], acc[mb]);
                        if (term == (PL == 2 ? 0 : 2) && mb == 0) issue_a(t + 1, sa ^ 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                issue_b(t + PF, (t + PF) % NSTB);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (wm == 0) mh_setprio<0>();                   // its epilogue yields to the other group's walk
    planes_stg_epilogue<G, PL>(p, reinterpret_cast<float*>(smem + G::LDS_TILES + wave * G::WBUF), acc, wm, wn, lane, n0, y00, x00, b, d, pre);
}

// ---- K-chunked variant (round 4, DispNet's 256 .. 1056-channel layers: Nets/DispNet.py:75-152) -------------------------------------------------
// The reduction runs over `nchunks` chunks of K16 * 16 channels.  The patch of ONE chunk (both planes) is an LDS buffer; NBUF buffers rotate: a
// dedicated LOADER wave (the last wave of the workgroup, no MFMA) issues the LDS DMA of chunk c + NBUF - 1 while the compute waves walk chunk c and
// waits only for chunk c + 1 (s_waitcnt vmcnt(<loads of the newest chunk>)): its own vmcnt counter, so the compute waves' fragment loads never queue
// behind the DMA.  One barrier per chunk.  The fragment bank is chunk-major ([chunk][tap][step]: mh_pack_weights kc16), so the weight ring runs
// straight across the chunk boundaries (T = 9 K16 is a multiple of the ring length for K16 = 4 / 8).
template <int MC, int WM, int WN, int MBW, int K16, int PL, int NBUF>
struct PlanesCkGeo : PlanesGeo<MC, WM, WN, MBW, K16, PL> {
    using B = PlanesGeo<MC, WM, WN, MBW, K16, PL>;
    static constexpr int BUF_BYTES = PL * B::PLANE_BYTES;
    static constexpr int NLOAD = PL * B::PLANE_BLKS;                 // DMA instructions of one chunk (all by the loader wave)
    static constexpr int LDS_CK = NBUF * BUF_BYTES > B::LDS_CS ? NBUF * BUF_BYTES : B::LDS_CS;
    static_assert(NLOAD <= 63, "the loader's vmcnt window");
    static_assert((9 * K16) % 4 == 0, "the weight ring must close on a chunk boundary");
};

template <int MC, int WM, int WN, int MBW, int K16, int PL, int NBUF>
__global__ __launch_bounds__((WM * WN + 1) * 64) void conv_planes_ck_kernel(PlanesArgs p) {
    using G = PlanesCkGeo<MC, WM, WN, MBW, K16, PL, NBUF>;
    constexpr int MR = G::MR, NW = G::NW, TR = G::TR, BN = G::BN, PR = G::PR, PC = G::PC;
    constexpr int NCK = G::NCK, NCK1 = G::NCK1, ROWP = G::ROWP, PLANE_BLKS = G::PLANE_BLKS, PLANE_BYTES = G::PLANE_BYTES, BUF_BYTES = G::BUF_BYTES;
    HIP_DYNAMIC_SHARED(float, smem_all)
    unsigned char* const smem = reinterpret_cast<unsigned char*>(smem_all);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave == NW;
    const int wm = loader ? 0 : wave / WN, wn = loader ? 0 : wave % WN;
    const int d = p.dil;
    const int nch = p.nchunks;

    const int lin = mh_xcd_remap(blockIdx.x, p.nwg);
    int tile_n, ttx, tty, cx, cy, b;
    mh_decode_tile(lin, p.dec, tile_n, ttx, tty, cx, cy, b);
    const int n0 = tile_n * BN;
    const int y00 = cy + d * (tty * TR), x00 = cx + d * (ttx * MC);
    PlanesEpiPre<G, PL> pre;
    planes_epilogue_prefetch<G, PL>(p, tid, n0, y00, x00, b, d, pre);

    // weight ring: the one-plane (plain bf16) walk spends 32 MBW cycles per step and DispNet's banks (up to 19 MB) come from HBM, not L2: eleven steps
    // in flight (44 VGPRs) instead of three (microbenchmark r04: conv4_1 42 us with three)
    constexpr int T = 9 * K16, NSTB = PL == 1 ? 12 : 4, PF = NSTB - 1;
    static_assert(T % NSTB == 0, "the weight ring must close on a chunk boundary");
    f32x16 acc[MBW];
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;

    if (loader) {
        // ---- the loader wave: chunk c -> buffer c % NBUF, NBUF - 1 chunks ahead of the walk ------------------------------------------------------
        const mh_dma_src rs_h = mh_make_dma_src(p.in_hi, p.in_bytes), rs_l = mh_make_dma_src(PL == 2 ? p.in_lo : p.in_hi, p.in_bytes);
        const int pix_b = p.in_pld * 2;
        const int pld8 = p.in_pld >> 3;                       // 16-byte chunks of a plane pixel
        // per 1 KB block of a plane buffer: the lane's patch pixel offset and channel chunk (chunk-independent), computed once
        int base[PLANE_BLKS], cc[PLANE_BLKS];
#pragma unroll
        for (int i = 0; i < PLANE_BLKS; ++i) {
            const int g = i * 64 + lane;
            const int pr = g / ROWP, rem = g - pr * ROWP;
            const int pc = rem / NCK1, c1 = rem - pc * NCK1;
            const int iy = y00 + (pr - 1) * d, ix = x00 + (pc - 1) * d;
            const bool ok = pr < PR && pc < PC && c1 < NCK && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            base[i] = ok ? ((b * p.H + iy) * p.W + ix) * pix_b + c1 * 16 : MH_OOB;
            cc[i] = ok ? c1 : (1 << 28);
        }
        auto stage = [&](int c) {
            unsigned char* const dst = smem + (c % NBUF) * BUF_BYTES;
            const int lim = c < nch ? pld8 - c * NCK : 0;       // channel chunks of this K chunk inside the plane row (chunks past the end: none)
#pragma unroll
            for (int i = 0; i < PLANE_BLKS; ++i) {
                const int off = cc[i] < lim ? base[i] + c * (NCK * 16) : MH_OOB;
                mh_glds16(rs_h, dst + i * 1024, off);
                if constexpr (PL == 2) mh_glds16(rs_l, dst + PLANE_BYTES + i * 1024, off);
            }
        };
        // every chunk slot issues exactly NLOAD loads (chunks past the end: all lanes out of range -> zeros into a buffer nobody reads), so the
        // vmcnt window is a constant
        for (int c = 0; c < NBUF - 1; ++c) stage(c);
        for (int c = 0; c < nch; ++c) {
            // chunk c must have landed: all but the loads of the NBUF - 2 newer chunks in flight
            if constexpr (NBUF >= 3) MH_WAIT_VMCNT((NBUF - 2) * G::NLOAD <= 63 ? (NBUF - 2) * G::NLOAD : 63);
            else MH_WAIT_VMCNT(0);
            __syncthreads();                                  // chunk c is visible; buffer (c - 1) % NBUF is free
            stage(c + NBUF - 1);
        }
        MH_WAIT_VMCNT(0);
        __syncthreads();                                      // the walk's last barrier
    } else {
        // ---- compute waves ------------------------------------------------------------------------------------------------------------------------
        const __amdgpu_buffer_rsrc_t rs_b = mh_make_rsrc(p.wb, p.wb_bytes);
        const int nt32 = (p.N + 31) >> 5;
        const int nt = tile_n * WN + wn;
        const int voff_b = nt < nt32 ? nt * (PL * 1024) + lane * 16 : MH_OOB;
        const int step_b = nt32 * (PL * 1024);
        const int t_all = nch * T;
        u32x4 fb[NSTB][PL];
        auto issue_b = [&](int gt, int slot) {
            if (gt < t_all) {
                fb[slot][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, voff_b, gt * step_b, 0);
                if constexpr (PL == 2) fb[slot][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, voff_b, gt * step_b + 1024, 0);
            }
        };
#pragma unroll
        for (int t = 0; t < PF; ++t) issue_b(t, t % NSTB);
        const int j = lane & 31, kg = lane >> 5;
        const int lr = MR == 1 ? 0 : (j >> 4), lc = MR == 1 ? j : (j & 15);
        const int a_off = (((wm * MBW * MR + lr) * ROWP + lc * NCK1 + kg) * 16);
        for (int c = 0; c < nch; ++c) {
            __syncthreads();                                  // chunk c has landed
            const unsigned char* const a_h = smem + (c % NBUF) * BUF_BYTES + a_off;
            const unsigned char* const a_l = a_h + PLANE_BYTES;
            u32x4 fa[2][MBW][PL];
            auto issue_a = [&](int t, int set) {
                if (t < T) {
                    const int tap = t / K16, s = t - tap * K16;
                    const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
                    for (int mb = 0; mb < MBW; ++mb) {
                        const int imm = (((mb * MR + ky) * ROWP + kx * NCK1 + 2 * s) * 16);
                        fa[set][mb][0] = *reinterpret_cast<const u32x4*>(a_h + imm);
                        if constexpr (PL == 2) fa[set][mb][1] = *reinterpret_cast<const u32x4*>(a_l + imm);
                    }
                }
            };
            issue_a(0, 0);
            const int gt0 = c * T;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int sa = t & 1, sb = t % NSTB;
#pragma unroll
                for (int term = (PL == 2 ? 0 : 2); term < 3; ++term) {
#pragma unroll
                    for (int mb = 0; mb < MBW; ++mb) {
                        acc[mb] = mh_mfma_bf16_32(fa[sa][mb][term == 0 ? PL - 1 : 0], fb[sb][term == 1 ? PL - 1 : 0], acc[mb]);
                        if (term == (PL == 2 ? 0 : 2) && mb == 0) issue_a(t + 1, sa ^ 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                issue_b(gt0 + t + PF, (t + PF) % NSTB);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();                                      // every wave is done with the patch buffers: the accumulator tile goes over them
    }
    planes_epilogue<G, PL>(p, smem_all, acc, tid, !loader, wm, wn, lane, n0, y00, x00, b, d, pre);
}

// ---- input gradient of the STRIDE-2 3x3 layers (MADNet pyramid conv3 / conv5 ...: Nets/MadNet.py:56-66, TF 'SAME' on even sizes: pad_before = 0) -------
// dx[y][x] = sum over taps with y - ky, x - kx even of dz[(y - ky) / 2][(x - kx) / 2] * w[ky][kx]: the four parity classes of (y, x) are four small
// convolutions over the SAME dz patch -- (even, even): taps {0,2} x {0,2} reading dz[i - {0,1}][j - {0,1}], (even, odd): {0,2} x {1}, (odd, even):
// {1} x {0,2}, (odd, odd): {1} x {1} -- 9 tap products per dz pixel in all, against 36 tap slots (27 of them multiplied by structural zeros, each with its
// gather) in the tiled kernel's zero-insertion form (r04 timeline: conv3's input gradient 27 - 34 us for 24 MB of traffic at 3.7x the algorithmic bytes).
// A wave owns one row of 32 dz pixels and four class accumulators; the dz patch (tile + one row above, one column to the left) is staged by LDS DMA from
// the bf16 shadow of dz, the bank is the one-plane mirrored / transposed image of mh_conv2d_planes_bwd (walk step t = forward tap 8 - t); each class
// leaves through the common epilogue on the stride-2 lattice of dx (mask of the layer's input from its bf16 shadow, fp32 + shadow stores).
// KH = 5 (round 6: DispNet conv2, 'SAME' pads 1 in front): y + 1 - ky even -- even rows take ky in {1, 3} (dz rows i, i - 1), odd rows ky in {0, 2, 4} (dz rows i + 1, i, i - 1):
// 2x2 + 2x3 + 3x2 + 3x3 = 25 tap products per dz pixel, the patch has a row / column on BOTH sides.  One formula for both sizes: class parity py = (ky + pt) & 1,
// dz row offset di = (py + pt - ky) / 2.
template <int WM, int WN, int K16, int KH = 3>
__global__ __launch_bounds__(WM * WN * 64) void conv_planes_s2bwd_kernel(PlanesArgs p) {
    using G = PlanesGeo<32, WM, WN, 1, K16, 1>;                      // epilogue geometry: one M-block per wave
    constexpr int NW = G::NW, TR = WM, BN = G::BN, NCK = G::NCK, NCK1 = G::NCK1;
    constexpr int PT = (KH - 2) / 2, BELOW = (KH == 5) ? 1 : 0;      // forward padding in front; dz rows / columns needed BEHIND the tile
    constexpr int PR = TR + 1 + BELOW, PC = 33 + BELOW, ROWP = PC * NCK1;
    constexpr int PLANE_BLKS = (PR * ROWP * 16 + 1023) / 1024;
    HIP_DYNAMIC_SHARED(float, smem_all)
    unsigned char* const smem = reinterpret_cast<unsigned char*>(smem_all);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int lin = mh_xcd_remap(blockIdx.x, p.nwg);
    int tile_n, ttx, tty, cx, cy, b;
    mh_decode_tile(lin, p.dec, tile_n, ttx, tty, cx, cy, b);
    const int n0 = tile_n * BN;
    const int i00 = tty * TR, j00 = ttx * 32;                        // dz position of tile pixel (0, 0)

    constexpr int T = KH * KH * K16, NSTB = 8, PF = NSTB - 1;
    const __amdgpu_buffer_rsrc_t rs_b = mh_make_rsrc(p.wb, p.wb_bytes);
    const int nt32 = (p.N + 31) >> 5;
    const int nt = tile_n * WN + wn;
    const int voff_b = nt < nt32 ? nt * 1024 + lane * 16 : MH_OOB;
    const int step_b = nt32 * 1024;
    u32x4 fb[NSTB];
    auto issue_b = [&](int t, int slot) { if (t < T) fb[slot] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, voff_b, planes_bank_step<K16, KH * KH>(t) * step_b, 0); };
#pragma unroll
    for (int t = 0; t < PF; ++t) issue_b(t, t % NSTB);
    {
        const mh_dma_src rs_h = mh_make_dma_src(p.in_hi, p.in_bytes);
        const int pix_b = p.in_pld * 2;
        for (int i = wave; i < PLANE_BLKS; i += NW) {
            const int g = i * 64 + lane;
            const int pr = g / ROWP, rem = g - pr * ROWP;
            const int pc = rem / NCK1, c = rem - pc * NCK1;
            const int iy = i00 + pr - 1, ix = j00 + pc - 1;
            const bool ok = pr < PR && c < NCK && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
            mh_glds16(rs_h, smem + i * 1024, ok ? ((b * p.Hin + iy) * p.Win + ix) * pix_b + c * 16 : MH_OOB);
        }
    }
    MH_WAIT_VMCNT(0);
    __syncthreads();
    f32x16 acc[4][1];                                                // class 2 py + px
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][0][r] = 0.f;
    {
        const int j = lane & 31, kg = lane >> 5;
        const unsigned char* const a0 = smem + ((wm * ROWP + j * NCK1 + kg) * 16);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int tap = t / K16, s = t - tap * K16;
            const int f = KH * KH - 1 - tap;                         // forward tap of this bank step (the bank is the mirrored image)
            const int ky = f / KH, kx = f - ky * KH;
            const int py = (ky + PT) & 1, px = (kx + PT) & 1;        // the parity class of dx this tap feeds
            const int di = (py + PT - ky) / 2, dj = (px + PT - kx) / 2;      // dz pixel = (i + di, j + dj), di / dj in {-1, 0, 1} (exact: the numerators are even)
            const int cls = 2 * py + px;
            const int imm = (((1 + di) * ROWP + (1 + dj) * NCK1 + 2 * s) * 16);      // patch (0, 0) = dz (i00 - 1, j00 - 1)
            const u32x4 fa = *reinterpret_cast<const u32x4*>(a0 + imm);
            acc[cls][0] = mh_mfma_bf16_32(fa, fb[t % NSTB], acc[cls][0]);
            issue_b(t + PF, (t + PF) % NSTB);
        }
    }
    __syncthreads();
    PlanesEpiPre<G, 2> pre;                                          // no bias in an input gradient; the mask is read inside the epilogue's loop (PL = 2 form)
    pre.b0 = (f32x4){0.f, 0.f, 0.f, 0.f}; pre.b1 = pre.b0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        planes_epilogue<G, 2>(p, smem_all, acc[c], tid, true, wm, wn, lane, n0, 2 * i00 + (c >> 1), 2 * j00 + (c & 1), b, 2, pre);
        __syncthreads();
    }
}

// ---- STRIDE-2 forward layers, KH x KH (3 / 5), split-bf16 or plain bf16 from planes (round 6) ------------------------------------------------------------
// What is computed: tf.nn.conv2d(x, w, strides 2, 'SAME') + bias + leaky (Nets/sharedLayers.py:54-66) of DispNet's conv2 (5x5, 64 -> 128 at 192x640 -> 96x320:
// Nets/DispNet.py:80-84 -- 25 GFLOP for the two towers, until now 2 x 131 us on the exact-fp32 tiled kernel) and of the pyramids' 3x3 down-sampling layers.
// out[y][x] = sum_{ky,kx,c} in[2y + ky - pt][2x + kx - pl][c] w[ky][kx][c]  (pt = pl = (KH - 2) / 2 on even sizes: TF pads mostly behind).
// The walk is conv_planes_kernel's: a wave owns MBW output rows x 32 output columns x 32 output channels, weight fragments from the 32x32x16 bank
// (KH * KH taps), no barrier, every LDS offset an immediate.  What changes is the patch: 2 MBW + KH - 2 input rows x 64 + KH - 2 input columns, staged by
// LDS DMA with the columns SPLIT BY PARITY -- [row][parity][half-column][2 K16 + 1 chunks] -- so that the 32 lanes of an A-operand read (output columns
// x .. x + 31 = input columns 2 x + kx: one parity, consecutive half-columns) stay the conflict-free stride of the stride-1 kernel.
template <int KH, int WN, int MBW, int K16, int PL>
struct PlanesS2Geo : PlanesGeo<32, 1, WN, MBW, K16, PL> {
    using B = PlanesGeo<32, 1, WN, MBW, K16, PL>;
    static constexpr int PR2 = 2 * MBW + KH - 2;                       // input rows of the patch
    static constexpr int HC = 32 + (KH - 1) / 2;                       // half-columns per parity (even parity needs one more than odd for odd KH)
    static constexpr int ROWP2 = 2 * HC * B::NCK1;                     // chunks per patch row
    static constexpr int PLANE_BLKS2 = (PR2 * ROWP2 * 16 + 1023) / 1024;
    static constexpr int PLANE_BYTES2 = PLANE_BLKS2 * 1024;
    static constexpr int LDS2 = PL * PLANE_BYTES2 > B::LDS_CS ? PL * PLANE_BYTES2 : B::LDS_CS;
};

template <int KH, int WN, int MBW, int K16, int PL>
__global__ __launch_bounds__(WN * 64) void conv_planes_s2fwd_kernel(PlanesArgs p) {
    using G = PlanesS2Geo<KH, WN, MBW, K16, PL>;
    using GB = typename G::B;
    constexpr int NW = WN, BN = GB::BN, NCK = GB::NCK, NCK1 = GB::NCK1;
    constexpr int PR = G::PR2, HC = G::HC, ROWP = G::ROWP2, PLANE_BLKS = G::PLANE_BLKS2, PLANE_BYTES = G::PLANE_BYTES2;
    HIP_DYNAMIC_SHARED(float, smem_all)
    unsigned char* const smem = reinterpret_cast<unsigned char*>(smem_all);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave;
    const int lin = mh_xcd_remap(blockIdx.x, p.nwg);
    int tile_n, ttx, tty, cx, cy, b;
    mh_decode_tile(lin, p.dec, tile_n, ttx, tty, cx, cy, b);
    const int n0 = tile_n * BN;
    const int y00 = tty * MBW, x00 = ttx * 32;                        // OUTPUT position of tile pixel (0, 0)
    PlanesEpiPre<GB, PL> pre;
    planes_epilogue_prefetch<GB, PL>(p, tid, n0, y00, x00, b, 1, pre);

    constexpr int T = KH * KH * K16, NSTB = 8, PF = NSTB - 1;
    const __amdgpu_buffer_rsrc_t rs_b = mh_make_rsrc(p.wb, p.wb_bytes);
    const int nt32 = (p.N + 31) >> 5;
    const int nt = tile_n * WN + wn;
    const int voff_b = nt < nt32 ? nt * (PL * 1024) + lane * 16 : MH_OOB;
    const int step_b = nt32 * (PL * 1024);
    u32x4 fb[NSTB][PL];
    auto issue_b = [&](int t, int slot) {
        if (t < T) {
            const int bt = planes_bank_step<K16, KH * KH>(t);
            fb[slot][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, voff_b, bt * step_b, 0);
            if constexpr (PL == 2) fb[slot][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, voff_b, bt * step_b + 1024, 0);
        }
    };
#pragma unroll
    for (int t = 0; t < PF; ++t) issue_b(t, t % NSTB);

    // ---- stage the patch: chunk g of a plane <- (patch row, parity, half-column, channel chunk); input pixel = (2 y00 - pt + row, 2 x00 - pl + 2 half + parity)
    {
        const mh_dma_src rs_h = mh_make_dma_src(p.in_hi, p.in_bytes), rs_l = mh_make_dma_src(PL == 2 ? p.in_lo : p.in_hi, p.in_bytes);
        const int pix_b = p.in_pld * 2;
        const int iy0 = 2 * y00 - p.pad_t, ix0 = 2 * x00 - p.pad_l;
        for (int i = wave; i < PLANE_BLKS; i += NW) {
            const int g = i * 64 + lane;
            const int pr = g / ROWP, rem = g - pr * ROWP;
            const int ph = rem / NCK1, c = rem - ph * NCK1;              // ph = parity * HC + half-column
            const int par = ph / HC, hc = ph - par * HC;
            const int iy = iy0 + pr, ix = ix0 + 2 * hc + par;
            const bool ok = pr < PR && c < NCK && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
            const int off = ok ? ((b * p.Hin + iy) * p.Win + ix) * pix_b + c * 16 : MH_OOB;
            mh_glds16(rs_h, smem + i * 1024, off);
            if constexpr (PL == 2) mh_glds16(rs_l, smem + PLANE_BYTES + i * 1024, off);
        }
    }
    MH_WAIT_VMCNT(0);
    __syncthreads();

    f32x16 acc[MBW];
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
    {
        const int j = lane & 31, kg = lane >> 5;
        const unsigned char* const a_h = smem + ((j * NCK1 + kg) * 16);
        const unsigned char* const a_l = a_h + PLANE_BYTES;
        u32x4 fa[2][MBW][PL];
        auto issue_a = [&](int t, int set) {
            if (t < T) {
                const int tap = t / K16, s = t - tap * K16;
                const int ky = tap / KH, kx = tap - ky * KH;
#pragma unroll
                for (int mb = 0; mb < MBW; ++mb) {
                    const int imm = (((2 * mb + ky) * ROWP + ((kx & 1) * HC + (kx >> 1)) * NCK1 + 2 * s) * 16);
                    fa[set][mb][0] = *reinterpret_cast<const u32x4*>(a_h + imm);
                    if constexpr (PL == 2) fa[set][mb][1] = *reinterpret_cast<const u32x4*>(a_l + imm);
                }
            }
        };
        issue_a(0, 0);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int sa = t & 1, sb = t % NSTB;
#pragma unroll
            for (int term = (PL == 2 ? 0 : 2); term < 3; ++term) {
#pragma unroll
                for (int mb = 0; mb < MBW; ++mb) {
                    acc[mb] = mh_mfma_bf16_32(fa[sa][mb][term == 0 ? PL - 1 : 0], fb[sb][term == 1 ? PL - 1 : 0], acc[mb]);
                    if (term == (PL == 2 ? 0 : 2) && mb == 0) issue_a(t + 1, sa ^ 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            issue_b(t + PF, (t + PF) % NSTB);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();                                 // every wave is done with the patch: the accumulator tile goes over it
    planes_epilogue<GB, PL>(p, smem_all, acc, tid, true, 0, wn, lane, n0, y00, x00, b, 1, pre);
}

// fp32 NHWC -> the two bf16 planes (the operands of conv_planes_kernel): tensors no plane-writing kernel produces (cost-volume buffers, exact-fp32
// layers' outputs).  lo may be null (then exactly mh_shadow_cast).
__global__ __launch_bounds__(256) void plane_split_kernel(const mh_plane_seg* __restrict__ segs, int nseg) {
    const int lo = mh_find_seg(segs, nseg, (int)blockIdx.x);
    const mh_plane_seg sg = segs[lo];
    const int g8 = sg.dst_ld >> 3;
    const int64_t item = (int64_t)((int)blockIdx.x - sg.blk0) * 256 + threadIdx.x;
    if (item >= sg.npix * g8) return;
    const int64_t pix = item / g8;
    const int c0 = (int)(item - pix * g8) * 8;
    const float* s = sg.src + pix * sg.src_ld + c0;
    float v[8];
    if (c0 + 8 <= sg.C && (sg.src_ld & 3) == 0 && ((uintptr_t)sg.src & 15) == 0) {
        const float4 a = *reinterpret_cast<const float4*>(s), bq = *reinterpret_cast<const float4*>(s + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = bq.x; v[5] = bq.y; v[6] = bq.z; v[7] = bq.w;
    } else {
        // channels [C, C + C2) come from the second source (a fused tf.concat: the context network's input = [features | disparity], MadNet.py:123)
        const float* s2 = sg.src2 ? sg.src2 + pix * sg.src2_ld : nullptr;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c0 + e;
            v[e] = c < sg.C ? s[e] : ((s2 && c < sg.C + sg.C2) ? s2[c - sg.C] : 0.f);
        }
    }
    unsigned hh[4], ll[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) mh_split_bf16x2(v[2 * e], v[2 * e + 1], hh[e], ll[e]);
    *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(sg.hi) + pix * sg.dst_ld + c0) = (u32x4){hh[0], hh[1], hh[2], hh[3]};
    if (sg.lo) *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(sg.lo) + pix * sg.dst_ld + c0) = (u32x4){ll[0], ll[1], ll[2], ll[3]};
}

std::atomic<int> g_planes_mode{0};       // mh_tune_conv_planes: bits 0-3 tile variant (0 = heuristic), bits 4 / 5 / 6 staggered tiles (dispatch_planes), bits 8 .. 15 timing experiments (PlanesArgs::dbg)
std::atomic<int> g_planes_launches{0};

template <int MC, int WM, int WN, int MBW, int K16, int PL>
int launch_planes(PlanesArgs& a, hipStream_t s, bool attr_only) {
    using G = PlanesGeo<MC, WM, WN, MBW, K16, PL>;
    static_assert(G::LDS <= 160 * 1024, "patch planes exceed the LDS");
    static std::atomic<uint64_t> attr_done{0};
    const uint64_t attr_dev = mh_device_bit();
    if (!(attr_done.load(std::memory_order_relaxed) & attr_dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_planes_kernel<MC, WM, WN, MBW, K16, PL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { mh_set_error("conv_planes: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_done.fetch_or(attr_dev);
    }
    if (attr_only) return 0;
    const int d = a.dil;
    a.tiles_y = mh_cdiv(mh_cdiv(a.H, d), G::TR);
    a.tiles_x = mh_cdiv(mh_cdiv(a.W, d), MC);
    a.ntiles_n = mh_cdiv(a.N, G::BN);
    a.nwg = a.B * d * d * a.tiles_y * a.tiles_x * a.ntiles_n;
    {
        const int64_t nwg64 = (int64_t)a.nwg;
        const int dmax = std::max(std::max(a.ntiles_n, a.tiles_x), std::max(a.tiles_y, (int)d));
        MH_REQUIRE(mh_fastdiv_ok(nwg64, dmax), MH_ERR_UNSUPPORTED, "tile decode: %lld workgroups x divisor %d exceeds the 2^32 range of the magic-multiplier division", (long long)nwg64, dmax);
    }
    a.dec = mh_make_tile_decode(a.ntiles_n, a.tiles_x, a.tiles_y, d);
    a.dbg = (g_planes_mode.load(std::memory_order_relaxed) >> 8) & 255;
    ++g_planes_launches;
    mh_note_kernel("conv_planes_kernel<MC=%d,%dx%d waves,MBW=%d,K16=%d,%s> tile %dx%d K=%d dil=%d grid %d lds %d", MC, WM, WN, MBW, K16, PL == 2 ? "bf16x3" : "bf16",
                   G::BM, G::BN, a.K, a.dil, a.nwg, G::LDS);
    hipLaunchKernelGGL((conv_planes_kernel<MC, WM, WN, MBW, K16, PL>), dim3(a.nwg), dim3(G::NTH), G::LDS, s, a);
    return mh_check_launch("conv_planes");
}

template <int MC, int WN, int MBW, int K16, int PL>
int launch_planes_stg(PlanesArgs& a, hipStream_t s, bool attr_only) {
    using G = PlanesStgGeo<MC, WN, MBW, K16, PL>;
    static_assert(G::LDS_STG <= 160 * 1024, "patch planes + the waves' transposition blocks exceed the LDS");
    static std::atomic<uint64_t> attr_done{0};
    const uint64_t attr_dev = mh_device_bit();
    if (!(attr_done.load(std::memory_order_relaxed) & attr_dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_planes_kernel_stg<MC, WN, MBW, K16, PL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { mh_set_error("conv_planes (staggered): hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_done.fetch_or(attr_dev);
    }
    if (attr_only) return 0;
    const int d = a.dil;
    a.tiles_y = mh_cdiv(mh_cdiv(a.H, d), G::TR);
    a.tiles_x = mh_cdiv(mh_cdiv(a.W, d), MC);
    a.ntiles_n = mh_cdiv(a.N, G::BN);
    a.nwg = a.B * d * d * a.tiles_y * a.tiles_x * a.ntiles_n;
    {
        const int64_t nwg64 = (int64_t)a.nwg;
        const int dmax = std::max(std::max(a.ntiles_n, a.tiles_x), std::max(a.tiles_y, (int)d));
        MH_REQUIRE(mh_fastdiv_ok(nwg64, dmax), MH_ERR_UNSUPPORTED, "tile decode: %lld workgroups x divisor %d exceeds the 2^32 range of the magic-multiplier division", (long long)nwg64, dmax);
    }
    a.dec = mh_make_tile_decode(a.ntiles_n, a.tiles_x, a.tiles_y, d);
    a.dbg = (g_planes_mode.load(std::memory_order_relaxed) >> 8) & 255;
    ++g_planes_launches;
    mh_note_kernel("conv_planes_kernel<MC=%d,2x%d waves staggered,MBW=%d,K16=%d,%s> tile %dx%d K=%d dil=%d grid %d lds %d", MC, WN, MBW, K16, PL == 2 ? "bf16x3" : "bf16",
                   G::BM, G::BN, a.K, a.dil, a.nwg, G::LDS_STG);
    hipLaunchKernelGGL((conv_planes_kernel_stg<MC, WN, MBW, K16, PL>), dim3(a.nwg), dim3(G::NTH), G::LDS_STG, s, a);
    return mh_check_launch("conv_planes (staggered)");
}

template <int MC, int WM, int WN, int MBW, int K16, int PL, int NBUF>
int launch_planes_ck(PlanesArgs& a, hipStream_t s, bool attr_only) {
    using G = PlanesCkGeo<MC, WM, WN, MBW, K16, PL, NBUF>;
    static_assert(G::LDS_CK <= 160 * 1024, "patch buffers exceed the LDS");
    static std::atomic<uint64_t> attr_done{0};
    const uint64_t attr_dev = mh_device_bit();
    if (!(attr_done.load(std::memory_order_relaxed) & attr_dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_planes_ck_kernel<MC, WM, WN, MBW, K16, PL, NBUF>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { mh_set_error("conv_planes_ck: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_done.fetch_or(attr_dev);
    }
    if (attr_only) return 0;
    const int d = a.dil;
    a.tiles_y = mh_cdiv(mh_cdiv(a.H, d), G::TR);
    a.tiles_x = mh_cdiv(mh_cdiv(a.W, d), MC);
    a.ntiles_n = mh_cdiv(a.N, G::BN);
    a.nwg = a.B * d * d * a.tiles_y * a.tiles_x * a.ntiles_n;
    {
        const int64_t nwg64 = (int64_t)a.nwg;
        const int dmax = std::max(std::max(a.ntiles_n, a.tiles_x), std::max(a.tiles_y, (int)d));
        MH_REQUIRE(mh_fastdiv_ok(nwg64, dmax), MH_ERR_UNSUPPORTED, "tile decode: %lld workgroups x divisor %d exceeds the 2^32 range of the magic-multiplier division", (long long)nwg64, dmax);
    }
    a.dec = mh_make_tile_decode(a.ntiles_n, a.tiles_x, a.tiles_y, d);
    a.nchunks = mh_cdiv(mh_cdiv(a.K, 16), K16);
    a.dbg = 0;
    ++g_planes_launches;
    mh_note_kernel("conv_planes_ck_kernel<%dx%d waves + loader,MBW=%d,chunk K16=%d x %d,%s,%d buffers> tile %dx%d K=%d dil=%d grid %d lds %d", WM, WN, MBW, K16, a.nchunks,
                   PL == 2 ? "bf16x3" : "bf16", NBUF, G::BM, G::BN, a.K, a.dil, a.nwg, G::LDS_CK);
    hipLaunchKernelGGL((conv_planes_ck_kernel<MC, WM, WN, MBW, K16, PL, NBUF>), dim3(a.nwg), dim3((WM * WN + 1) * 64), G::LDS_CK, s, a);
    return mh_check_launch("conv_planes_ck");
}

template <int WM, int WN, int K16, int KH = 3>
int launch_planes_s2bwd(PlanesArgs& a, hipStream_t s) {
    using G = PlanesGeo<32, WM, WN, 1, K16, 1>;
    constexpr int BELOW = (KH == 5) ? 1 : 0;
    constexpr int PR = WM + 1 + BELOW, ROWP = (33 + BELOW) * G::NCK1;
    constexpr int LDS_P = ((PR * ROWP * 16 + 1023) / 1024) * 1024;
    constexpr int LDS = LDS_P > G::LDS_CS ? LDS_P : G::LDS_CS;
    static std::atomic<uint64_t> attr_done{0};
    const uint64_t attr_dev = mh_device_bit();
    if (!(attr_done.load(std::memory_order_relaxed) & attr_dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_planes_s2bwd_kernel<WM, WN, K16, KH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { mh_set_error("conv_planes_s2bwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_done.fetch_or(attr_dev);
    }
    a.tiles_y = mh_cdiv(a.Hin, WM);
    a.tiles_x = mh_cdiv(a.Win, 32);
    a.ntiles_n = mh_cdiv(a.N, G::BN);
    a.nwg = a.B * a.tiles_y * a.tiles_x * a.ntiles_n;
    {
        const int64_t nwg64 = (int64_t)a.nwg;
        const int dmax = std::max(std::max(a.ntiles_n, a.tiles_x), std::max(a.tiles_y, (int)1));
        MH_REQUIRE(mh_fastdiv_ok(nwg64, dmax), MH_ERR_UNSUPPORTED, "tile decode: %lld workgroups x divisor %d exceeds the 2^32 range of the magic-multiplier division", (long long)nwg64, dmax);
    }
    a.dec = mh_make_tile_decode(a.ntiles_n, a.tiles_x, a.tiles_y, 1);
    a.dbg = 0;
    ++g_planes_launches;
    mh_note_kernel("conv_planes_s2bwd_kernel<%dx%d waves,K16=%d,%dx%d,bf16> dz tile %dx32 -> dx %dx64, K=%d N=%d grid %d lds %d", WM, WN, K16, KH, KH, WM, 2 * WM, a.K, a.N, a.nwg, LDS);
    hipLaunchKernelGGL((conv_planes_s2bwd_kernel<WM, WN, K16, KH>), dim3(a.nwg), dim3(WM * WN * 64), LDS, s, a);
    return mh_check_launch("conv_planes_s2bwd");
}

template <int KH, int WN, int MBW, int K16, int PL>
int launch_planes_s2fwd(PlanesArgs& a, hipStream_t s, bool attr_only) {
    using G = PlanesS2Geo<KH, WN, MBW, K16, PL>;
    static_assert(G::LDS2 <= 160 * 1024, "patch planes exceed the LDS");
    static std::atomic<uint64_t> attr_done{0};
    const uint64_t attr_dev = mh_device_bit();
    if (!(attr_done.load(std::memory_order_relaxed) & attr_dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_planes_s2fwd_kernel<KH, WN, MBW, K16, PL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { mh_set_error("conv_planes_s2fwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_done.fetch_or(attr_dev);
    }
    if (attr_only) return 0;
    a.tiles_y = mh_cdiv(a.H, MBW);
    a.tiles_x = mh_cdiv(a.W, 32);
    a.ntiles_n = mh_cdiv(a.N, WN * 32);
    a.nwg = a.B * a.tiles_y * a.tiles_x * a.ntiles_n;
    {
        const int64_t nwg64 = (int64_t)a.nwg;
        const int dmax = std::max(std::max(a.ntiles_n, a.tiles_x), std::max(a.tiles_y, 1));
        MH_REQUIRE(mh_fastdiv_ok(nwg64, dmax), MH_ERR_UNSUPPORTED, "tile decode: %lld workgroups x divisor %d exceeds the 2^32 range of the magic-multiplier division", (long long)nwg64, dmax);
    }
    a.dec = mh_make_tile_decode(a.ntiles_n, a.tiles_x, a.tiles_y, 1);
    a.dbg = 0;
    ++g_planes_launches;
    mh_note_kernel("conv_planes_s2fwd_kernel<%dx%d,1x%d waves,MBW=%d,K16=%d,%s> tile %dx%d K=%d grid %d lds %d", KH, KH, WN, MBW, K16, PL == 2 ? "bf16x3" : "bf16", MBW * 32, WN * 32,
                   a.K, a.nwg, G::LDS2);
    hipLaunchKernelGGL((conv_planes_s2fwd_kernel<KH, WN, MBW, K16, PL>), dim3(a.nwg), dim3(WN * 64), G::LDS2, s, a);
    return mh_check_launch("conv_planes_s2fwd");
}

// the stride-2 forward instances: (kernel size, K16, 32-column waves) -> launcher.  DispNet conv2 (5x5, 64 -> 128); MADNet / DispNet 3x3 down-sampling layers as they get wired
struct PlanesS2Inst { int kh, k16, wn, pl, mbw; int (*launch)(PlanesArgs&, hipStream_t, bool); };
#define S2_INST(KH, WN, MBW, K16, PL) {KH, K16, WN, PL, MBW, &launch_planes_s2fwd<KH, WN, MBW, K16, PL>}
const PlanesS2Inst g_planes_s2_inst[] = {
    S2_INST(5, 4, 2, 4, 2), S2_INST(5, 4, 2, 4, 1),                                   // 5x5 64 -> 128: two output rows per tile is what the LDS holds (137 KB of patch planes)
    S2_INST(5, 4, 2, 10, 1),                                                           // 5x5 145 -> 256 (DispNet conv3, plain bf16), TWO output rows per tile: 157 KB of patch (DispNetSchedule.PLANES_S2_CONV3)
    // (round 6, r6j: a ONE-row instance of that layer measured 299 us against 73 us on the tiled kernel -- its 250-step walk was past hipcc's pragma-unroll budget and ran
    //  from dynamically indexed registers, see the Makefile; the two-row instance below, built with the larger budget, replaced it)
    S2_INST(3, 1, 4, 1, 2), S2_INST(3, 1, 2, 1, 2), S2_INST(3, 2, 4, 2, 2), S2_INST(3, 2, 2, 2, 2), S2_INST(3, 2, 1, 2, 2), S2_INST(3, 3, 2, 4, 2), S2_INST(3, 3, 1, 4, 2),
};
// the instance of a layer: the TALLEST tile (most output rows per wave = fewest weight-fragment loads per MFMA) that still gives every CU a workgroup, else the shortest
const PlanesS2Inst* planes_s2_find(int kh, int k16, int n32, int pl, int64_t rows_x_coltiles = -1) {
    const PlanesS2Inst* best = nullptr;
    for (const PlanesS2Inst& I : g_planes_s2_inst) {
        if (!(I.kh == kh && I.k16 == k16 && (I.wn == n32 || (n32 > 4 && I.wn == 4)) && I.pl == pl)) continue;       // more than 128 columns: 128-column tiles
        if (!best) { best = &I; continue; }
        if (rows_x_coltiles < 0) continue;
        const bool fills_b = rows_x_coltiles / best->mbw >= 192, fills_i = rows_x_coltiles / I.mbw >= 192;      // (three quarters of the CUs: 240 two-row tiles beat 480 one-row tiles)
        if ((fills_i && (!fills_b || I.mbw > best->mbw)) || (!fills_i && !fills_b && I.mbw < best->mbw)) best = &I;
    }
    return best;
}

// ---- instance table + tile choice ---------------------------------------------------------------------------------------------------------
// Every (N rounded up to 32, K16) pair has the 128-pixel instances of both M-block shapes; the pairs MADNet's layers use also have 64- / 32-pixel
// instances for grids that would not fill the chip (the 1/8-resolution level: 60 tiles of 128 pixels on 256 CUs).
struct PlanesInst { int mc, wm, wn, mbw, k16, lds; int (*launch)(PlanesArgs&, hipStream_t, bool); int pl; int nbuf; int stg; };    // nbuf > 0: K-chunked (k16 = the chunk); stg: staggered wave groups
#define STG_INST(MC, WN, MBW, K16, PL) {MC, 2, WN, MBW, K16, PlanesStgGeo<MC, WN, MBW, K16, PL>::LDS_STG, &launch_planes_stg<MC, WN, MBW, K16, PL>, PL, 0, 1}
#define STG_BOTH(WN, MBW, K16, PL) STG_INST(32, WN, MBW, K16, PL), STG_INST(16, WN, MBW, K16, PL)
#define PL_INST(MC, WM, WN, MBW, K16) {MC, WM, WN, MBW, K16, PlanesGeo<MC, WM, WN, MBW, K16, 2>::LDS, &launch_planes<MC, WM, WN, MBW, K16, 2>, 2, 0}
#define P1_INST(MC, WM, WN, MBW, K16) {MC, WM, WN, MBW, K16, PlanesGeo<MC, WM, WN, MBW, K16, 1>::LDS, &launch_planes<MC, WM, WN, MBW, K16, 1>, 1, 0}
#define CK_INST(WN, MBW, PL, NBUF) {32, 1, WN, MBW, MH_PLANES_KC16, PlanesCkGeo<32, 1, WN, MBW, MH_PLANES_KC16, PL, NBUF>::LDS_CK, &launch_planes_ck<32, 1, WN, MBW, MH_PLANES_KC16, PL, NBUF>, PL, NBUF}
#define P1_BOTH(WM, WN, MBW, K16) P1_INST(32, WM, WN, MBW, K16), P1_INST(16, WM, WN, MBW, K16)
#define PL_BOTH(WM, WN, MBW, K16) PL_INST(32, WM, WN, MBW, K16), PL_INST(16, WM, WN, MBW, K16)
#define PL_BASE(K16) PL_BOTH(1, 4, 4, K16), PL_BOTH(2, 3, 2, K16), PL_BOTH(2, 2, 2, K16), PL_BOTH(4, 1, 1, K16)
const PlanesInst g_planes_inst[] = {
    PL_BASE(2), PL_BASE(3), PL_BASE(4), PL_BASE(5), PL_BASE(6), PL_BASE(8),
    // 128 columns: 64- and 32-pixel tiles (K = 33 / 38 / 70 / 128: the estimators' and the context network's first and second layers)
    PL_BOTH(1, 4, 2, 3), PL_BOTH(1, 4, 2, 5), PL_BOTH(1, 4, 2, 8), PL_BOTH(1, 4, 1, 5), PL_BOTH(1, 4, 1, 8),
    // 96 columns (128 -> 96): 64 pixels x 3 waves, 32 pixels x 3 waves
    PL_BOTH(1, 3, 2, 8), PL_BOTH(1, 3, 1, 8),
    // 64 columns (96 -> 64, 64 -> 64): 64 pixels x 4 waves, 32 pixels x 2 waves
    PL_BOTH(2, 2, 1, 6), PL_BOTH(2, 2, 1, 4), PL_BOTH(1, 2, 1, 6), PL_BOTH(1, 2, 1, 4),
    // 32 columns (64 -> 32, 32 -> 32): 64 pixels x 2 waves
    PL_BOTH(2, 1, 1, 4), PL_BOTH(2, 1, 1, 2),
    // ---- one plane (plain bf16): the input gradients of those layers -- (reduction over Cout, output columns = Cin):
    // 128 -> 128 (K16 8, 128 columns), 128 -> 96 (K16 6, 128), 96 -> 64 (K16 4, 96), 64 -> 32 (K16 2, 64), 64 -> 64 (K16 4, 64), 32 -> 32 (K16 2, 32)
    P1_BOTH(1, 4, 4, 8), P1_BOTH(1, 4, 2, 8), P1_BOTH(1, 4, 1, 8), P1_BOTH(1, 4, 4, 6), P1_BOTH(1, 4, 2, 6), P1_BOTH(1, 4, 1, 6),
    P1_BOTH(2, 3, 2, 4), P1_BOTH(1, 3, 2, 4), P1_BOTH(1, 3, 1, 4),
    P1_BOTH(2, 2, 2, 2), P1_BOTH(2, 2, 1, 2), P1_BOTH(1, 2, 1, 2), P1_BOTH(2, 2, 2, 4), P1_BOTH(2, 2, 1, 4), P1_BOTH(1, 2, 1, 4),
    P1_BOTH(4, 1, 1, 2), P1_BOTH(2, 1, 1, 2),
    // ---- DispNet (Nets/DispNet.py:75-152) ------------------------------------------------------------------------------------------------
    // split-bf16 forward of the two finest iconv layers: 193 -> 64 (K16 13: 64-pixel tiles, the patch planes fill the LDS) and 97 -> 32 (K16 7)
    PL_INST(32, 1, 2, 2, 13), PL_INST(32, 1, 2, 1, 13), PL_INST(32, 4, 1, 1, 7), PL_INST(32, 2, 1, 1, 7),
    // their input gradients (one plane): reduction over 64 / 32 output channels, 193 / 97 columns in 128-column tiles
    P1_INST(32, 1, 4, 4, 4), P1_INST(32, 1, 4, 2, 4), P1_INST(32, 1, 4, 1, 4), P1_INST(32, 1, 4, 4, 2), P1_INST(32, 1, 4, 2, 2),
    // K-chunked (chunks of 64 channels): every layer / input gradient with a reduction over more than 128 channels, 128- or 64-column tiles
    CK_INST(4, 4, 1, 3), CK_INST(4, 2, 1, 3), CK_INST(4, 1, 1, 3), CK_INST(2, 2, 1, 3), CK_INST(2, 1, 1, 3),
    CK_INST(4, 2, 2, 3), CK_INST(4, 1, 2, 3), CK_INST(2, 2, 2, 3), CK_INST(2, 1, 2, 3), CK_INST(4, 4, 2, 2), CK_INST(3, 4, 1, 3), CK_INST(3, 2, 1, 3), CK_INST(3, 2, 2, 3), CK_INST(3, 1, 2, 3),
    // ---- staggered wave groups (128 x 128 tiles as two out-of-phase halves): the 128-column layers of the estimators / the context network and their input gradients
    STG_BOTH(4, 2, 3, 2), STG_BOTH(4, 2, 8, 2), STG_BOTH(4, 2, 6, 1), STG_BOTH(4, 2, 8, 1), STG_BOTH(3, 2, 8, 2), STG_BOTH(2, 2, 6, 2),
};
constexpr int N_PLANES_INST = sizeof(g_planes_inst) / sizeof(g_planes_inst[0]);

// estimated duration of a launch on 256 CUs, in units of one M-block's K walk: workgroup rounds x (M-blocks per wave x SIMD oversubscription + a
// fixed share for staging / epilogue / launch that grows as the walk gets shorter).  Measured anchors (profiles/r04_microbench_planes.txt):
// 240 x 128-pixel tiles = 480 x 64-pixel tiles at 96x320; 60 x 128 pixels is 1.6x slower than 120 x 64 pixels at 48x160; sub-lattices of a dilation
// whose 32-column tiles spill into a second round (dilation 4: 288 workgroups) lose 1.5x against the 16-column shape (240).
float planes_cost(const PlanesInst& I, const PlanesArgs& a, int* nwg_out) {
    const int mr = 32 / I.mc, tr = I.wm * I.mbw * mr, d = a.dil;
    const int64_t nwg = (int64_t)a.B * d * d * mh_cdiv(mh_cdiv(a.H, d), tr) * mh_cdiv(mh_cdiv(a.W, d), I.mc) * mh_cdiv(a.N, I.wn * 32);
    const int nw = I.wm * I.wn + (I.nbuf ? 1 : 0);
    int wpc = (160 * 1024) / I.lds;                         // co-resident workgroups per CU: LDS, and at most 8 waves worth scheduling for
    if (wpc * nw > 8) wpc = 8 / nw > 0 ? 8 / nw : 1;
    if (wpc < 1) wpc = 1;
    // K-chunked instances: the walk is nchunks chunk walks long (in units of a K16 = 8 walk), and a wave with few M-blocks is bound by its weight
    // fragments (1 KB per step per wave from L2, ~64 B/clk/CU: four waves need two M-blocks' worth of MFMA time per step to hide them)
    float walk = (float)I.mbw;
    if (I.nbuf) {
        const int nch = mh_cdiv(mh_cdiv(a.K, 16), I.k16);
        const float per_step = (float)I.mbw * (I.pl == 2 ? 3.f : 1.f);
        walk = (per_step < 2.f ? 2.f : per_step) / (I.pl == 2 ? 3.f : 1.f) * (float)(nch * I.k16) / 8.f;
    }
    // round by round: a CU holds min(wpc, what is left / 256) workgroups, its busiest SIMD ceil(workgroups x waves / 4) waves
    float t = 0.f;
    for (int64_t left = nwg; left > 0; left -= 256 * wpc) {
        const int64_t per_cu = (left + 255) / 256 < wpc ? (left + 255) / 256 : wpc;
        const int64_t simd = (per_cu * (I.wm * I.wn) + 3) / 4;
        t += walk * (float)simd + (I.pl == 2 ? 20.f : 60.f) / (float)(I.nbuf ? 8 : I.k16);
    }
    if (nwg_out) *nwg_out = (int)nwg;
    return t;
}

// Which instances serve a layer: the bank's layout is fixed by the reduction length alone (mh_planes_kc16: whole-K image up to 128 channels and for
// the two DispNet shapes with whole-K instances, chunk-major beyond), so a layer has either whole-K or chunked candidates, never both.
// Columns: up to 128 -> the instance's width must be the layer's (rounded to 32); more -> 128-column tiles (chunked: 64-column tiles too).
bool planes_inst_fits(const PlanesInst& I, int k16, int n32, int pl) {
    if (I.pl != pl || I.stg) return false;           // (staggered instances REPLACE the chosen 128-pixel instance: dispatch_planes)
    const bool chunked = mh_planes_kc16(k16 * 16) != 0;
    if (chunked != (I.nbuf != 0)) return false;
    if (!chunked && I.k16 != k16) return false;
    if (n32 <= 4) return chunked ? (I.wn == n32 || (n32 == 3 && I.wn == 4) || (n32 == 1 && I.wn == 2)) : I.wn == n32;
    return I.wn == 4 || (chunked && I.wn == 2);
}

int dispatch_planes(PlanesArgs& a, hipStream_t s, bool all, int pl = 2) {
    if (all) {
        for (int i = 0; i < N_PLANES_INST; ++i)
            if (int rc = g_planes_inst[i].launch(a, s, true)) return rc;
        return 0;
    }
    const int v = g_planes_mode.load(std::memory_order_relaxed) & 15;
    const int k16 = mh_cdiv(a.K, 16), n32 = mh_cdiv(a.N, 32);
    // forced variants (mh_tune_conv_planes): 1 / 2 / 5 = 128- / 64- / 32-pixel tiles of 32-column M-blocks, 3 / 4 / 6 = the same of 16-column M-blocks
    const int want_mc = (v == 3 || v == 4 || v == 6) ? 16 : 32, want_px = (v == 1 || v == 3) ? 128 : ((v == 2 || v == 4) ? 64 : 32);
    const PlanesInst* best = nullptr;
    float best_cost = 0.f;
    for (int i = 0; i < N_PLANES_INST; ++i) {
        const PlanesInst& I = g_planes_inst[i];
        if (!planes_inst_fits(I, k16, n32, pl)) continue;
        float c = planes_cost(I, a, nullptr);
        if (v != 0) c = (I.mc == want_mc ? 0.f : 1000.f) + (float)abs(I.wm * I.mbw * 32 - want_px);        // forced: the closest available shape
        else c -= 1e-3f * (float)(I.wm * I.mbw) + (I.mc == 32 ? 5e-4f : 0.f);                                // ties: the larger tile, then the one-row M-block
        if (!best || c < best_cost) { best = &I; best_cost = c; }
    }
    if (!best) {
        mh_set_error("mh_conv2d_planes%s: no instance for K = %d, N = %d", pl == 1 ? "_bwd" : "", a.K, a.N);
        return MH_ERR_UNSUPPORTED;
    }
    // a 128-pixel tile with one workgroup per CU: the staggered form of the same tile where it exists -- an OPTION (mh_tune_conv_planes bit 4: forward layers, bit 5: input
    // gradients), off by default.  Alone a launch gains 1 - 2.6 us; inside the FULL step the input gradients run beside the filter-gradient lanes, whose workgroups then find
    // neither the LDS nor the wave slots they had (the step LOSES 11 - 15 us), and the forward layers alone measured -6.6 .. +9.4 us over three boxes (r6n - r6r)
    const int gm = g_planes_mode.load(std::memory_order_relaxed);
    const bool stagger = (gm & (pl == 2 ? 16 : 32)) != 0;
    int nwg_best = 0;
    planes_cost(*best, a, &nwg_best);
    // (one workgroup per CU is the premise: with two co-resident workgroups -- dilation 16, 512 tiles -- the staggered form measured 23.2 us against 20.4)
    if (stagger && !best->nbuf && best->wm * best->mbw == 4 && nwg_best <= 256 && (!(gm & 128) || best->wn == 4))
        for (int i = 0; i < N_PLANES_INST; ++i) {
            const PlanesInst& J = g_planes_inst[i];
            if (J.stg && J.mc == best->mc && J.k16 == best->k16 && J.pl == pl && J.wn == best->wn) return J.launch(a, s, false);
        }
    return best->launch(a, s, false);
}

}  // namespace

int mh_conv_planes_init() {
    PlanesArgs a = {};
    for (const PlanesS2Inst& I : g_planes_s2_inst)
        if (int rc = I.launch(a, nullptr, true)) return rc;
    return dispatch_planes(a, nullptr, true);
}

extern "C" int mh_tune_conv_planes(int mode) {
    g_planes_mode = mode < 0 ? 0 : mode;
    return g_planes_launches.exchange(0);
}

// bytes of the two-plane image (one plane: half).  K-chunked layouts (mh_planes_kc16) pad the reduction to whole chunks.
extern "C" int64_t mh_pack32_bytes(int32_t taps, int32_t K, int32_t N) {
    int k16 = (K + 15) / 16;
    const int kc = mh_planes_kc16(K);
    if (kc) k16 = (k16 + kc - 1) / kc * kc;
    return (int64_t)taps * k16 * ((N + 31) / 32) * 2048;
}

static bool planes_has_instance(int k16, int n32, int pl) {
    for (int i = 0; i < N_PLANES_INST; ++i)
        if (planes_inst_fits(g_planes_inst[i], k16, n32, pl)) return true;
    return false;
}

// ---- input gradient of a stride-1 'SAME' 3x3 layer from bf16 shadows: dx = conv2d_backprop_input(dz, w) [* leaky'(mask)] ----------------------------
// stride-2 layers (forward 3x3, 'SAME' on even sizes: no padding in front): Cout in {32, 64} (K16 2 / 4), Cin <= 32
static bool planes_s2bwd_ok(const mh_conv_desc* d) {
    if (!(d->stride == 2 && d->dil == 1 && d->Hi == 2 * d->Ho && d->Wi == 2 * d->Wo && (d->in_ld == 0 || d->in_ld >= ((d->K + 7) & ~7)))) return false;
    // (accumulation onto earlier contributions -- DispNet's skip connection conv1a, MADNet's conv5 whose input is a cost-volume level -- is served by every stride-2 instance)
    if (d->kh == 3 && d->kw == 3 && d->pad_t == 0 && d->pad_l == 0) return (d->N == 32 || d->N == 64) && d->K >= 1 && d->K <= 32;
    // 5x5 (DispNet conv2: 64 -> 128): reduction over 128 output channels (K16 8), 33 .. 64 gradient columns (two 32-column waves)
    if (d->kh == 5 && d->kw == 5 && d->pad_t == 1 && d->pad_l == 1) return (d->N == 128 && d->K > 32 && d->K <= 64) || (d->N == 256 && d->K > 32 && d->K <= 2048);      // conv2 ; conv3 (145 columns in 64-column tiles)
    return false;
}

extern "C" int mh_conv2d_planes_bwd_ok(const mh_conv_desc* d) {
    if (!d) return 0;
    if (d->stride == 2) return planes_s2bwd_ok(d) ? 1 : 0;
    if (!(d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad_t == d->dil && d->pad_l == d->dil && d->Hi == d->Ho && d->Wi == d->Wo)) return 0;
    if (d->dil < 1 || d->dil > 64 || d->K < 1 || d->K > 2048 || d->N < 1 || d->N > 2048) return 0;          // d = the FORWARD layer: K = Cin = the gradient's columns
    if (d->in_ld > 0 && d->in_ld < ((d->K + 7) & ~7)) return 0;                                              // dx rows must hold Cin rounded up to 8 (8 columns per lane)
    return planes_has_instance((d->N + 15) / 16, (d->K + 31) / 32, 1) ? 1 : 0;
}

extern "C" int mh_conv2d_planes_bwd(const mh_conv_desc* d, const void* dz_hi, int32_t dz_pld, const void* wb32t, const void* mask_hi, int32_t mask_pld,
                                    float* dx, void* dx_hi, int32_t dx_pld, void* stream) {
    MH_REQUIRE(d && dz_hi && wb32t, MH_ERR_ARG, "mh_conv2d_planes_bwd: null descriptor / dz plane / fragment bank");
    MH_REQUIRE(dx || dx_hi, MH_ERR_ARG, "mh_conv2d_planes_bwd: no output");
    MH_REQUIRE(mh_conv2d_planes_bwd_ok(d), MH_ERR_UNSUPPORTED, "mh_conv2d_planes_bwd: 'SAME' 3x3 layers with an instance (stride 1; stride 2: Cout 32 or 64, Cin <= 32, even sizes) or the stride-2 5x5 layer 33..64 -> 128");
    MH_REQUIRE(d->B > 0 && d->Hi > 0 && d->Wi > 0, MH_ERR_ARG, "mh_conv2d_planes_bwd: non-positive size");
    const int k16 = (d->N + 15) / 16;
    const int k8 = (d->K + 7) & ~7;
    MH_REQUIRE(dz_pld >= k16 * 16 && (dz_pld & 7) == 0, MH_ERR_ARG, "mh_conv2d_planes_bwd: dz_pld must cover Cout rounded up to 16 (multiple of 8)");
    MH_REQUIRE(mh_aligned16(dz_hi) && mh_aligned16(wb32t) && mh_aligned16(mask_hi) && mh_aligned16(dx_hi), MH_ERR_ALIGN, "mh_conv2d_planes_bwd: 16-byte aligned planes / bank");
    MH_REQUIRE(!mask_hi || (mask_pld >= k8 && (mask_pld & 7) == 0), MH_ERR_ARG, "mh_conv2d_planes_bwd: mask_pld must cover Cin rounded up to 8 (multiple of 8)");
    // the epilogue stores 8 columns per lane: rows must hold Cin rounded up to 8 (the concat buffers of DispNet are allocated that way)
    if (dx) MH_REQUIRE(d->in_ld >= k8 && (d->in_ld & 3) == 0 && mh_aligned16(dx), MH_ERR_ALIGN, "mh_conv2d_planes_bwd: dx rows (d->in_ld floats >= Cin rounded up to 8) must be 16-byte aligned");
    if (dx_hi) MH_REQUIRE(dx_pld >= k8 && (dx_pld & 7) == 0, MH_ERR_ARG, "mh_conv2d_planes_bwd: dx_pld must cover Cin rounded up to 8 (multiple of 8)");
    // accumulate: dx = (what dx holds + this gradient) * mask -- the LAST contribution's form (the mask and the shadow are those of the total)
    MH_REQUIRE(!d->accumulate || (d->stride == 2 && dx), MH_ERR_UNSUPPORTED, "mh_conv2d_planes_bwd: accumulation only in the stride-2 forms, onto the fp32 map");
    const int64_t npix = (int64_t)d->B * d->Hi * d->Wi;
    MH_REQUIRE(npix * dz_pld * 2 < (1ll << 31) && npix * d->in_ld * 4 < (1ll << 31) && npix * (int64_t)dx_pld * 2 < (1ll << 31) && npix * (int64_t)mask_pld * 2 < (1ll << 31),
               MH_ERR_UNSUPPORTED, "mh_conv2d_planes_bwd: tensors must be < 2 GiB");
    PlanesArgs a = {};
    a.in_hi = (const unsigned short*)dz_hi; a.in_lo = nullptr; a.wb = wb32t; a.bias = nullptr;
    a.mask_hi = (const unsigned short*)mask_hi; a.mask_pld = mask_pld; a.mask_alpha = d->mask_alpha;
    a.mask_c0 = d->mask_c0; a.mask_c1 = (d->mask_c0 == 0 && d->mask_c1 == 0) ? 0x7fffffff : d->mask_c1;
    a.nchunks = 1;
    a.out = dx; a.out_hi = (unsigned short*)dx_hi; a.out_lo = nullptr;
    a.in_bytes = (unsigned)(npix * dz_pld * 2);
    a.wb_bytes = (unsigned)(mh_pack32_bytes(d->kh * d->kw, d->N, d->K) / 2);
    a.out_bytes = dx ? (unsigned)(npix * d->in_ld * 4) : 0u;
    a.outp_bytes = (unsigned)(npix * dx_pld * 2);
    a.in_pld = dz_pld; a.out_ld = d->in_ld; a.out_pld = dx_pld;
    a.B = d->B; a.H = d->Hi; a.W = d->Wi; a.K = d->N; a.N = d->K; a.dil = d->dil;      // the walk reduces over Cout and produces Cin columns
    a.alpha = 1.0f;
    a.Hin = d->Ho; a.Win = d->Wo;
    a.acc_out = d->accumulate ? 1 : 0;
    if (d->stride == 2) {
        a.in_bytes = (unsigned)((int64_t)d->B * d->Ho * d->Wo * dz_pld * 2);
        a.dil = 1;
        const bool few = (int64_t)d->B * mh_cdiv(d->Ho, 4) * mh_cdiv(d->Wo, 32) < 256;       // fewer than a workgroup per CU: two-row tiles
        if (d->kh == 5 && d->N == 256) return launch_planes_s2bwd<2, 2, 16, 5>(a, (hipStream_t)stream);        // DispNet conv3: reduction over 256 channels, two dz rows per tile (72 KB patch)
        if (d->kh == 5) return few ? launch_planes_s2bwd<2, 2, 8, 5>(a, (hipStream_t)stream) : launch_planes_s2bwd<4, 2, 8, 5>(a, (hipStream_t)stream);
        if (d->N == 32) return few ? launch_planes_s2bwd<2, 1, 2>(a, (hipStream_t)stream) : launch_planes_s2bwd<4, 1, 2>(a, (hipStream_t)stream);
        return few ? launch_planes_s2bwd<2, 1, 4>(a, (hipStream_t)stream) : launch_planes_s2bwd<4, 1, 4>(a, (hipStream_t)stream);
    }
    return dispatch_planes(a, (hipStream_t)stream, false, 1);
}

static __global__ __launch_bounds__(256) void plane_split_one_kernel(mh_plane_seg sg) {
    const int g8 = sg.dst_ld >> 3;
    const int64_t item = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (item >= sg.npix * g8) return;
    const int64_t pix = item / g8;
    const int c0 = (int)(item - pix * g8) * 8;
    const float* s = sg.src + pix * sg.src_ld + c0;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (c0 + e < sg.C) ? s[e] : 0.f;
    unsigned hh[4], ll[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) mh_split_bf16x2(v[2 * e], v[2 * e + 1], hh[e], ll[e]);
    *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(sg.hi) + pix * sg.dst_ld + c0) = (u32x4){hh[0], hh[1], hh[2], hh[3]};
    if (sg.lo) *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(sg.lo) + pix * sg.dst_ld + c0) = (u32x4){ll[0], ll[1], ll[2], ll[3]};
}
// one tensor, arguments by value (the fallback of mh_conv2d_sh4 for kernel families whose epilogue does not write the lo plane)
int mh_plane_split_one(const float* src, int src_ld, int C, void* hi, void* lo, int dst_ld, int64_t npix, hipStream_t s) {
    mh_plane_seg sg = {};
    sg.src = src; sg.hi = hi; sg.lo = lo; sg.npix = npix; sg.C = C; sg.src_ld = src_ld; sg.dst_ld = dst_ld;
    hipLaunchKernelGGL(plane_split_one_kernel, dim3((unsigned)((npix * (dst_ld / 8) + 255) / 256)), dim3(256), 0, s, sg);
    return mh_check_launch("plane_split_one");
}

extern "C" int mh_plane_split(const mh_plane_seg* segs_device, int32_t nseg, int32_t nblocks, void* stream) {
    MH_REQUIRE(segs_device && nseg > 0 && nblocks > 0, MH_ERR_ARG, "mh_plane_split: empty segment table");
    hipLaunchKernelGGL(plane_split_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, segs_device, nseg);
    return mh_check_launch("plane_split");
}

// stride-2 forward: KH x KH 'SAME' on even sizes (pads (KH - 2) / 2 in front), an instance for (KH, K16, N / 32), N a multiple of 32
static bool planes_s2fwd_ok(const mh_conv_desc* d) {
    if (!(d->stride == 2 && d->mode == 0 && d->kh == d->kw && (d->kh == 3 || d->kh == 5) && d->dil == 1 && !d->accumulate)) return false;
    if (d->Hi != 2 * d->Ho || d->Wi != 2 * d->Wo || d->pad_t != (d->kh - 2) / 2 || d->pad_l != (d->kh - 2) / 2) return false;
    if (d->N % 32 != 0 || d->K < 1) return false;
    return planes_s2_find(d->kh, (d->K + 15) / 16, d->N / 32, d->precision == 1 ? 1 : 2) != nullptr;
}

extern "C" int mh_conv2d_planes_ok(const mh_conv_desc* d) {
    if (!d) return 0;
    if (d->stride == 2) return planes_s2fwd_ok(d) ? 1 : 0;
    if (!(d->kh == 3 && d->kw == 3 && d->stride == 1 && d->mode == 0 && d->pad_t == d->dil && d->pad_l == d->dil && d->Hi == d->Ho && d->Wi == d->Wo)) return 0;
    if (d->accumulate || d->dil < 1 || d->dil > 64 || d->N < 1 || d->N > 2048 || (d->N & 7) || d->K < 1 || d->K > 2048) return 0;
    return planes_has_instance((d->K + 15) / 16, (d->N + 31) / 32, d->precision == 1 ? 1 : 2) ? 1 : 0;
}

extern "C" int mh_conv2d_planes(const mh_conv_desc* d, const void* in_hi, const void* in_lo, int32_t in_pld, const void* wb32, const float* bias,
                                float* out, void* out_hi, void* out_lo, int32_t out_pld, void* stream) {
    MH_REQUIRE(d && in_hi && wb32, MH_ERR_ARG, "mh_conv2d_planes: null descriptor / input plane / fragment bank");
    const int pl = d->precision == 1 ? 1 : 2;               // precision 1: plain bf16 from the hi plane and a ONE-plane bank; else split-bf16
    MH_REQUIRE(pl == 1 || in_lo, MH_ERR_ARG, "mh_conv2d_planes: the split-bf16 form needs the lo plane");
    MH_REQUIRE(out || out_hi || out_lo, MH_ERR_ARG, "mh_conv2d_planes: no output");
    MH_REQUIRE(mh_conv2d_planes_ok(d), MH_ERR_UNSUPPORTED, "mh_conv2d_planes: forward stride-1 'SAME' 3x3 layers (or stride-2 3x3 / 5x5 on even sizes) with an instance, N a multiple of 8, no accumulation");
    MH_REQUIRE(d->B > 0 && d->Hi > 0 && d->Wi > 0, MH_ERR_ARG, "mh_conv2d_planes: non-positive size");
    const int k16 = (d->K + 15) / 16;
    MH_REQUIRE(in_pld >= k16 * 16 && (in_pld & 7) == 0, MH_ERR_ARG, "mh_conv2d_planes: in_pld must cover K rounded up to 16 (multiple of 8)");
    MH_REQUIRE(mh_aligned16(in_hi) && mh_aligned16(in_lo) && mh_aligned16(wb32), MH_ERR_ALIGN, "mh_conv2d_planes: 16-byte aligned planes / bank");
    if (out) MH_REQUIRE(d->out_ld >= d->N && (d->out_ld & 3) == 0 && mh_aligned16(out), MH_ERR_ALIGN, "mh_conv2d_planes: out rows must be 16-byte aligned");
    if (out_hi || out_lo) {
        MH_REQUIRE(out_pld >= d->N && (out_pld & 7) == 0, MH_ERR_ARG, "mh_conv2d_planes: out_pld must cover N (multiple of 8)");
        MH_REQUIRE(mh_aligned16(out_hi) && mh_aligned16(out_lo), MH_ERR_ALIGN, "mh_conv2d_planes: 16-byte aligned output planes");
    }
    const int64_t npix_in = (int64_t)d->B * d->Hi * d->Wi;
    const int64_t npix = (int64_t)d->B * d->Ho * d->Wo;            // (= npix_in at stride 1)
    MH_REQUIRE(npix_in * in_pld * 2 < (1ll << 31) && npix * d->out_ld * 4 < (1ll << 31) && npix * (int64_t)out_pld * 2 < (1ll << 31), MH_ERR_UNSUPPORTED,
               "mh_conv2d_planes: tensors must be < 2 GiB");
    PlanesArgs a = {};
    a.in_hi = (const unsigned short*)in_hi; a.in_lo = (const unsigned short*)(pl == 2 ? in_lo : nullptr); a.wb = wb32; a.bias = bias;
    a.mask_hi = nullptr; a.mask_pld = 0; a.mask_alpha = 1.0f; a.mask_c0 = 0; a.mask_c1 = 0x7fffffff;
    a.nchunks = 1;
    a.out = out; a.out_hi = (unsigned short*)out_hi; a.out_lo = (unsigned short*)out_lo;
    a.in_bytes = (unsigned)(npix_in * in_pld * 2);
    a.wb_bytes = (unsigned)(mh_pack32_bytes(d->kh * d->kw, d->K, d->N) / (pl == 1 ? 2 : 1));
    a.out_bytes = out ? (unsigned)(npix * d->out_ld * 4) : 0u;
    a.outp_bytes = (unsigned)(npix * out_pld * 2);
    a.in_pld = in_pld; a.out_ld = d->out_ld; a.out_pld = out_pld;
    a.B = d->B; a.H = d->Ho; a.W = d->Wo; a.K = d->K; a.N = d->N; a.dil = d->dil;
    a.alpha = d->alpha;
    if (d->stride == 2) {
        a.Hin = d->Hi; a.Win = d->Wi; a.pad_t = d->pad_t; a.pad_l = d->pad_l; a.dil = 1;
        return planes_s2_find(d->kh, k16, d->N / 32, pl, (int64_t)d->B * d->Ho * mh_cdiv(d->Wo, 32))->launch(a, (hipStream_t)stream, false);
    }
    return dispatch_planes(a, (hipStream_t)stream, false, pl);
}
