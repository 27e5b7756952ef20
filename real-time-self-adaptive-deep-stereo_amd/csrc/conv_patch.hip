// conv_patch.hip -- patch-staged bf16-MFMA kernel for the stride-1, "SAME"-padded 3x3 (dilated) layers of the
// disparity estimators / context network (Nets/MadNet.py:70-118 via Nets/sharedLayers.py:54-92) and their input
// gradients, in the throughput mode (mh_conv_desc.precision = 1).
//
// Why a second kernel: the implicit-GEMM kernel of conv.hip re-gathers the input once per tap (9 x 128 pixels x 64
// channels of fp32 per K-tile pair) and converts it to bf16 on every pass: at 13x the fp32 MFMA rate it is bound by the
// 64 B/clk L1 path and the ~200 VALU instructions of a K-step, not by the matrix cores (profiles/r01_pmc_roofline.json:
// 7.5 M VALU for 0.9 M MFMA).  Here the workgroup stages the input ONCE:
//   * output tile = TH x 16 pixels of ONE dilation sub-lattice (pixels y = cy + d*i, x = cx + d*j): on its own lattice a
//     dilated 3x3 conv is a dense 3x3 conv, so the halo is one lattice pixel whatever the dilation;
//   * the (TH+2) x 18 patch is converted to bf16 once and kept in LDS ([pixel][channel], row stride K+16 halfs: the
//     b128 lane groups of gfx950 read 16 consecutive pixels conflict free for strides = 8 mod 16 dwords);
//   * the K loop only streams the weights (64 k-values per barrier, LDS double buffered, two register stages = prefetch
//     distance 2) and walks (tap, 32-channel chunk): the A fragments are ds_read_b128 at patch[(ty+oy)*18 + tx+ox][c..c+7];
//   * the walk is interleaved by hand (one LDS read / global load / convert + LDS store after each MFMA, pinned with
//     sched_barrier) and, for K = 64 / 128, specialised on the channel count so that every LDS offset is an immediate.
// Per 128x128 tile and 3x3x128 taps the L1 traffic drops from 864 KB to ~330 KB and the loader VALU work by ~5x.
// Results are those of conv_igemm_kernel<BF16> up to the fp32 summation order (same bf16 rounding of both operands).
// Measured on MI355X (profiles/r01_microbench_conv_patch.txt): 3x3 128->128 @ 96x320 forward 33 -> 19.6 us, dgrad 35.5 -> 23 us.
// Dispatch: mh_conv_patch_ok() below; build with -mllvm -amdgpu-mfma-vgpr-form=1 (csrc/Makefile) -- with AGPR accumulators
// hipcc shuffles them through v_accvgpr_* moves at every loop back-edge.
#include "conv_args.h"
#include <stdlib.h>
#include <algorithm>
#include <atomic>

namespace {

struct PatchGeo {
    int TH, tiles_y, tiles_x, ntiles_n, nwg;
    int KP, CPT, PS;        // channels rounded to 32, 32-chunks per tap, patch row stride (halfs)
    int nchunk;             // 9 * CPT
    int patch_halfs;        // (TH+2)*18*PS
    float inv_kp4;          // 1 / (KP/4)
    unsigned mul_kp8;       // ceil(2^32 / (KP/8)): q / (KP/8) by multiplication (shadow staging)
    int dbg;                // timing experiments (scripts/microbench.py patchdbg): 1 = skip the K walk, 2 = skip the patch staging
    mh_tile_decode dec;     // magic multipliers of the workgroup -> tile decode (mh_common.h)
    mh_fastdiv f_kp4, f_pc; // ... and of the small-layer kernel's patch staging (item -> (pixel, 4-channel group), pixel -> (row, column))
    // small-layer kernel: where a workgroup runs (8 XCDs with an L2 each; block b runs on XCD b % 8).  xuse < 8: the launch has 8 * per blocks and only the
    // `per` blocks of XCDs 0 .. xuse - 1 work (the others exit at once) -- a layer of <= 64 workgroups reads its bank and its patch through ONE L2.
    // n_major: the column tile is the SLOW index of the logical order, so an XCD's contiguous chunk shares bank slices instead of input patches.
    int xuse, per, n_major;
    mh_fastdiv f_pt;        // pixel tiles per column tile (n_major decode)
};

constexpr int LSB = 80;     // weight tile row stride (halfs): 64 k + 16 pad = 40 dwords (conflict-free b128 reads)
constexpr int PW = 18;      // patch width (16 + halo)

// CPTC > 0: the channel count is the compile-time constant 32 * CPTC (2 or 4 chunks per tap): tiles never straddle a tap,
// no channel masking, and every LDS offset of the K walk is an instruction immediate (see the specialised walk below).
// X3 (split-bf16, precision code 2; forward only): the patch and the weight tiles are kept as TWO bf16 planes, hi = bf16(x)
// and lo = bf16(x - hi), and every product runs as three MFMAs (lo*hi + hi*lo + hi*hi; lo*lo < 2^-16 is dropped): the
// forward pass stays within ~2^-16 relative of exact fp32 at bf16 MFMA rate / 3 instead of the fp32 MFMA rate (/ 16).  The
// split is paid once per staged element (patch: once per workgroup, not once per tap).
template <int WM, int WN, int MT, int NT, bool DGRAD, int CPTC = 0, bool X3 = false>
__global__ __launch_bounds__(WM * WN * 64) __attribute__((amdgpu_waves_per_eu(1, 2))) void conv_patch_kernel(ConvArgs p, PatchGeo g) {
    static_assert(!X3 || !DGRAD, "the split-bf16 instances are forward only");
    constexpr int NTH = WM * WN * 64;
    constexpr int BM = WM * MT * 16, BN = WN * NT * 16;
    constexpr int TH = BM / 16;
    HIP_DYNAMIC_SHARED(float, smem_all)
    unsigned short* const Ph = reinterpret_cast<unsigned short*>(smem_all);
    unsigned short* const Bh = Ph + (X3 ? 2 : 1) * g.patch_halfs;   // [2][BN][LSB]   (X3: the lo patch plane sits at Ph + patch_halfs,
    constexpr int BLO = 2 * BN * LSB;                               //                  the lo weight tiles at Bh + BLO)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, lq = lane >> 4;
    const int d = p.dil;

    const int lin = mh_xcd_remap(blockIdx.x, g.nwg);
    int tile_n, ttx, tty, cx, cy, b;
    mh_decode_tile(lin, g.dec, tile_n, ttx, tty, cx, cy, b);
    const int n0 = tile_n * BN;
    const int y00 = cy + d * (tty * TH), x00 = cx + d * (ttx * 16);      // image position of tile pixel (0, 0)

    const __amdgpu_buffer_rsrc_t rs_in = mh_make_rsrc(p.in, p.in_bytes);
    const __amdgpu_buffer_rsrc_t rs_w = mh_make_rsrc(p.w, p.w_bytes);

    // ---- weight tile loader: tile t = 32-chunks 2t, 2t+1 of the (tap, chunk) walk -------------------------------------
    // forward (HWIO, n contiguous): unit = 4 k-rows x 4 columns, transposed in registers, 4 ds_write_b64 (XOR-swizzled
    // 16-byte chunks, as conv_igemm_kernel's BT path);  dgrad (w[tap][n][k], k contiguous): float4 along k, 1 ds_write_b64.
    // Addressing = thread-constant byte offset + wave-uniform (tap, chunk) offset kept in a scalar cursor; every load is
    // unconditional (dead lanes / chunks past the walk get an out-of-range offset and read zeros).
    constexpr int FU = 512 / NTH;                   // forward units per thread and tile (256 unit slots per chunk)
    constexpr int DI = (16 * BN) / NTH;             // dgrad float4 items per thread and tile
    constexpr int NB = DGRAD ? DI : 4 * FU;
    constexpr int NV = DGRAD ? DI : FU;
    float4 rb0[NB], rb1[NB];                        // two register stages: tiles t+1 and t+2 are in flight while tile t is multiplied
    int voff[NV], vk[NV], vch[NV];                  // byte offset inside a (tap, chunk) slab, first k inside the chunk, chunk of the tile
#pragma unroll
    for (int jj = 0; jj < NV; ++jj) {
        const int q = tid + NTH * jj;
        if (!DGRAD) {
            const int wi = q & 255;
            const int n4 = wi % (BN / 4), kq = wi / (BN / 4);
            const int n = n0 + n4 * 4;
            vch[jj] = __builtin_amdgcn_readfirstlane(q >> 8);
            vk[jj] = kq * 4;
            voff[jj] = (kq < 8 && n < p.N) ? (kq * 4 * p.N + n) * 4 : MH_OOB;
        } else {
            const int ch = q / (8 * BN), rem = q - ch * (8 * BN);
            const int nn = rem >> 3, kg = rem & 7;
            vch[jj] = ch;                            // (8 * BN) % 64 == 0: wave uniform
            vk[jj] = kg * 4;
            voff[jj] = (n0 + nn < p.N) ? ((n0 + nn) * p.K + kg * 4) * 4 : MH_OOB;
        }
    }
    int l_tap = 0, l_c32 = 0;                        // cursor of the next chunk pair to LOAD (scalar)
    int l_base[2], l_krem[2];                        // (tap, chunk) slab offsets of the tile being loaded
    auto load_begin = [&]() {
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const bool in = l_tap < 9;
            // forward: slab (tap, c32) starts at row tap*K + c32*32 of the [9*K][N] matrix; dgrad: column c32*32 of w[tap][n][:]
            l_base[ch] = !DGRAD ? ((l_tap * p.K + l_c32 * 32) * p.N) * 4 : (l_tap * p.N * p.K + l_c32 * 32) * 4;
            l_krem[ch] = in ? p.K - l_c32 * 32 : 0;
            if (++l_c32 == g.CPT) { l_c32 = 0; ++l_tap; }
        }
    };
    // ONE 16-byte load (op o of NB) of the tile opened by load_begin()
    auto load_op = [&](float4 (&rb)[NB], int o) {
        const int jj = DGRAD ? o : o >> 2, r = DGRAD ? 0 : o & 3;
        const int bs = vch[jj] ? l_base[1] : l_base[0], kr = vch[jj] ? l_krem[1] : l_krem[0];
        rb[o] = mh_buf_load4(rs_w, (voff[jj] != MH_OOB && vk[jj] + r < kr) ? voff[jj] + bs + r * p.N * 4 : MH_OOB);   // dgrad: K % 4 == 0 (vecB)
    };
    // ONE 8-byte LDS store (piece o of NB): forward = column r of the transposed 4x4 unit, dgrad = one float4 along k
    auto store_op = [&](int buf, const float4 (&rb)[NB], int o) {
        unsigned short* Bb = Bh + buf * (BN * LSB);
        if (!DGRAD) {
            const int jj = o >> 2, r = o & 3;
            const int wi = (tid + NTH * jj) & 255;
            const int n4 = wi % (BN / 4), kq = wi / (BN / 4);
            if (kq < 8) {
                const int kk = vch[jj] * 32 + kq * 4;
                const float4 v0 = rb[4 * jj], v1 = rb[4 * jj + 1], v2 = rb[4 * jj + 2], v3 = rb[4 * jj + 3];
                unsigned short* dst = Bb + (n4 * 4 + r) * LSB + ((((kk >> 3) ^ n4) & 7) << 3) + (kk & 7);
                const float e0 = r == 0 ? v0.x : r == 1 ? v0.y : r == 2 ? v0.z : v0.w, e1 = r == 0 ? v1.x : r == 1 ? v1.y : r == 2 ? v1.z : v1.w;
                const float e2 = r == 0 ? v2.x : r == 1 ? v2.y : r == 2 ? v2.z : v2.w, e3 = r == 0 ? v3.x : r == 1 ? v3.y : r == 2 ? v3.z : v3.w;
                if constexpr (X3) {
                    uint2 hi, lo;
                    mh_split_bf16x2(e0, e1, hi.x, lo.x);
                    mh_split_bf16x2(e2, e3, hi.y, lo.y);
                    *reinterpret_cast<uint2*>(dst) = hi;
                    *reinterpret_cast<uint2*>(dst + BLO) = lo;
                } else
                *reinterpret_cast<uint2*>(dst) = make_uint2(mh_pack_bf16(e0, e1), mh_pack_bf16(e2, e3));
            }
        } else {
            const int q = tid + NTH * o;
            const int rem = q - vch[o] * (8 * BN);
            const int nn = rem >> 3, kg = rem & 7;
            *reinterpret_cast<uint2*>(Bb + nn * LSB + vch[o] * 32 + kg * 4) = make_uint2(mh_pack_bf16(rb[o].x, rb[o].y), mh_pack_bf16(rb[o].z, rb[o].w));
        }
    };
    auto load_b = [&](float4 (&rb)[NB]) {
        load_begin();
#pragma unroll
        for (int o = 0; o < NB; ++o) load_op(rb, o);
    };
    auto store_b = [&](int buf, const float4 (&rb)[NB]) {
#pragma unroll
        for (int o = 0; o < NB; ++o) store_op(buf, rb, o);
    };

    load_b(rb0);                                     // tile 0
    load_b(rb1);                                     // tile 1

    // ---- stage the input patch once (bf16): U independent 16-byte loads in flight per thread ----------------------------
    if (DGRAD && !X3 && p.in_shadow) {
        // the producer left a bf16 copy of this tensor (the operand of the streamed filter gradient: pixel stride KP, zero padded): 8 channels
        // per 16-byte load, no conversion, half the bytes.  Bit-identical to the fp32 path (the same round-to-nearest-even, done earlier).
        constexpr int U = 6;
        const int kp8 = g.KP >> 3;
        const int items = (g.dbg & 2) ? 0 : (TH + 2) * PW * kp8;
        const __amdgpu_buffer_rsrc_t rs_sh = mh_make_rsrc(p.in_shadow, p.in_shadow_bytes);
        for (int q0 = tid; q0 < items; q0 += NTH * U) {
            u32x4 v[U];
            int lo[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = q0 + u * NTH;
                const int ppx = (int)__umulhi((unsigned)q, g.mul_kp8), c8 = q - ppx * kp8;        // q / kp8 (mul = ceil(2^32 / kp8), q < 2^16)
                const int pi = ppx / PW, pj = ppx - pi * PW;
                const int iy = y00 + (pi - 1) * d, ix = x00 + (pj - 1) * d;
                const bool ok = q < items && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
                v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_sh, ok ? (((b * p.Hi + iy) * p.Wi + ix) * g.KP + c8 * 8) * 2 : MH_OOB, 0, 0);
                lo[u] = q < items ? ppx * g.PS + c8 * 8 : -1;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (lo[u] >= 0) *reinterpret_cast<u32x4*>(Ph + lo[u]) = v[u];
        }
    } else {
        constexpr int U = NTH == 512 ? 12 : 16;
        const int kp4 = g.KP >> 2;
        const int items = (g.dbg & 2) ? 0 : (TH + 2) * PW * kp4;
        // item q = (patch pixel (pi, pj), 4-channel group c4); q advances by NTH per load: the cursor follows incrementally
        // (no division in the loop); one cursor for the loads, a second one for the LDS stores
        const int dpp = NTH / kp4, dc4 = NTH - dpp * kp4;
        const int dpi = dpp / PW, dpj = dpp - dpi * PW;
        const int ppx0 = (int)(((float)tid + 0.5f) * g.inv_kp4);
        int c4 = tid - ppx0 * kp4, pi = ppx0 / PW, pj = ppx0 - pi * PW;
        int iy = y00 + (pi - 1) * d, ix = x00 + (pj - 1) * d;
        int off = (((b * p.Hi + iy) * p.Wi + ix) * p.in_ld + c4 * 4) * 4;                // global byte offset of the item
        int lds = ppx0 * g.PS + c4 * 4;                                                   // its LDS position (halfs)
        int c4s = c4, pis = pi, pjs = pj;
        // uniform steps: adds only inside the loop (v_mul_lo_u32 is quarter rate)
        const int s_col = d * p.in_ld * 4, s_row = d * p.Wi * p.in_ld * 4;
        const int st_off = dpi * s_row + dpj * s_col + dc4 * 16, st_iy = dpi * d, st_ix = dpj * d;
        const int w_off = s_col - kp4 * 16, r_off = s_row - PW * s_col, r_ix = PW * d;
        const int st_lds = dpp * g.PS + dc4 * 4, w_lds = g.PS - kp4 * 4;
        for (int q0 = tid; q0 < items; q0 += NTH * U) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool ok = (pi < TH + 2) && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi && (c4 < p.G);
                v[u] = mh_buf_load4(rs_in, ok ? off : MH_OOB);
                c4 += dc4; pj += dpj; pi += dpi; iy += st_iy; ix += st_ix; off += st_off;
                if (c4 >= kp4) { c4 -= kp4; ++pj; ix += d; off += w_off; }
                if (pj >= PW) { pj -= PW; ++pi; iy += d; ix -= r_ix; off += r_off; }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (pis < TH + 2) {
                    float4 w = v[u];
                    w.y = (c4s * 4 + 1 < p.K) ? w.y : 0.f;       // the row padding between K and in_ld is not ours to trust
                    w.z = (c4s * 4 + 2 < p.K) ? w.z : 0.f;
                    w.w = (c4s * 4 + 3 < p.K) ? w.w : 0.f;
                    if constexpr (X3) {
                        uint2 hi, lo;
                        mh_split_bf16x2(w.x, w.y, hi.x, lo.x);
                        mh_split_bf16x2(w.z, w.w, hi.y, lo.y);
                        *reinterpret_cast<uint2*>(Ph + lds) = hi;
                        *reinterpret_cast<uint2*>(Ph + g.patch_halfs + lds) = lo;
                    } else
                    *reinterpret_cast<uint2*>(Ph + lds) = make_uint2(mh_pack_bf16(w.x, w.y), mh_pack_bf16(w.z, w.w));
                }
                c4s += dc4; pjs += dpj; pis += dpi; lds += st_lds;
                if (c4s >= kp4) { c4s -= kp4; ++pjs; lds += w_lds; }
                if (pjs >= PW) { pjs -= PW; ++pis; }
            }
        }
    }
    store_b(0, rb0);
    // vmcnt(0): nothing may be in flight at the loop headers below.  The tile-1 loads have landed long ago (they were issued
    // before the patch loads), but if the scoreboard of hipcc's waitcnt pass still carries them into the loop, every
    // register it recycles inside the loop gets a conservative s_waitcnt vmcnt at the header -- a full drain per iteration.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // K walk, hand-interleaved.  Tile t = chunks 2t, 2t+1 out of LDS buffer t&1; tile t+1 waits in one register stage
    // and tile t+2 is loaded into the other.  All waves of the workgroup meet at one barrier per tile, so they run the same
    // phase at the same time: issued phase by phase (loads, LDS reads, MFMAs, LDS stores) the matrix cores idle through
    // every other phase.  Instead ONE other operation is issued after each MFMA and sched_barrier pins that order:
    //   MFMAs of chunk 2t   : + LDS reads of the fragments of chunk 2t+1, then the first half of the tile-(t+2) loads
    //   MFMAs of chunk 2t+1 : + the other half of the loads, then convert + LDS-store tile t+1 into the other buffer
    //   barrier, LDS reads of the fragments of chunk 2t+2 (the only exposed latency), next tile.
    // Loads past the walk read zeros; a chunk past the walk (odd chunk counts) multiplies a zero weight tile.
    const int ntile = (g.dbg & 1) ? 0 : (g.nchunk + 1) >> 1;
    const unsigned short* const Pw = Ph + ((wm * MT) * PW + li) * g.PS + lq * 8;          // tile row wm*MT, column li
    int c_tap = 0, c_c32 = 0;                        // cursor of the next chunk whose fragments are READ (scalar)
    auto next_a = [&]() -> const unsigned short* {
        const int tap = c_tap < 9 ? c_tap : 8;
        const int ky = tap / 3, kx = tap - ky * 3;
        const int oy = DGRAD ? 2 - ky : ky, ox = DGRAD ? 2 - kx : kx;
        const unsigned short* Ab = Pw + (oy * PW + ox) * g.PS + c_c32 * 32;
        if (++c_c32 == g.CPT) { c_c32 = 0; ++c_tap; }
        return Ab;
    };
    constexpr int NF = MT + NT, MM = MT * NT;
    // fragment read f of chunk `ch` of buffer `buf`: f < MT -> A row block f, else B column block f - MT
    auto frag_op = [&](int buf, int ch, const unsigned short* Ab, u32x4 (&fa)[MT], u32x4 (&fb)[NT], int f) {
        if (f < MT) {
            fa[f] = *reinterpret_cast<const u32x4*>(Ab + f * PW * g.PS);
        } else {
            const int j = f - MT;
            const unsigned short* Bb = Bh + buf * (BN * LSB) + (wn * NT * 16 + li) * LSB;
            if (!DGRAD) fb[j] = *reinterpret_cast<const u32x4*>(Bb + j * 16 * LSB + ((((ch * 4 + lq) ^ ((wn * NT * 16 + j * 16 + li) >> 2)) & 7) << 3));
            else fb[j] = *reinterpret_cast<const u32x4*>(Bb + j * 16 * LSB + ch * 32 + lq * 8);
        }
    };
    constexpr int NL0 = NB / 2;                      // loads issued with the first chunk
    constexpr int OPS0 = NF + NL0, OPS1 = (NB - NL0) + NB;
    u32x4 fa0[MT], fb0[NT], fa1[MT], fb1[NT];
    // one tile: multiply out of `buf`, load into rbl, store rbs into buf^1
    auto tile = [&](int buf, float4 (&rbl)[NB], const float4 (&rbs)[NB]) {
        load_begin();
        const unsigned short* Ab1 = next_a();
#pragma unroll
        for (int m = 0; m < MM; ++m) {
            acc[m / NT][m % NT] = mh_mfma_bf16(fa0[m / NT], fb0[m % NT], acc[m / NT][m % NT]);
#pragma unroll
            for (int o = m * OPS0 / MM; o < (m + 1) * OPS0 / MM; ++o) {
                if (o < NF) frag_op(buf, 1, Ab1, fa1, fb1, o);
                else load_op(rbl, o - NF);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int m = 0; m < MM; ++m) {
            acc[m / NT][m % NT] = mh_mfma_bf16(fa1[m / NT], fb1[m % NT], acc[m / NT][m % NT]);
#pragma unroll
            for (int o = m * OPS1 / MM; o < (m + 1) * OPS1 / MM; ++o) {
                if (o < NB - NL0) load_op(rbl, NL0 + o);
                else store_op(buf ^ 1, rbs, o - (NB - NL0));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        const unsigned short* Ab0 = next_a();
#pragma unroll
        for (int f = 0; f < NF; ++f) frag_op(buf ^ 1, 0, Ab0, fa0, fb0, f);
    };
    // ---- split-bf16 walk: the same tile schedule with the lo planes alongside -- 2 x NF fragment reads and 3 x MM MFMAs per
    // chunk (t = 0: lo(A)*hi(B), 1: hi(A)*lo(B), 2: hi(A)*hi(B); t outermost so consecutive MFMAs hit different accumulators)
    u32x4 la0[X3 ? MT : 1], lb0[X3 ? NT : 1], la1[X3 ? MT : 1], lb1[X3 ? NT : 1];
    auto frag_x3 = [&](int buf, int ch, const unsigned short* Ab, u32x4 (&fa)[MT], u32x4 (&fb)[NT], u32x4 (&la)[X3 ? MT : 1], u32x4 (&lb)[X3 ? NT : 1], int f) {
        if constexpr (X3) {
            if (f < NF) { frag_op(buf, ch, Ab, fa, fb, f); return; }
            const int f2 = f - NF;
            if (f2 < MT) {
                la[f2] = *reinterpret_cast<const u32x4*>(Ab + g.patch_halfs + f2 * PW * g.PS);
            } else {
                const int j = f2 - MT;
                const unsigned short* Bb = Bh + BLO + buf * (BN * LSB) + (wn * NT * 16 + li) * LSB;
                lb[j] = *reinterpret_cast<const u32x4*>(Bb + j * 16 * LSB + ((((ch * 4 + lq) ^ ((wn * NT * 16 + j * 16 + li) >> 2)) & 7) << 3));
            }
        }
    };
    auto mfma_x3 = [&](int m, const u32x4 (&fa)[MT], const u32x4 (&fb)[NT], const u32x4 (&la)[X3 ? MT : 1], const u32x4 (&lb)[X3 ? NT : 1]) {
        if constexpr (X3) {
            const int t = m / MM, mm = m % MM, i = mm / NT, j = mm % NT;
            acc[i][j] = mh_mfma_bf16(t == 0 ? la[i] : fa[i], t == 1 ? lb[j] : fb[j], acc[i][j]);
        }
    };
    auto tile_x3 = [&](int buf, float4 (&rbl)[NB], const float4 (&rbs)[NB]) {
        constexpr int M3 = 3 * MM, XO0 = 2 * NF + NL0;
        load_begin();
        const unsigned short* Ab1 = next_a();
#pragma unroll
        for (int m = 0; m < M3; ++m) {
            mfma_x3(m, fa0, fb0, la0, lb0);
#pragma unroll
            for (int o = m * XO0 / M3; o < (m + 1) * XO0 / M3; ++o) {
                if (o < 2 * NF) frag_x3(buf, 1, Ab1, fa1, fb1, la1, lb1, o);
                else load_op(rbl, o - 2 * NF);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int m = 0; m < M3; ++m) {
            mfma_x3(m, fa1, fb1, la1, lb1);
#pragma unroll
            for (int o = m * OPS1 / M3; o < (m + 1) * OPS1 / M3; ++o) {
                if (o < NB - NL0) load_op(rbl, NL0 + o);
                else store_op(buf ^ 1, rbs, o - (NB - NL0));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        const unsigned short* Ab0 = next_a();
#pragma unroll
        for (int f = 0; f < 2 * NF; ++f) frag_x3(buf ^ 1, 0, Ab0, fa0, fb0, la0, lb0, f);
    };
    if constexpr (X3 && CPTC == 0) {
        {
            const unsigned short* Ab0 = next_a();
#pragma unroll
            for (int f = 0; f < 2 * NF; ++f) frag_x3(0, 0, Ab0, fa0, fb0, la0, lb0, f);
        }
        for (int t = 0; t + 1 < ntile; t += 2) {
            tile_x3(0, rb0, rb1);
            tile_x3(1, rb1, rb0);
        }
        if (ntile & 1) tile_x3(0, rb0, rb1);
    } else if constexpr (CPTC > 0) {
        // ---- specialised walk: K == 32 * CPTC.  Patch row stride and chunk offsets are compile-time, so the fragment reads
        // are `ds_read_b128 v, vbase offset:imm` off one VGPR per tap; a weight load is one v_add (thread-constant offset +
        // scalar tile base; a base of 2^31 pushes every lane out of range = zeros past the walk) + the buffer load.
        constexpr int PSC = 32 * CPTC + 16;
        int vofs[NB];
#pragma unroll
        for (int o = 0; o < NB; ++o) {
            const int jj = DGRAD ? o : o >> 2, r = DGRAD ? 0 : o & 3;
            if (!DGRAD) vofs[o] = voff[jj] != MH_OOB ? voff[jj] + (vch[jj] * 32 + r) * p.N * 4 : MH_OOB;
            else vofs[o] = voff[jj] != MH_OOB ? voff[jj] + vch[jj] * 128 : MH_OOB;
        }
        auto tap_ptr = [&](int tap) -> const unsigned short* {
            const int tc = tap < 9 ? tap : 8;
            const int ky = tc / 3, kx = tc - ky * 3;
            const int oy = DGRAD ? 2 - ky : ky, ox = DGRAD ? 2 - kx : kx;
            return Ph + ((wm * MT + oy) * PW + li + ox) * PSC + lq * 8;
        };
        // byte offset of the weight slab of chunk c (even) of tap `tap`; past the walk: 2^31
        auto w_base = [&](int tap, int c) -> unsigned {
            if (tap >= 9) return 0x80000000u;
            return !DGRAD ? (unsigned)(((tap * CPTC + c) * 32 * p.N) * 4) : (unsigned)((tap * p.N * p.K + c * 32) * 4);
        };
        auto frag_c = [&](int buf, int ch, const unsigned short* Ab, u32x4 (&fa)[MT], u32x4 (&fb)[NT], int f) {
            if (f < MT) fa[f] = *reinterpret_cast<const u32x4*>(Ab + f * PW * PSC);
            else frag_op(buf, ch, Ab, fa, fb, f);
        };
        // split-bf16: fragment read f of 2*NF -- hi planes first, then the lo planes (patch lo plane = a compile-time offset)
        constexpr int PLO = (TH + 2) * PW * PSC;
        auto frag_c3 = [&](int buf, int ch, const unsigned short* Ab, u32x4 (&fa)[MT], u32x4 (&fb)[NT], u32x4 (&la)[X3 ? MT : 1], u32x4 (&lb)[X3 ? NT : 1], int f) {
            if constexpr (X3) {
                if (f < NF) { frag_c(buf, ch, Ab, fa, fb, f); return; }
                const int f2 = f - NF;
                if (f2 < MT) la[f2] = *reinterpret_cast<const u32x4*>(Ab + PLO + f2 * PW * PSC);
                else frag_x3(buf, ch, Ab, fa, fb, la, lb, f);
            }
        };
        auto tile_c3 = [&](int buf, float4 (&rbl)[NB], const float4 (&rbs)[NB], const unsigned short* A1, const unsigned short* An, unsigned base) {
            if constexpr (X3) {
                constexpr int M3 = 3 * MM, XO0 = 2 * NF + NL0;
#pragma unroll
                for (int m = 0; m < M3; ++m) {
                    mfma_x3(m, fa0, fb0, la0, lb0);
#pragma unroll
                    for (int o = m * XO0 / M3; o < (m + 1) * XO0 / M3; ++o) {
                        if (o < 2 * NF) frag_c3(buf, 1, A1, fa1, fb1, la1, lb1, o);
                        else rbl[o - 2 * NF] = mh_buf_load4(rs_w, (int)((unsigned)vofs[o - 2 * NF] + base));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int m = 0; m < M3; ++m) {
                    mfma_x3(m, fa1, fb1, la1, lb1);
#pragma unroll
                    for (int o = m * OPS1 / M3; o < (m + 1) * OPS1 / M3; ++o) {
                        if (o < NB - NL0) rbl[NL0 + o] = mh_buf_load4(rs_w, (int)((unsigned)vofs[NL0 + o] + base));
                        else store_op(buf ^ 1, rbs, o - (NB - NL0));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                __syncthreads();
#pragma unroll
                for (int f = 0; f < 2 * NF; ++f) frag_c3(buf ^ 1, 0, An, fa0, fb0, la0, lb0, f);
            }
        };
        // multiply the tile in `buf` (fragments of its first chunk are in fa0/fb0), load the tile at `base` into rbl,
        // store rbs into buf^1; A1 / An = patch pointers of this tile's second chunk / the next tile's first chunk
        auto tile_c = [&](int buf, float4 (&rbl)[NB], const float4 (&rbs)[NB], const unsigned short* A1, const unsigned short* An, unsigned base) {
            if constexpr (X3) { tile_c3(buf, rbl, rbs, A1, An, base); return; }
#pragma unroll
            for (int m = 0; m < MM; ++m) {
                acc[m / NT][m % NT] = mh_mfma_bf16(fa0[m / NT], fb0[m % NT], acc[m / NT][m % NT]);
#pragma unroll
                for (int o = m * OPS0 / MM; o < (m + 1) * OPS0 / MM; ++o) {
                    if (o < NF) frag_c(buf, 1, A1, fa1, fb1, o);
                    else rbl[o - NF] = mh_buf_load4(rs_w, (int)((unsigned)vofs[o - NF] + base));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int m = 0; m < MM; ++m) {
                acc[m / NT][m % NT] = mh_mfma_bf16(fa1[m / NT], fb1[m % NT], acc[m / NT][m % NT]);
#pragma unroll
                for (int o = m * OPS1 / MM; o < (m + 1) * OPS1 / MM; ++o) {
                    if (o < NB - NL0) rbl[NL0 + o] = mh_buf_load4(rs_w, (int)((unsigned)vofs[NL0 + o] + base));
                    else store_op(buf ^ 1, rbs, o - (NB - NL0));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
#pragma unroll
            for (int f = 0; f < NF; ++f) frag_c(buf ^ 1, 0, An, fa0, fb0, f);
        };
        const unsigned short* a_cur = tap_ptr(0);
        if constexpr (X3) {
#pragma unroll
            for (int f = 0; f < 2 * NF; ++f) frag_c3(0, 0, a_cur, fa0, fb0, la0, lb0, f);
        } else {
#pragma unroll
            for (int f = 0; f < NF; ++f) frag_c(0, 0, a_cur, fa0, fb0, f);
        }
        if (!(g.dbg & 1)) {
            if constexpr (CPTC == 4) {
                // two tiles per tap: chunks (0, 1) out of buffer 0, chunks (2, 3) out of buffer 1; loads run one tap ahead
                for (int tap = 0; tap < 9; ++tap) {
                    const unsigned short* a_nxt = tap_ptr(tap + 1);
                    tile_c(0, rb0, rb1, a_cur + 32, a_cur + 64, w_base(tap + 1, 0));
                    tile_c(1, rb1, rb0, a_cur + 96, a_nxt, w_base(tap + 1, 2));
                    a_cur = a_nxt;
                }
            } else {
                // one tile per tap: even taps out of buffer 0, odd taps out of buffer 1; loads run two taps ahead
                for (int tap = 0; tap < 8; tap += 2) {
                    const unsigned short* a_b = tap_ptr(tap + 1);
                    const unsigned short* a_c = tap_ptr(tap + 2);
                    tile_c(0, rb0, rb1, a_cur + 32, a_b, w_base(tap + 2, 0));
                    tile_c(1, rb1, rb0, a_b + 32, a_c, w_base(tap + 3, 0));
                    a_cur = a_c;
                }
                tile_c(0, rb0, rb1, a_cur + 32, a_cur, 0x80000000u);
            }
        }
    } else {
    {
        const unsigned short* Ab0 = next_a();
#pragma unroll
        for (int f = 0; f < NF; ++f) frag_op(0, 0, Ab0, fa0, fb0, f);
    }
    // (an odd last tile is peeled: a skip path inside the loop would reach the loop header with the OTHER register stage
    //  in flight, and the waitcnt pass then drains the loads at every header -- seen as s_waitcnt vmcnt(1) in the ISA)
    for (int t = 0; t + 1 < ntile; t += 2) {
        tile(0, rb0, rb1);                           // multiply tile t, load t+2, store t+1
        tile(1, rb1, rb0);                           // multiply tile t+1, load t+3, store t+2
    }
    if (ntile & 1) tile(0, rb0, rb1);
    }

    // ---- epilogue: accumulator tile through LDS, then bias + leaky (+ accumulate) (+ leaky-grad mask), 16-byte rows --------
    constexpr int CS = BN + 4;
    float* const Cs = smem_all;                       // [BM][CS], over the patch / weight tiles (all reads are behind the last barrier)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Cs[((wm * MT + i) * 16 + lq * 4 + r) * CS + wn * NT * 16 + j * 16 + li] = acc[i][j][r];
    __syncthreads();
    constexpr int C4 = BN / 4;
    constexpr int RP = NTH / C4;
    constexpr int PASSES = (BM + RP - 1) / RP;
    const __amdgpu_buffer_rsrc_t rs_out = mh_make_rsrc(p.out, p.out_bytes);
    const __amdgpu_buffer_rsrc_t rs_mask = mh_make_rsrc(p.mask_ref ? p.mask_ref : p.out, p.mask_ref ? p.mask_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rs_msh = mh_make_rsrc(p.mask_shadow ? (const void*)p.mask_shadow : (const void*)p.out, p.mask_shadow ? p.mask_shadow_bytes : 0u);
    const int c4 = tid % C4;
    const int n = n0 + c4 * 4;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && n < p.N) bv = *reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll 4
    for (int ps = 0; ps < PASSES; ++ps) {
        const int row = tid / C4 + ps * RP;
        const int rr = row < BM ? row : 0;
        const int y = y00 + (rr >> 4) * d, x = x00 + (rr & 15) * d;
        const bool ok = (tid < RP * C4) && (row < BM) && (y < p.Ho) && (x < p.Wo) && (n < p.N);
        const int m = (b * p.Ho + y) * p.Wo + x;
        float4 v = *reinterpret_cast<const float4*>(&Cs[rr * CS + c4 * 4]);
        const int ooff = ok ? (m * p.out_ld + n) * 4 : MH_OOB;
        float4 old = make_float4(0.f, 0.f, 0.f, 0.f), mk = make_float4(1.f, 1.f, 1.f, 1.f);
        if (p.accumulate) old = mh_buf_load4(rs_out, ooff);
        if (DGRAD && p.mask_shadow) {
            // the mask only asks for the sign: 8 bytes of the activation's bf16 shadow instead of 16 of the fp32 tensor (bf16 keeps sign and zero)
            const u32x2 m2 = __builtin_amdgcn_raw_buffer_load_b64(rs_msh, ok ? (m * p.mask_shadow_ld + n) * 2 : MH_OOB, 0, 0);
            const unsigned lo2 = m2[0], hi2 = m2[1];
            mk.x = __builtin_bit_cast(float, lo2 << 16); mk.y = __builtin_bit_cast(float, lo2 & 0xffff0000u);
            mk.z = __builtin_bit_cast(float, hi2 << 16); mk.w = __builtin_bit_cast(float, hi2 & 0xffff0000u);
        } else
        if (p.mask_ref) mk = mh_buf_load4(rs_mask, ok ? (m * p.mask_ld + n) * 4 : MH_OOB);
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        if (p.alpha != 1.0f) {
            v.x = v.x > 0.f ? v.x : p.alpha * v.x; v.y = v.y > 0.f ? v.y : p.alpha * v.y;
            v.z = v.z > 0.f ? v.z : p.alpha * v.z; v.w = v.w > 0.f ? v.w : p.alpha * v.w;
        }
        v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
        if (p.mask_ref) {
            v.x *= (mk.x > 0.f || n + 0 < p.mask_c0 || n + 0 >= p.mask_c1) ? 1.0f : p.mask_alpha;
            v.y *= (mk.y > 0.f || n + 1 < p.mask_c0 || n + 1 >= p.mask_c1) ? 1.0f : p.mask_alpha;
            v.z *= (mk.z > 0.f || n + 2 < p.mask_c0 || n + 2 >= p.mask_c1) ? 1.0f : p.mask_alpha;
            v.w *= (mk.w > 0.f || n + 3 < p.mask_c0 || n + 3 >= p.mask_c1) ? 1.0f : p.mask_alpha;
        }
        if (ok && !(DGRAD && p.no_f32_out)) *reinterpret_cast<float4*>(p.out + (int64_t)m * p.out_ld + n) = v;
        if (ok && p.shadow) *reinterpret_cast<uint2*>(p.shadow + (int64_t)m * p.shadow_ld + n) = make_uint2(mh_pack_bf16(v.x, v.y), mh_pack_bf16(v.z, v.w));
    }
}


// ---- "fragment bank" variant (forward, stride-1 3x3): the weights are NOT staged through LDS.  mh_pack_weights (below) rewrites a
// filter bank once per step into the exact register image of the MFMA B operand -- bank[(tap * CPT + chunk)][16-column tile][plane]
// [lane][8 bf16], plane = hi (and lo for split-bf16) -- so a wave fetches a fragment with ONE coalesced 1 KB buffer load per plane
// (the waves of a workgroup that share a column tile hit L1, the workgroups of an XCD hit its L2).  What that removes from the K walk
// of conv_patch_kernel: the fp32 weight loads + hi/lo conversion + LDS stores of every workgroup, half of the LDS fragment reads
// (LDS bound: profiles/r02_experiments.txt #6) and the per-tile barrier -- the patch is read-only after staging, so the waves run
// free.  Three register stages of B (prefetch distance 2-3 chunks) and of A (LDS, distance 1).
template <int WM, int WN, int MT, int NT, bool X3>
__global__ __launch_bounds__(WM * WN * 64) __attribute__((amdgpu_waves_per_eu(1, 2))) void conv_bank_kernel(ConvArgs p, PatchGeo g) {
    constexpr int NTH = WM * WN * 64;
    constexpr int BM = WM * MT * 16, BN = WN * NT * 16;
    constexpr int TH = BM / 16;
    constexpr int PL = X3 ? 2 : 1;
    HIP_DYNAMIC_SHARED(float, smem_all)
    unsigned short* const Ph = reinterpret_cast<unsigned short*>(smem_all);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, lq = lane >> 4;
    const int d = p.dil;

    const int lin = mh_xcd_remap(blockIdx.x, g.nwg);
    int tile_n, ttx, tty, cx, cy, b;
    mh_decode_tile(lin, g.dec, tile_n, ttx, tty, cx, cy, b);
    const int n0 = tile_n * BN;
    const int y00 = cy + d * (tty * TH), x00 = cx + d * (ttx * 16);

    const __amdgpu_buffer_rsrc_t rs_in = mh_make_rsrc(p.in, p.in_bytes);
    const __amdgpu_buffer_rsrc_t rs_b = mh_make_rsrc(p.wb, p.wb_bytes);

    // ---- B fragments: thread-constant offset of column tile j (plane 0) + chunk * stride_b ------------------------------
    const int np16 = (p.N + 15) >> 4;
    const int stride_b = np16 * PL * 1024;
    int voff_b[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int nt = tile_n * (BN / 16) + wn * NT + j;
        voff_b[j] = nt < np16 ? nt * PL * 1024 + lane * 16 : MH_OOB;
    }
    u32x4 fb[3][NT][PL], fa[3][MT][PL];
    const int nchunk = (g.dbg & 1) ? 0 : g.nchunk;                    // 9 * CPT: a multiple of 3
    auto issue_b = [&](u32x4 (&f)[NT][PL], int q) {
        const int base = q < nchunk ? q * stride_b : MH_OOB;         // past the walk: out of range = zeros
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int pl = 0; pl < PL; ++pl)
                f[j][pl] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, (int)((unsigned)voff_b[j] + (unsigned)base) + pl * 1024, 0, 0);
    };
    issue_b(fb[0], 0);
    issue_b(fb[1], 1);
    issue_b(fb[2], 2);

    // ---- stage the input patch once (as conv_patch_kernel) ---------------------------------------------------------------
    {
        constexpr int U = NTH == 512 ? 12 : 16;
        const int kp4 = g.KP >> 2;
        const int items = (g.dbg & 2) ? 0 : (TH + 2) * PW * kp4;
        const int dpp = NTH / kp4, dc4 = NTH - dpp * kp4;
        const int dpi = dpp / PW, dpj = dpp - dpi * PW;
        const int ppx0 = (int)(((float)tid + 0.5f) * g.inv_kp4);
        int c4 = tid - ppx0 * kp4, pi = ppx0 / PW, pj = ppx0 - pi * PW;
        int iy = y00 + (pi - 1) * d, ix = x00 + (pj - 1) * d;
        int off = (((b * p.Hi + iy) * p.Wi + ix) * p.in_ld + c4 * 4) * 4;
        int lds = ppx0 * g.PS + c4 * 4;
        int c4s = c4, pis = pi, pjs = pj;
        const int s_col = d * p.in_ld * 4, s_row = d * p.Wi * p.in_ld * 4;
        const int st_off = dpi * s_row + dpj * s_col + dc4 * 16, st_iy = dpi * d, st_ix = dpj * d;
        const int w_off = s_col - kp4 * 16, r_off = s_row - PW * s_col, r_ix = PW * d;
        const int st_lds = dpp * g.PS + dc4 * 4, w_lds = g.PS - kp4 * 4;
        for (int q0 = tid; q0 < items; q0 += NTH * U) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool ok = (pi < TH + 2) && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi && (c4 < p.G);
                v[u] = mh_buf_load4(rs_in, ok ? off : MH_OOB);
                c4 += dc4; pj += dpj; pi += dpi; iy += st_iy; ix += st_ix; off += st_off;
                if (c4 >= kp4) { c4 -= kp4; ++pj; ix += d; off += w_off; }
                if (pj >= PW) { pj -= PW; ++pi; iy += d; ix -= r_ix; off += r_off; }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (pis < TH + 2) {
                    float4 w = v[u];
                    w.y = (c4s * 4 + 1 < p.K) ? w.y : 0.f;
                    w.z = (c4s * 4 + 2 < p.K) ? w.z : 0.f;
                    w.w = (c4s * 4 + 3 < p.K) ? w.w : 0.f;
                    if constexpr (X3) {
                        uint2 hi, lo;
                        mh_split_bf16x2(w.x, w.y, hi.x, lo.x);
                        mh_split_bf16x2(w.z, w.w, hi.y, lo.y);
                        *reinterpret_cast<uint2*>(Ph + lds) = hi;
                        *reinterpret_cast<uint2*>(Ph + g.patch_halfs + lds) = lo;
                    } else
                    *reinterpret_cast<uint2*>(Ph + lds) = make_uint2(mh_pack_bf16(w.x, w.y), mh_pack_bf16(w.z, w.w));
                }
                c4s += dc4; pjs += dpj; pis += dpi; lds += st_lds;
                if (c4s >= kp4) { c4s -= kp4; ++pjs; lds += w_lds; }
                if (pjs >= PW) { pjs -= PW; ++pis; }
            }
        }
    }
    __syncthreads();

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- K walk: chunk q = (tap q / CPT, 32 channels q % CPT); no barrier ----------------------------------------------
    const unsigned short* const Pw = Ph + ((wm * MT) * PW + li) * g.PS + lq * 8;
    int a_tap = 0, a_c = 0;                          // scalar cursor of the next chunk whose A fragments are read
    auto issue_a = [&](u32x4 (&f)[MT][PL]) {
        const int tap = a_tap < 9 ? a_tap : 8;
        const int ky = tap / 3, kx = tap - ky * 3;
        const unsigned short* Ab = Pw + (ky * PW + kx) * g.PS + a_c * 32;
        if (++a_c == g.CPT) { a_c = 0; ++a_tap; }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            f[i][0] = *reinterpret_cast<const u32x4*>(Ab + i * PW * g.PS);
            if constexpr (X3) f[i][1] = *reinterpret_cast<const u32x4*>(Ab + g.patch_halfs + i * PW * g.PS);
        }
    };
    constexpr int MM = MT * NT, M3 = (X3 ? 3 : 1) * MM;
    issue_a(fa[0]);
    // nothing in flight at the loop header the first time (the B stages landed behind the patch loads): otherwise hipcc's waitcnt pass
    // merges "unknown" into the header state and drains vmcnt(0) on every iteration
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (int q = 0; q < nchunk; q += 3) {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            constexpr int dummy = 0; (void)dummy;
            const int ns = (s + 1) % 3;
#pragma unroll
            for (int m = 0; m < M3; ++m) {
                const int t = X3 ? m / MM : 2, mm = m % MM, i = mm / NT, j = mm % NT;
                // split-bf16: lo(A)*hi(B), hi(A)*lo(B), hi(A)*hi(B) -- t outermost: consecutive MFMAs hit different accumulators
                acc[i][j] = mh_mfma_bf16(fa[s][i][t == 0 ? PL - 1 : 0], fb[s][j][t == 1 ? PL - 1 : 0], acc[i][j]);
                if (m == 0) issue_a(fa[ns]);                          // A fragments of the next chunk (LDS latency << one chunk of MFMAs)
                __builtin_amdgcn_sched_barrier(0);
            }
            issue_b(fb[s], q + s + 3);                                // this stage's registers are free again: chunk q+s+3
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();                                 // every wave is done with the patch: the accumulator tile goes over it

    constexpr int CS = BN + 4;
    float* const Cs = smem_all;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Cs[((wm * MT + i) * 16 + lq * 4 + r) * CS + wn * NT * 16 + j * 16 + li] = acc[i][j][r];
    __syncthreads();
    constexpr int C4 = BN / 4;
    constexpr int RP = NTH / C4;
    constexpr int PASSES = (BM + RP - 1) / RP;
    const __amdgpu_buffer_rsrc_t rs_out = mh_make_rsrc(p.out, p.out_bytes);
    const __amdgpu_buffer_rsrc_t rs_mask = mh_make_rsrc(p.mask_ref ? p.mask_ref : p.out, p.mask_ref ? p.mask_bytes : 0u);
    const int c4 = tid % C4;
    const int n = n0 + c4 * 4;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && n < p.N) bv = *reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll 4
    for (int ps = 0; ps < PASSES; ++ps) {
        const int row = tid / C4 + ps * RP;
        const int rr = row < BM ? row : 0;
        const int y = y00 + (rr >> 4) * d, x = x00 + (rr & 15) * d;
        const bool ok = (tid < RP * C4) && (row < BM) && (y < p.Ho) && (x < p.Wo) && (n < p.N);
        const int m = (b * p.Ho + y) * p.Wo + x;
        float4 v = *reinterpret_cast<const float4*>(&Cs[rr * CS + c4 * 4]);
        const int ooff = ok ? (m * p.out_ld + n) * 4 : MH_OOB;
        float4 old = make_float4(0.f, 0.f, 0.f, 0.f), mk = make_float4(1.f, 1.f, 1.f, 1.f);
        if (p.accumulate) old = mh_buf_load4(rs_out, ooff);
        if (p.mask_ref) mk = mh_buf_load4(rs_mask, ok ? (m * p.mask_ld + n) * 4 : MH_OOB);
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        if (p.alpha != 1.0f) {
            v.x = v.x > 0.f ? v.x : p.alpha * v.x; v.y = v.y > 0.f ? v.y : p.alpha * v.y;
            v.z = v.z > 0.f ? v.z : p.alpha * v.z; v.w = v.w > 0.f ? v.w : p.alpha * v.w;
        }
        v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
        if (p.mask_ref) {
            v.x *= (mk.x > 0.f || n + 0 < p.mask_c0 || n + 0 >= p.mask_c1) ? 1.0f : p.mask_alpha;
            v.y *= (mk.y > 0.f || n + 1 < p.mask_c0 || n + 1 >= p.mask_c1) ? 1.0f : p.mask_alpha;
            v.z *= (mk.z > 0.f || n + 2 < p.mask_c0 || n + 2 >= p.mask_c1) ? 1.0f : p.mask_alpha;
            v.w *= (mk.w > 0.f || n + 3 < p.mask_c0 || n + 3 >= p.mask_c1) ? 1.0f : p.mask_alpha;
        }
        if (ok) *reinterpret_cast<float4*>(p.out + (int64_t)m * p.out_ld + n) = v;
        if (ok && p.shadow) *reinterpret_cast<uint2*>(p.shadow + (int64_t)m * p.shadow_ld + n) = make_uint2(mh_pack_bf16(v.x, v.y), mh_pack_bf16(v.z, v.w));
    }
}

// ---- fragment-bank kernel of the SMALL layers (1/16 - 1/64 resolution: a few hundred output pixels, K = 9 x 32..224) ------------------
// There a launch is one exposed memory round trip after the other (profiles/r02_experiments.txt #15: 1.5 us prologue, 1.7 us first
// K-tile, ~1 us per further K-tile pair, 11 us in all for 17 MFLOP).  Here every byte the workgroup needs is requested in the first few
// hundred cycles: 16 waves share one 32-pixel x 32-column output tile and SPLIT THE REDUCTION -- wave w takes chunks w, w+16, w+32, (w+48)
// of the (tap, 32-channel) walk, fetches their weight fragments from the bank straight into registers (<= 4 chunks x 2 column tiles, all
// issued before anything else), the 4 x 18 input patch is staged once by all 1024 threads, then <= 16 (48) MFMAs per wave, the 16 partial
// tiles meet in LDS and thread e finishes output element e (bias, leaky, accumulate, leaky-gradient mask).
// DGRAD: in = dz, bank = mh_pack_weights(trans = 1) of the HWIO bank, taps mirrored.  PL = 2: split-bf16 (three MFMAs per product).
template <bool DGRAD, int PL>
__global__ __launch_bounds__(1024) void conv_bank_small_kernel(ConvArgs p, PatchGeo g) {
    constexpr int NTH = 1024, TH = 2, BM = 32, BN = 32, MT = 2, NT = 2, NCH = 4;
    HIP_DYNAMIC_SHARED(float, smem_all)
    unsigned short* const Ph = reinterpret_cast<unsigned short*>(smem_all);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int d = p.dil;
    const int st = DGRAD ? 1 : p.stride;             // forward: stride 1 (any dilation) or stride 2 (dilation 1): output pixel (i, j) reads patch (i*st + ky, j*st + kx)
    const int PC = st * 16 + 3 - st;                 // patch columns: 18 / 33 ; rows: st * TH + 3 - st = 4 / 5

    int lin;
    if (g.xuse < 8) {                                 // (uniform per workgroup: a whole block leaves before any barrier)
        const int xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3;
        lin = xcd * g.per + idx;
        if (xcd >= g.xuse || lin >= g.nwg) return;
    } else lin = mh_xcd_remap(blockIdx.x, g.nwg);
    int tile_n, ttx, tty, cx, cy, b;
    if (g.n_major) {
        const int tn = mh_fdiv(lin, g.f_pt);
        mh_decode_tile(lin - tn * (int)g.f_pt.d, g.dec, tile_n, ttx, tty, cx, cy, b);        // (dec.ntn = 1 here: tile_n comes back 0)
        tile_n = tn;
    } else mh_decode_tile(lin, g.dec, tile_n, ttx, tty, cx, cy, b);
    const int n0 = tile_n * BN;
    const int y00 = cy + d * (tty * TH), x00 = cx + d * (ttx * 16);

    const __amdgpu_buffer_rsrc_t rs_in = mh_make_rsrc(p.in, p.in_bytes);
    const __amdgpu_buffer_rsrc_t rs_b = mh_make_rsrc(p.wb, p.wb_bytes);

    // (the bias of this thread's output element: requested now, used in the epilogue -- a dependent global load there cost the launch a memory
    //  round trip behind the reduction: round 4, scripts/exp/node_floor.py)
    // Round 5: these three were CONDITIONAL global loads -- behind each `if` hipcc joins with s_waitcnt vmcnt(0), so the launch spent a whole memory round
    // trip on them before it had even requested its weight fragments (ISA of round 4's build; ~1 us of every 5 us small-layer node).  Range-checked buffer
    // loads now: unconditional, an absent operand is a descriptor of zero bytes (reads 0).
    const int my_n = n0 + (tid & 31);
    const int my_row = tid >> 5;
    const int my_y = y00 + (my_row >> 4) * d, my_x = x00 + (my_row & 15) * d;
    const bool my_ok = my_y < p.Ho && my_x < p.Wo && my_n < p.N;
    const int64_t my_m = ((int64_t)b * p.Ho + my_y) * p.Wo + my_x;
    const __amdgpu_buffer_rsrc_t rs_bias = mh_make_rsrc(p.bias ? (const void*)p.bias : (const void*)p.in, p.bias ? (unsigned)(p.N * 4) : 0u);
    const __amdgpu_buffer_rsrc_t rs_old = mh_make_rsrc(p.out, p.accumulate ? p.out_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rs_mk = mh_make_rsrc(p.mask_ref ? (const void*)p.mask_ref : (const void*)p.in, p.mask_ref ? p.mask_bytes : 0u);
    float my_bias, my_old, my_mk;          // (requested behind the first round of patch loads, below: issued first, an unlucky register reuse made hipcc wait for them
                                           //  in the middle of the fragment loads)
    // ---- all weight fragments of this wave's chunks: requested first ------------------------------------------------------
    const int np16 = (p.N + 15) >> 4;
    const int stride_b = np16 * PL * 1024;
    u32x4 fb[NCH][NT][PL];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int q = wave + 16 * c;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int nt = tile_n * (BN / 16) + j;
            const int off = (q < g.nchunk && nt < np16) ? q * stride_b + nt * PL * 1024 + lane * 16 : MH_OOB;
#pragma unroll
            for (int pl = 0; pl < PL; ++pl) fb[c][j][pl] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, off == MH_OOB ? MH_OOB : off + pl * 1024, 0, 0);
        }
    }

    // ---- stage the 4 x 18 input patch (bf16; hi + lo planes for PL = 2) ---------------------------------------------------
    // Round 5: straight-line, every load of the thread in flight before the first conversion.  The loop this replaces issued ONE load per iteration
    // behind an s_waitcnt vmcnt(0) (a 128-channel layer's 2304 patch vectors = three serial memory round trips, on top of the weight fragments'), and any
    // loop / branch here makes hipcc's waitcnt pass drain the fragment loads at its header: no loop (the launcher guarantees items <= ROUNDS * U * 1024),
    // no branch (offsets by select, the division by multiply-high with a select for the divisor 1).
    {
        constexpr int U = 4, ROUNDS = 2;
        const int kp4 = g.KP >> 2;
        const int items = (st * TH + 3 - st) * PC * kp4;
        auto fdiv = [](int n, const mh_fastdiv& f) { const int h = (int)__umulhi((unsigned)n, f.m); return f.d > 1 ? h : n; };
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
            if (rd > 0 && items <= rd * U * NTH) break;          // (uniform; nothing is in flight here)
            float4 wv[U];
            int ldsv[U], c4v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = tid + (rd * U + u) * NTH;
                const int pp = fdiv(q, g.f_kp4), c4 = q - pp * kp4;
                const int pi = fdiv(pp, g.f_pc), pj = pp - pi * PC;
                const int iy = y00 * st - p.pad_t + pi * d, ix = x00 * st - p.pad_l + pj * d;      // (stride 1: pad = dilation)
                const bool ok = q < items && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi && (c4 < p.G);
                int off = (((b * p.Hi + iy) * p.Wi + ix) * p.in_ld + c4 * 4) * 4;
                MH_KEEP_VGPR(off);                               // (materialised: the select below stays a v_cndmask, not a branch around the multiplies)
                wv[u] = mh_buf_load4(rs_in, ok ? off : MH_OOB);
                ldsv[u] = q < items ? pp * g.PS + c4 * 4 : -1;
                c4v[u] = c4;
            }
            if (rd == 0) {
                my_bias = mh_buf_load1(rs_bias, my_n < p.N ? my_n * 4 : MH_OOB);
                my_old = mh_buf_load1(rs_old, my_ok ? (int)((my_m * p.out_ld + my_n) * 4) : MH_OOB);
                my_mk = mh_buf_load1(rs_mk, my_ok ? (int)((my_m * p.mask_ld + my_n) * 4) : MH_OOB);      // (no mask: the value is not used)
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float4 w = wv[u];
                const int c4 = c4v[u];
                w.y = (c4 * 4 + 1 < p.K) ? w.y : 0.f;           // the row padding between K and in_ld is not ours to trust
                w.z = (c4 * 4 + 2 < p.K) ? w.z : 0.f;
                w.w = (c4 * 4 + 3 < p.K) ? w.w : 0.f;
                const int lds = ldsv[u];
                if (lds >= 0) {
                    if constexpr (PL == 2) {
                        uint2 hi, lo;
                        mh_split_bf16x2(w.x, w.y, hi.x, lo.x);
                        mh_split_bf16x2(w.z, w.w, hi.y, lo.y);
                        *reinterpret_cast<uint2*>(Ph + lds) = hi;
                        *reinterpret_cast<uint2*>(Ph + g.patch_halfs + lds) = lo;
                    } else
                    *reinterpret_cast<uint2*>(Ph + lds) = make_uint2(mh_pack_bf16(w.x, w.y), mh_pack_bf16(w.z, w.w));
                }
            }
        }
    }
    __syncthreads();

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned short* const Pw = Ph + (li * st) * g.PS + lq * 8;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int q = wave + 16 * c;
        if (q < g.nchunk) {                           // wave uniform
            const int tap = q / g.CPT, c32 = q - tap * g.CPT;
            const int ky = tap / 3, kx = tap - ky * 3;
            const int oy = DGRAD ? 2 - ky : ky, ox = DGRAD ? 2 - kx : kx;
            const unsigned short* Ab = Pw + (oy * PC + ox) * g.PS + c32 * 32;
            u32x4 fa[MT][PL];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int pl = 0; pl < PL; ++pl) fa[i][pl] = *reinterpret_cast<const u32x4*>(Ab + pl * g.patch_halfs + (i * st) * PC * g.PS);
#pragma unroll
            for (int t = (PL == 2 ? 0 : 2); t < 3; ++t)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = mh_mfma_bf16(fa[i][t == 0 ? PL - 1 : 0], fb[c][j][t == 1 ? PL - 1 : 0], acc[i][j]);
        }
    }
    __syncthreads();                                 // the patch is dead: the partial tiles go over it
    constexpr int CS = BN + 1;                       // [16 waves][32 rows][33]
    float* const Cs = smem_all;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Cs[(wave * BM + i * 16 + lq * 4 + r) * CS + j * 16 + li] = acc[i][j][r];
    __syncthreads();
    {
        const int row = tid >> 5, col = tid & 31;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) v += Cs[(w * BM + row) * CS + col];
        const int n = n0 + col;
        if (my_ok) {
            const int64_t m = my_m;
            v += my_bias;
            if (p.alpha != 1.0f) v = v > 0.f ? v : p.alpha * v;
            float* dst = p.out + m * p.out_ld + n;
            v += my_old;                                  // (0 unless accumulate)
            if (p.mask_ref) v *= (my_mk > 0.f || n < p.mask_c0 || n >= p.mask_c1) ? 1.0f : p.mask_alpha;
            *dst = v;
            if (p.shadow) p.shadow[m * p.shadow_ld + n] = (unsigned short)mh_pack_bf16(v, 0.f);
        }
    }
}

// fragment bank writer: one thread per (chunk, 16-column tile, lane): 8 k-values of one output column -> 16 bytes per plane
__global__ __launch_bounds__(256) void pack_weights_kernel(const mh_pack_seg* __restrict__ segs, int nseg) {
    const int lo = mh_find_seg(segs, nseg, (int)blockIdx.x);
    const mh_pack_seg sg = segs[lo];
    const int e = ((int)blockIdx.x - sg.blk0) * 256 + (int)threadIdx.x;
    const int lane = e & 63, qt = e >> 6;
    if (sg.trans == 2 || sg.trans == 3) {
        // the 32x32x16 register image of conv_planes_kernel: bank[chunk][tap][s][32-column tile][plane][lane][8 bf16], lane l holding the 8 reduction
        // channels 16 (chunk * kc16 + s) + 8 (l >> 5) .. + 7 of column 32 tile + (l & 31).  kc16 = 0: one chunk = the whole reduction.
        //   trans 2 (forward): K = Cin, N = Cout, src[tap][K][N]; planes = 2 (hi + lo, split-bf16) or 1 (plain bf16)
        //   trans 3 (input gradient, mh_conv2d_planes_bwd): ONE plane, reduction over Cout (sg.K), columns = Cin (sg.N): the same HWIO bank read
        //   transposed (src[tap][N][K]) with the taps MIRRORED (tap t of the walk = tap taps-1-t of the forward layer) -- the 'SAME' 3x3 input
        //   gradient then IS the forward walk over dz
        const int k16 = (sg.K + 15) >> 4, nt32 = (sg.N + 31) >> 5;
        const int kc = sg.kc16 > 0 ? sg.kc16 : k16, nch = (k16 + kc - 1) / kc;
        if (qt >= nch * sg.taps * kc * nt32) return;
        const int nt = qt % nt32, q = qt / nt32;
        const int st = q % kc, q2 = q / kc;
        const int tap = q2 % sg.taps, ch = q2 / sg.taps;
        const int k0 = (ch * kc + st) * 16 + (lane >> 5) * 8, n = nt * 32 + (lane & 31);
        const int ts = sg.trans == 3 ? sg.taps - 1 - tap : tap;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            v[j] = (k0 + j < sg.K && n < sg.N) ? (sg.trans == 3 ? sg.src[((int64_t)ts * sg.N + n) * sg.K + k0 + j] : sg.src[((int64_t)ts * sg.K + k0 + j) * sg.N + n]) : 0.f;
        unsigned hh[4], ll[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) mh_split_bf16x2(v[2 * j], v[2 * j + 1], hh[j], ll[j]);
        const int pln = (sg.trans == 3 || sg.planes < 2) ? 1 : 2;
        u32x4* dst = reinterpret_cast<u32x4*>(sg.dst) + (int64_t)qt * (64 * pln) + lane;
        dst[0] = (u32x4){hh[0], hh[1], hh[2], hh[3]};
        if (pln == 2) dst[64] = (u32x4){ll[0], ll[1], ll[2], ll[3]};
        return;
    }
    const int cpt = (sg.K + 31) >> 5, np16 = (sg.N + 15) >> 4;
    if (qt >= sg.taps * cpt * np16) return;
    const int nt = qt % np16, q = qt / np16;
    const int c = q % cpt, tap = q / cpt;
    const int k0 = c * 32 + (lane >> 4) * 8, n = nt * 16 + (lane & 15);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
        v[j] = (k0 + j < sg.K && n < sg.N) ? (sg.trans ? sg.src[((int64_t)tap * sg.N + n) * sg.K + k0 + j] : sg.src[((int64_t)tap * sg.K + k0 + j) * sg.N + n]) : 0.f;
    unsigned hh[4], ll[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) mh_split_bf16x2(v[2 * j], v[2 * j + 1], hh[j], ll[j]);
    const u32x4 h = {hh[0], hh[1], hh[2], hh[3]}, l = {ll[0], ll[1], ll[2], ll[3]};
    u32x4* dst = reinterpret_cast<u32x4*>(sg.dst) + ((int64_t)qt * sg.planes) * 64 + lane;
    dst[0] = h;
    if (sg.planes > 1) dst[64] = l;
}

constexpr size_t PATCH_LDS_MAX = 150 * 1024;

size_t patch_lds(int TH, int BM, int BN, int KP, bool x3 = false) {
    const size_t tiles = ((size_t)(TH + 2) * PW * (KP + 16) * 2 + (size_t)2 * BN * LSB * 2) * (x3 ? 2 : 1);
    const size_t cs = (size_t)BM * (BN + 4) * 4;
    return tiles > cs ? tiles : cs;
}

size_t bank_lds(int TH, int BM, int BN, int KP, bool x3) {
    const size_t patch = (size_t)(TH + 2) * PW * (KP + 16) * 2 * (x3 ? 2 : 1);
    const size_t cs = (size_t)BM * (BN + 4) * 4;
    return patch > cs ? patch : cs;
}

// mode: 0 = off, 1 = heuristic tile, 64 / 128 = forced pixel tile; bit 8: the 8-wave variant of the 128-pixel tile
constexpr int PATCH_DEFAULT = 1;
std::atomic<int> g_patch_mode{-2};   // -2: not resolved yet (MH_CONV_PATCH in the environment overrides the default); process-wide tuning hook
int patch_mode() {
    int m = g_patch_mode.load(std::memory_order_relaxed);
    if (m == -2) { const char* e = getenv("MH_CONV_PATCH"); m = e ? atoi(e) : PATCH_DEFAULT; g_patch_mode.store(m, std::memory_order_relaxed); }
    return m;
}
std::atomic<int> g_patch_launches{0};     // since the last mh_tune_conv_patch() call (tests check that the kernel under test really ran)

template <int WM, int WN, int MT, int NT, bool DGRAD, int CPTC = 0, bool X3 = false>
int launch_patch(ConvArgs& a, hipStream_t s) {
    constexpr int BM = WM * MT * 16, BN = WN * NT * 16, TH = BM / 16;
    static std::atomic<uint64_t> attr_done{0};
    const uint64_t attr_dev = mh_device_bit();
    if (!(attr_done.load(std::memory_order_relaxed) & attr_dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_patch_kernel<WM, WN, MT, NT, DGRAD, CPTC, X3>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)PATCH_LDS_MAX);
        if (e != hipSuccess) { mh_set_error("conv_patch: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_done.fetch_or(attr_dev);
    }
    if (a.M < 0) return 0;
    PatchGeo g;
    g.TH = TH;
    const int d = a.dil;
    g.tiles_y = mh_cdiv(mh_cdiv(a.Ho, d), TH);
    g.tiles_x = mh_cdiv(mh_cdiv(a.Wo, d), 16);
    g.ntiles_n = mh_cdiv(a.N, BN);
    g.nwg = a.B * d * d * g.tiles_y * g.tiles_x * g.ntiles_n;
    {
        const int64_t nwg64 = (int64_t)g.nwg;
        const int dmax = std::max(std::max(g.ntiles_n, g.tiles_x), std::max(g.tiles_y, (int)d));
        MH_REQUIRE(mh_fastdiv_ok(nwg64, dmax), MH_ERR_UNSUPPORTED, "tile decode: %lld workgroups x divisor %d exceeds the 2^32 range of the magic-multiplier division", (long long)nwg64, dmax);
    }
    g.dec = mh_make_tile_decode(g.ntiles_n, g.tiles_x, g.tiles_y, d);
    g.KP = (a.K + 31) & ~31;
    g.CPT = g.KP / 32;
    g.PS = g.KP + 16;
    g.nchunk = 9 * g.CPT;
    g.patch_halfs = (TH + 2) * PW * g.PS;
    g.inv_kp4 = 1.0f / (float)(g.KP / 4);
    g.mul_kp8 = (unsigned)(((1ull << 32) + (g.KP / 8) - 1) / (g.KP / 8));
    g.dbg = (patch_mode() >> 9) & 3;
    const size_t lds = patch_lds(TH, BM, BN, g.KP, X3);
    ++g_patch_launches;
    mh_note_kernel("conv_patch_kernel<%d,%d,%d,%d,%s,CPTC=%d,%s> tile %dx%d K=%d dil=%d grid %d lds %d", WM, WN, MT, NT, DGRAD ? "dgrad" : "fwd", CPTC,
                   X3 ? "bf16x3" : "bf16", BM, BN, a.K, a.dil, g.nwg, (int)lds);
    hipLaunchKernelGGL((conv_patch_kernel<WM, WN, MT, NT, DGRAD, CPTC, X3>), dim3(g.nwg), dim3(WM * WN * 64), lds, s, a, g);
    return mh_check_launch("conv_patch");
}

std::atomic<int> g_bank_launches{0};
template <int WM, int WN, int MT, int NT, bool X3>
int launch_bank(ConvArgs& a, hipStream_t s) {
    constexpr int BM = WM * MT * 16, BN = WN * NT * 16, TH = BM / 16;
    static std::atomic<uint64_t> attr_done{0};
    const uint64_t attr_dev = mh_device_bit();
    if (!(attr_done.load(std::memory_order_relaxed) & attr_dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bank_kernel<WM, WN, MT, NT, X3>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)PATCH_LDS_MAX);
        if (e != hipSuccess) { mh_set_error("conv_bank: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_done.fetch_or(attr_dev);
    }
    if (a.M < 0) return 0;
    PatchGeo g;
    g.TH = TH;
    const int d = a.dil;
    g.tiles_y = mh_cdiv(mh_cdiv(a.Ho, d), TH);
    g.tiles_x = mh_cdiv(mh_cdiv(a.Wo, d), 16);
    g.ntiles_n = mh_cdiv(a.N, BN);
    g.nwg = a.B * d * d * g.tiles_y * g.tiles_x * g.ntiles_n;
    {
        const int64_t nwg64 = (int64_t)g.nwg;
        const int dmax = std::max(std::max(g.ntiles_n, g.tiles_x), std::max(g.tiles_y, (int)d));
        MH_REQUIRE(mh_fastdiv_ok(nwg64, dmax), MH_ERR_UNSUPPORTED, "tile decode: %lld workgroups x divisor %d exceeds the 2^32 range of the magic-multiplier division", (long long)nwg64, dmax);
    }
    g.dec = mh_make_tile_decode(g.ntiles_n, g.tiles_x, g.tiles_y, d);
    g.KP = (a.K + 31) & ~31;
    g.CPT = g.KP / 32;
    g.PS = g.KP + 16;
    g.nchunk = 9 * g.CPT;
    g.patch_halfs = (TH + 2) * PW * g.PS;
    g.inv_kp4 = 1.0f / (float)(g.KP / 4);
    g.mul_kp8 = (unsigned)(((1ull << 32) + (g.KP / 8) - 1) / (g.KP / 8));
    g.dbg = (patch_mode() >> 9) & 3;
    const size_t lds = bank_lds(TH, BM, BN, g.KP, X3);
    ++g_patch_launches;
    ++g_bank_launches;
    mh_note_kernel("conv_bank_kernel<%d,%d,%d,%d,%s> tile %dx%d K=%d dil=%d grid %d lds %d", WM, WN, MT, NT, X3 ? "bf16x3" : "bf16", BM, BN, a.K, a.dil,
                   g.nwg, (int)lds);
    hipLaunchKernelGGL((conv_bank_kernel<WM, WN, MT, NT, X3>), dim3(g.nwg), dim3(WM * WN * 64), lds, s, a, g);
    return mh_check_launch("conv_bank");
}

std::atomic<int> g_bank_small_place{-1};          // mh_tune_conv_bank_small: workgroup placement of the small-layer kernel (-1 = the model)
template <bool DGRAD, int PL>
int launch_bank_small(ConvArgs& a, hipStream_t s) {
    static std::atomic<uint64_t> attr_done{0};
    const uint64_t attr_dev = mh_device_bit();
    if (!(attr_done.load(std::memory_order_relaxed) & attr_dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bank_small_kernel<DGRAD, PL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)PATCH_LDS_MAX);
        if (e != hipSuccess) { mh_set_error("conv_bank_small: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_done.fetch_or(attr_dev);
    }
    if (a.M < 0) return 0;
    PatchGeo g;
    g.TH = 2;
    const int d = a.dil;
    g.tiles_y = mh_cdiv(mh_cdiv(a.Ho, d), 2);
    g.tiles_x = mh_cdiv(mh_cdiv(a.Wo, d), 16);
    g.ntiles_n = mh_cdiv(a.N, 32);
    g.nwg = a.B * d * d * g.tiles_y * g.tiles_x * g.ntiles_n;
    {
        const int64_t nwg64 = (int64_t)g.nwg;
        const int dmax = std::max(std::max(g.ntiles_n, g.tiles_x), std::max(g.tiles_y, (int)d));
        MH_REQUIRE(mh_fastdiv_ok(nwg64, dmax), MH_ERR_UNSUPPORTED, "tile decode: %lld workgroups x divisor %d exceeds the 2^32 range of the magic-multiplier division", (long long)nwg64, dmax);
    }
    g.KP = (a.K + 31) & ~31;
    // ---- placement (round 6): which L2s see the layer.  Bytes an XCD's chunk pulls through its L2 under the two logical orders, for X = xuse XCDs:
    //   pixel-major (column tile fastest): every XCD reads the whole bank, the input once over all XCDs  ->  X * bank + in
    //   column-major (n_major)           : an XCD reads ntn / X (>= 1) column slices and every pixel of them -> max(X, ntn) / ntn * bank + min(X, ntn) * in
    // and X itself: the fewest XCDs that still give every workgroup a CU of its own (32 CUs per XCD), so that a 24-workgroup layer is ONE L2's business.
    {
        // 0 = round-5 order (all XCDs, pixel-major), 1 = the order model on all XCDs (DEFAULT: time-neutral, less L2 fill traffic), 2 = confinement only, 3 = both.
        // Measured (profiles/r06_microbench_small_placement.txt): 5.1 - 5.8 us per node in EVERY mode; confinement costs the step +2 us (1.3330 vs 1.3309 ms).
        const int mode_raw = g_bank_small_place.load(std::memory_order_relaxed);
        const int mode = mode_raw < 0 ? 1 : mode_raw;
        const int npt = g.nwg / g.ntiles_n;
        int X = 8;
        if (mode >= 2) { X = 1; while (X < 8 && mh_cdiv(g.nwg, X) > 32) X *= 2; }
        const double bank = 9.0 * g.KP * a.N * 2.0 * PL, in = (double)a.B * (DGRAD ? a.Ho : a.Hi) * (DGRAD ? a.Wo : a.Wi) * a.K * 4.0;
        const double pix_major = X * bank + in;
        const double col_major = (double)std::max(X, g.ntiles_n) / g.ntiles_n * bank + std::min(X, g.ntiles_n) * in;
        g.n_major = (mode != 0 && mode != 2 && col_major < pix_major) ? 1 : 0;
        g.xuse = X;
        g.per = mh_cdiv(g.nwg, X);
        g.f_pt = mh_make_fastdiv(npt);
        MH_REQUIRE(mh_fastdiv_ok((int64_t)g.nwg, npt), MH_ERR_UNSUPPORTED, "tile decode: %d workgroups x %d pixel tiles exceeds the magic-multiplier range", g.nwg, npt);
    }
    g.dec = mh_make_tile_decode(g.n_major ? 1 : g.ntiles_n, g.tiles_x, g.tiles_y, d);
    g.CPT = g.KP / 32;
    g.PS = g.KP + 16;
    g.nchunk = 9 * g.CPT;
    const int st = DGRAD ? 1 : a.stride;
    g.patch_halfs = (st * 2 + 3 - st) * (st * 16 + 3 - st) * g.PS;
    g.f_kp4 = mh_make_fastdiv(g.KP / 4); g.f_pc = mh_make_fastdiv(st * 16 + 3 - st);
    g.inv_kp4 = 1.0f / (float)(g.KP / 4);
    g.mul_kp8 = (unsigned)(((1ull << 32) + (g.KP / 8) - 1) / (g.KP / 8));
    g.dbg = 0;
    const size_t patch = (size_t)g.patch_halfs * 2 * PL, cs = (size_t)16 * 32 * 33 * 4;
    const size_t lds = patch > cs ? patch : cs;
    ++g_bank_launches;
    const int nblocks = g.xuse < 8 ? 8 * g.per : g.nwg;
    mh_note_kernel("conv_bank_small_kernel<%s,%s> tile 32x32 K=%d N=%d dil=%d grid %d lds %d", DGRAD ? "dgrad" : "fwd", PL == 2 ? "bf16x3" : "bf16", a.K, a.N, a.dil,
                   g.nwg, (int)lds);
    hipLaunchKernelGGL((conv_bank_small_kernel<DGRAD, PL>), dim3(nblocks), dim3(1024), lds, s, a, g);
    return mh_check_launch("conv_bank_small");
}

// the channel-count-specialised instances exist for the 8-wave tile (the one the heuristic dispatches)
template <int WM, int WN, int MT, int NT, bool DGRAD>
int launch_patch_k(ConvArgs& a, hipStream_t s) {
    const bool all = a.M < 0;
    const bool generic = !all && ((patch_mode() >> 11) & 1) != 0;     // tuning hook (mode bit 11): never take the compile-time-K instances
    int rc = 0;
    if constexpr (WM == 4) {
        if (all || (a.K == 128 && !generic)) { rc = launch_patch<WM, WN, MT, NT, DGRAD, 4>(a, s); if (!all || rc) return rc; }
        if (all || (a.K == 64 && !generic)) { rc = launch_patch<WM, WN, MT, NT, DGRAD, 2>(a, s); if (!all || rc) return rc; }
    }
    return launch_patch<WM, WN, MT, NT, DGRAD>(a, s);
}

template <int WM, int WN, int MT, bool DGRAD>
int launch_patch_n(ConvArgs& a, hipStream_t s, int bn) {
    const bool all = a.M < 0;
    int rc = 0;
    if (all || bn == 128) { rc = launch_patch_k<WM, WN, MT, 4, DGRAD>(a, s); if (!all || rc) return rc; }
    if (all || bn == 96) { rc = launch_patch_k<WM, WN, MT, 3, DGRAD>(a, s); if (!all || rc) return rc; }
    if (all || bn == 64) { rc = launch_patch_k<WM, WN, MT, 2, DGRAD>(a, s); if (!all || rc) return rc; }
    return rc;
}

// (round 3: a 32-column tile for the layers with <= 32 output columns -- 32 -> 32, 64 -> 32 and their input gradients: with the 64-column tile half of
//  every MFMA and of the epilogue was padding: 15.0 -> 13.6 us forward, 16.5 -> 14.9 us input gradient for the pyramid's 32 -> 32 layer at 96x320 x 2.
//  A 16-column tile for the 16 -> 16 layer at 192x640 was measured too and is SLOWER than the tiled kernel (32.3 vs 28.7 us forward, 41.4 vs 28.4 us input
//  gradient): per-workgroup latency, not padding, bounds that layer -- removed.)
int patch_bn(const ConvArgs& a) {
    if (a.N <= 32) return 32;
    return a.x3 ? (a.N > 64 ? 128 : 64) : (a.N > 96 ? 128 : (a.N > 64 ? 96 : 64));
}

// Tile choice of the heuristic mode (measured on MI355X, profiles/r01_microbench_conv_patch.txt): the 128-pixel tile with 8
// waves for the forward pass and the input gradient alike (its compile-time-K instances take K = 64 / 128).
int patch_bm(const ConvArgs& a) {
    if (a.x3) return a.N > 64 ? 64 : 128;      // split-bf16: 64 pixels x 128 columns or 128 pixels x 64 columns (two planes of everything in LDS)
    if ((patch_mode() & 0xff) == 64 || (patch_mode() & 0xff) == 128) return patch_mode() & 0xff;
    return 128;
}
bool patch_w8(const ConvArgs& a) { return (patch_mode() & 0xff) == 1 ? true : (patch_mode() & 0x100) != 0; }

}  // namespace

extern "C" int mh_tune_conv_patch(int mode) {
    g_patch_mode = mode < 0 ? PATCH_DEFAULT : mode;
    return g_patch_launches.exchange(0);
}

extern "C" int64_t mh_pack_bytes(int32_t taps, int32_t K, int32_t N, int32_t planes) {
    return (int64_t)taps * ((K + 31) / 32) * ((N + 15) / 16) * planes * 1024;
}
extern "C" int mh_pack_weights(const mh_pack_seg* segs_device, int32_t nseg, int32_t nblocks, void* stream) {
    MH_REQUIRE(segs_device && nseg > 0 && nblocks > 0, MH_ERR_ARG, "mh_pack_weights: empty segment table");
    hipLaunchKernelGGL(pack_weights_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, segs_device, nseg);
    return mh_check_launch("pack_weights");
}
std::atomic<int> g_bank_small_maxpix{-2};          // -2: the default (4096; the engines' bank_small_maxpix must agree: they pack the banks for it)
int bank_small_maxpix() {
    const int m = g_bank_small_maxpix.load(std::memory_order_relaxed);
    return m == -2 ? 4096 : m;
}
std::atomic<int> g_bank_small_tile_wgs{-1};       // -1: the default (200); 0 = never
int bank_small_tile_wgs() {
    const int m = g_bank_small_tile_wgs.load(std::memory_order_relaxed);
    return m < 0 ? 200 : m;
}
extern "C" int mh_tune_conv_bank_small(int mode) { return g_bank_small_place.exchange(mode < 0 ? -1 : mode); }
extern "C" int mh_tune_conv_bank_tile(int max_wgs) { return g_bank_small_tile_wgs.exchange(max_wgs < 0 ? -1 : max_wgs); }
extern "C" int mh_tune_conv_bank(int small_maxpix) {
    g_bank_small_maxpix = small_maxpix < 0 ? -2 : small_maxpix;
    return g_bank_launches.exchange(0);
}

// small-layer bank kernel: stride-1 "SAME" 3x3, bank present, reduction <= 64 chunks (K <= 224), few enough pixels that the tiled kernels
// are latency bound (default <= 4096 output pixels: the 1/16-1/64 levels; mh_tune_conv_bank overrides, 0 = off)
bool mh_conv_bank_small_ok(const ConvArgs& a) {
    // input gradients: twice the forward limit (the 1/8-resolution level too: 1.804-1.809 -> 1.800-1.802 ms per step against the tiled kernel
    // there, profiles/r03_experiments.txt #11; in the forward pass that level runs split-bf16 on the big bank kernel)
    const int maxpix = a.mode == 1 ? 2 * bank_small_maxpix() : bank_small_maxpix();
    if (!a.wb || !(a.bf16 || a.x3) || (a.x3 && a.mode != 0)) return false;
    const bool s1 = a.stride == 1 && a.pad_t == a.dil && a.pad_l == a.dil && a.Hi == a.Ho && a.Wi == a.Wo;
    const bool s2 = a.stride == 2 && a.mode == 0 && a.dil == 1 && a.pad_t >= 0 && a.pad_t <= 1 && a.pad_l >= 0 && a.pad_l <= 1 &&
                    a.Ho == (a.Hi + 1) / 2 && a.Wo == (a.Wi + 1) / 2;
    if (!(a.kh == 3 && a.kw == 3 && (s1 || s2))) return false;
    if (a.ncls != 0 || a.N < 16 || a.K < 16 || a.dil > 64 || !a.vecA) return false;
    if (9 * ((a.K + 31) / 32) > 64) return false;
    if (s2 && (size_t)5 * 33 * (((a.K + 31) & ~31) + 16) * 2 * (a.x3 ? 2 : 1) > PATCH_LDS_MAX) return false;
    if ((int64_t)a.B * a.Ho * a.Wo > maxpix) return false;
    if ((s2 ? 5 * 33 : 4 * 18) * (((a.K + 31) & ~31) / 4) > 2 * 4 * 1024) return false;      // the patch staging is two straight-line rounds of 4 loads per thread
    const int d = a.dil;
    const int64_t cover = (int64_t)d * d * mh_cdiv(mh_cdiv(a.Ho, d), 2) * 2 * mh_cdiv(mh_cdiv(a.Wo, d), 16) * 16;
    return cover * 100 <= (int64_t)a.Ho * a.Wo * 250 && (int64_t)a.B * cover / 32 * mh_cdiv(a.N, 32) < (1 << 20);
}
int mh_conv_bank_small_launch(ConvArgs& a, hipStream_t s) {
    const bool all = a.M < 0;
    int rc = 0;
    if (all || (a.mode == 0 && a.x3)) { rc = launch_bank_small<false, 2>(a, s); if (!all || rc) return rc; }
    if (all || (a.mode == 0 && !a.x3)) { rc = launch_bank_small<false, 1>(a, s); if (!all || rc) return rc; }
    if (all || a.mode == 1) { rc = launch_bank_small<true, 1>(a, s); if (!all || rc) return rc; }
    return rc;
}

bool mh_conv_patch_ok(const ConvArgs& a) {
    if (patch_mode() == 0) return false;
    if (!((a.bf16 || (a.x3 && a.mode == 0)) && a.vecA && a.vecB && (a.vecC || (a.vecCpad && a.mode == 1 && a.bf16)))) return false;
    if (!(a.kh == 3 && a.kw == 3 && a.stride == 1 && a.pad_t == a.dil && a.pad_l == a.dil && a.Hi == a.Ho && a.Wi == a.Wo)) return false;
    constexpr int min_n = 32;                                                                   // the 32-column tile: from 32 output columns
    if (a.ncls != 0 || a.N < min_n || a.K < 32 || a.dil > 64) return false;
    if (a.x3 && a.mode == 0 && !a.wb && (a.N < 48 || a.K < 32)) return false;                      // the LDS-staged split-bf16 instances keep their floor
    if (a.mode == 1 && (patch_mode() & 0x1000)) return false;                                  // mode bit 12: forward layers only
    if (a.mode == 1 && (patch_mode() & 0x2000) && a.K != 64 && a.K != 128) return false;       // mode bit 13: no generic-K input gradients
    const int bm = patch_bm(a), bn = patch_bn(a);
    if (patch_lds(bm / 16, bm, bn, (a.K + 31) & ~31, a.x3 != 0) > PATCH_LDS_MAX) return false;
    // lattice fill: the share of tile pixels that are real output pixels (small images under a large dilation waste tiles)
    const int d = a.dil, TH = bm / 16;
    const int64_t cover = (int64_t)d * d * mh_cdiv(mh_cdiv(a.Ho, d), TH) * TH * mh_cdiv(mh_cdiv(a.Wo, d), 16) * 16;
    if ((patch_mode() & 0xff) == 1) {
        constexpr int min_pix = 24576, max_cover = 220;          // thresholds of the heuristic
        // lattice tiles may cover up to 2.2x the image: round 1 set 125 % (bf16 kernel vs the bf16 gather kernel, stand-alone); in the round-2 step the
        // context layers of dilation 8 / 16 (213 % at 96x320) are faster on the patch / bank kernels than on the tiled ones, forward (where the
        // alternative is exact fp32: 40 us) and input gradient alike: 1.986 -> 1.956 ms, bf16 mode 1.887 -> 1.869, MAD 1.104 -> 1.087 (r03y3)
        if (cover * 100 > (int64_t)a.Ho * a.Wo * max_cover) return false;
        // default: fewer than ~200 128-pixel tiles = not one workgroup per CU.  Split-bf16 competes with the exact-fp32 gather
        // kernel instead of the bf16 one, which moves the break-even down to the 1/8-resolution level (7680 pixels: 20 us vs
        // 24-29 us; at 1920 pixels fp32 wins, 14 vs 19.5 us -- profiles/r02_microbench_x3dbg.txt)
        // (measured in situ: 128-column layers 24-28 -> 19-20 us, the 64-column one 17.6 -> 20.8 us: wide layers only)
        // with a fragment bank the split-bf16 kernel also beats the exact-fp32 gather kernel on the <= 64-column layers at 1/8 resolution
        constexpr int bank_min_pix = 7680;
        const int need = (a.x3 && a.wb && a.mode == 0) ? (bank_min_pix < min_pix ? bank_min_pix : min_pix) : ((a.x3 && a.N > 64) ? min_pix * 5 / 16 : min_pix);
        if ((int64_t)a.B * a.Ho * a.Wo < need) return false;
    }
    return (int64_t)a.B * d * d * mh_cdiv(mh_cdiv(a.Ho, d), TH) * mh_cdiv(mh_cdiv(a.Wo, d), 16) < (1 << 30);
}

int mh_conv_patch_launch(ConvArgs& a, hipStream_t s) {
    const bool all = a.M < 0;
    const bool dg = a.mode == 1;
    const int bm = all ? 0 : patch_bm(a), bn = all ? 0 : patch_bn(a);
    const bool w8 = all ? false : patch_w8(a);
    int rc = 0;
    // fragment-bank instances (mh_conv2d_wb: the layer's weights packed by mh_pack_weights; split-bf16 forward)
    {
        const bool wb = !all && a.wb && a.x3 && a.mode == 0 && !(patch_mode() & 0x4000);       // mode bit 14: ignore the bank
        const bool big = !all && (patch_mode() & 0x8000) != 0;                                  // mode bit 15: 128-pixel tile (64x32 wave tiles)
        if (all || (wb && bn == 128 && big)) { rc = launch_bank<2, 4, 4, 2, true>(a, s); if (!all || rc) return rc; }
        const bool w4 = !all && (patch_mode() & 0x10000) != 0;                                 // mode bit 16: 64-pixel tile with 4 waves of 64x32 (half the fragment bytes per MFMA)
        if (all || (wb && bn == 128 && w4)) { rc = launch_bank<1, 4, 4, 2, true>(a, s); if (!all || rc) return rc; }
        // under-filled grids (the 1/8-resolution level: 120 tiles of 64x128, 60 of 128x64 on 256 CUs): a 64-pixel x 64-column tile with 4 waves
        // doubles / quadruples the workgroups (the patch is staged once per column tile more; profiles/r03_experiments.txt #16)
        const int small_wg = bank_small_tile_wgs();
        bool few = false;
        if (wb && (bn == 128 || bn == 64)) {
            const int d = a.dil, th = bn == 128 ? 4 : 8;
            few = (int64_t)a.B * d * d * mh_cdiv(mh_cdiv(a.Ho, d), th) * mh_cdiv(mh_cdiv(a.Wo, d), 16) < small_wg;
        }
        if (all || (wb && few)) { rc = launch_bank<2, 2, 2, 2, true>(a, s); if (!all || rc) return rc; }
        if (all || (wb && bn == 128)) { rc = launch_bank<2, 4, 2, 2, true>(a, s); if (!all || rc) return rc; }
        if (all || (wb && bn == 64)) { rc = launch_bank<4, 2, 2, 2, true>(a, s); if (!all || rc) return rc; }
        if (all || (wb && bn == 32)) { rc = launch_bank<8, 1, 1, 2, true>(a, s); if (!all || rc) return rc; }
    }
    // split-bf16 forward instances (precision code 2)
    {
        const bool generic = !all && ((patch_mode() >> 11) & 1) != 0;      // tuning hook (mode bit 11): generic-K instances only
        if (all || (a.x3 && bn == 128 && a.K == 128 && !generic)) { rc = launch_patch<2, 4, 2, 2, false, 4, true>(a, s); if (!all || rc) return rc; }
        if (all || (a.x3 && bn == 128 && a.K == 64 && !generic)) { rc = launch_patch<2, 4, 2, 2, false, 2, true>(a, s); if (!all || rc) return rc; }
        if (all || (a.x3 && bn == 128)) { rc = launch_patch<2, 4, 2, 2, false, 0, true>(a, s); if (!all || rc) return rc; }
        if (all || (a.x3 && bn == 64 && a.K == 128 && !generic)) { rc = launch_patch<4, 2, 2, 2, false, 4, true>(a, s); if (!all || rc) return rc; }
        if (all || (a.x3 && bn == 64 && a.K == 64 && !generic)) { rc = launch_patch<4, 2, 2, 2, false, 2, true>(a, s); if (!all || rc) return rc; }
        if (all || (a.x3 && bn == 64)) { rc = launch_patch<4, 2, 2, 2, false, 0, true>(a, s); if (!all || rc) return rc; }
    }
    if (all || (!a.x3 && bn == 32 && !dg)) { rc = launch_patch<4, 2, 2, 1, false>(a, s); if (!all || rc) return rc; }
    if (all || (!a.x3 && bn == 32 && dg)) { rc = launch_patch<4, 2, 2, 1, true>(a, s); if (!all || rc) return rc; }
    if (all || (!dg && bm == 128 && !w8)) { rc = launch_patch_n<2, 2, 4, false>(a, s, bn); if (!all || rc) return rc; }
    if (all || (dg && bm == 128 && !w8)) { rc = launch_patch_n<2, 2, 4, true>(a, s, bn); if (!all || rc) return rc; }
    if (all || (!dg && bm == 128 && w8)) { rc = launch_patch_n<4, 2, 2, false>(a, s, bn); if (!all || rc) return rc; }
    if (all || (dg && bm == 128 && w8)) { rc = launch_patch_n<4, 2, 2, true>(a, s, bn); if (!all || rc) return rc; }
    if (all || (!dg && bm == 64)) { rc = launch_patch_n<2, 2, 2, false>(a, s, bn); if (!all || rc) return rc; }
    if (all || (dg && bm == 64)) { rc = launch_patch_n<2, 2, 2, true>(a, s, bn); if (!all || rc) return rc; }
    if (all) return 0;
    mh_set_error("conv_patch: no variant for bm=%d bn=%d", bm, bn);
    return MH_ERR_UNSUPPORTED;
}
