// conv.hip -- implicit-GEMM convolution on the gfx950 matrix cores for the MADNet/DispNet conv stacks, in two arithmetic
// modes selected per launch (mh_conv_desc.precision): exact fp32 (v_mfma_f32_16x16x4_f32, bitwise an fmaf chain -- the
// parity path) and bf16 inputs with fp32 accumulation (v_mfma_f32_16x16x32_bf16 -- the throughput path); tensors are
// fp32 in HBM either way.
//
// Replaces tf.nn.conv2d / tf.nn.atrous_conv2d / tf.nn.conv2d_transpose + bias_add + leaky
// (Nets/sharedLayers.py:54-92) and, with mode=1 / w_trans=1, their input gradients.
//
// GEMM view:  C[m][n] = sum_{tap,k} A[m][(tap,k)] * B[(tap,k)][n]
//   m = output pixel (b,oy,ox) -- NHWC so C row m lives at out + m*out_ld
//   A[m][(tap,k)] = in[b, iy(oy,ky), ix(ox,kx), k]  (0 outside the image)
//   B[(tap,k)][n] = w[tap][k][n] (forward, HWIO as stored) or w[tap][n][k] (dgrad)
// K is walked in groups of 4 channels flattened across taps (group g -> tap=g/G, c4=g%G,
// G=ceil(K/4)), so odd channel counts (3, 38, 33, 197 ...) cost no padded MFMA work beyond
// the last group.  One workgroup = 4 waves computes a BM x BN tile; a K-tile = 32 / 64 / 128
// k-values; global->register prefetch of tile t+1 (t+2 for the small tiles) overlaps the MFMAs of
// tile t; LDS is double buffered with ONE barrier per K-tile.
// Variants in this file: conv_igemm_kernel (the tiled kernel, incl. intra-workgroup split-K groups for the small tiles and
// the 4 parity classes of a stride-2 input gradient / transposed conv in one launch), conv_n1_fwd_kernel (single output
// channel), conv_thin_kernel (weights-stationary direct conv of the image layer).
#include "mh_common.h"
#include "conv_args.h"
#include <stdlib.h>
#include <atomic>

namespace {

// LDS tiles are k-contiguous for BOTH operands (As[row][k], Bs[col][k], row stride KT+4 floats) so
// every lane fetches 4 consecutive k of its row/column with ONE ds_read_b128 and feeds 4 MFMAs:
// MFMA step (s,t) contracts k = s*16 + (lane>>4)*4 + t -- any k permutation is legal as long as A
// and B agree.  KT (k-values per K-tile) is 32 for the big tiles and 64 for the small, latency-bound
// ones: fewer barriers, 2-4x more bytes in flight per barrier.
// The B tile is written TRANSPOSED in the forward case (global float4 runs along n, LDS rows along k):
// consecutive lanes then hit rows 4*LS floats apart = the same 2 banks.  XOR-ing the 4-float group
// index with row bits (bijective per row, keeps each float4 intact) spreads those stores over 16
// banks; the same involution is applied on the ds_read_b128 side.
template <int GPT>
__device__ __forceinline__ int swz_group(int row, int g) { return g ^ ((row >> 2) & (GPT - 1)); }

// DGRAD selects the gather geometry + weight orientation at compile time (forward: mode 0 / HWIO
// rows along n; dgrad & conv2d_transpose: mode 1 / rows along k); VEC = 16-byte global loads legal.
// Every global load is UNCONDITIONAL (clamped address + select): a load inside a conditional block
// makes hipcc drain vmcnt(0) before the MFMA block and kills the prefetch overlap.
// UNI: every K-tile lies inside ONE tap (4*G % KT == 0, and stride 1 for DGRAD): the (tap, channel)
// cursor is then wave-uniform (SALU) and a row's byte offset is base(row) + offset(tile) -- the loader
// shrinks from ~20 to ~7 VALU per 16-byte load, which is what bounds the small, 1-wave-per-SIMD tiles.
// BF16 (throughput mode): activations / weights stay fp32 in HBM and are loaded exactly as in the fp32
// kernel, but the LDS tiles hold bf16 (v_cvt_pk_bf16_f32, round-to-nearest-even, at the LDS store) and the
// contraction runs on v_mfma_f32_16x16x32_bf16 with fp32 accumulation (13x the fp32-MFMA rate): the kernel
// becomes loader/L2-bound.  The forward B tile (HWIO rows run along n) is loaded as units of 4 consecutive
// k rows and transposed in registers so that it is stored with 8-byte writes like the other tiles.
// KG > 1 (intra-workgroup split-K, small tiles on a grid smaller than the chip): KG groups of 4 waves each own
// their LDS tiles and walk every KG-th K-tile; the partial accumulator tiles meet in LDS in the epilogue.  A 32x32
// tile at one wave per SIMD is bound by its ~200-instruction K-step (issue latency, not bytes): KG groups
// interleave KG such instruction streams per SIMD.
// X3 (forward, BF16 layout): split-bf16 -- hi and lo planes of both tiles in LDS, three bf16 MFMAs per product (lo*hi + hi*lo + hi*hi): the
// forward layers without a patch / bank instance (strided, thin, wide-K, 5x5 / 7x7: the MADNet pyramid, most of DispNet) stay inside the
// 1e-3 px tolerance at a third of the bf16 MFMA rate instead of on the exact-fp32 MFMA (a sixteenth, and 8x the MFMA instruction count).
// RAG (forward, UNI, bf16): K is NOT a multiple of KT (DispNet's iconv layers read 1024+1, 768+1, 384+1 concat channels): the uniform-tap walk runs over K
// rounded up to whole K-tiles per tap -- weight rows >= K are dropped by one offset compare per load, activation groups >= K load nothing (partial
// groups are zeroed by the store-time select) -- instead of falling back to the per-thread (tap, channel) cursor of the generic loader.
template <int WM, int WN, int MT, int NT, int KT, bool DGRAD, bool VEC, bool UNI, bool BF16, int KG = 1, bool X3 = false, bool RAG = false>
__global__ __launch_bounds__(256 * KG) void conv_igemm_kernel(ConvArgs p) {
    static_assert(!X3 || (BF16 && !DGRAD && VEC), "split-bf16 instances: forward, vector path, bf16 tile layout");
    static_assert(!RAG || (UNI && BF16 && !DGRAD && VEC && !X3), "ragged-K instances: bf16 forward on the uniform-tap loader");
#ifdef MH_PHASE_TIMING
    const unsigned long long ts0 = __builtin_amdgcn_s_memtime(), tr0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0;
#define MH_STAMP(v) v = __builtin_amdgcn_s_memtime()
#else
#define MH_STAMP(v)
#endif
    constexpr int BM = WM * MT * 16;
    constexpr int BN = WN * NT * 16;
    // LDS row stride (elements: floats / bf16).  bf16: 80 halfs = 40 dwords: with the b128 lane groups of gfx950 a 36-dword
    // stride costs a 2-way conflict on every operand read, 40 is conflict free (brute-forced over the lane groups of
    // MI355X_MICROARCH.md; SQ_LDS_BANK_CONFLICT confirmed the model: profiles/r01_pmc_roofline.json)
    constexpr int LS = KT + (BF16 ? 16 : 4);
    constexpr int GPT = KT / 4;                        // 4-channel groups per K-tile
    constexpr int RPP = 256 / GPT;                     // rows loaded per pass
    constexpr int AROWS = (BM + RPP - 1) / RPP;
    constexpr int BVEC = KT * BN / 4;                  // float4 items in a B tile
    constexpr bool BT = BF16 && !DGRAD;                // B tile loaded as 4-row units, transposed in registers
    constexpr int UN = GPT * (BN / 4);                 // such units per tile
    constexpr int BITEMS = BT ? 4 * ((UN + 255) / 256) : (BVEC + 255) / 256;
    constexpr bool PF2 = (BM * BN <= 32 * 64) && (BM <= 64) && VEC;   // small tiles: deep register prefetch (see the K loop); the
                                                                    // 128x16 tile (full-resolution layers) is throughput bound
    // prefetch distance of the small tiles: 4 K-tiles in flight for the 32x32 tile (16 staging VGPRs per stage: the 16-wave
    // split-K variant stays under its 128-VGPR budget), 2 for the wider small tiles
    // (the bf16 forward B tile keeps 4-row units = twice the staging registers: distance 4 would spill there)
    // MEASURED (profiles/r02_experiments.txt): distance 4 made every 32x32-tile layer SLOWER (13.0 -> 16.3 us in situ, fp32 and
    // bf16 alike) -- these launches are not waiting on the load round trip but issuing it (address VALU + zero-padding loads of
    // 16 waves); the instances stay at distance 2.  The distance-4 schedule is kept for reference behind this constant.
    constexpr int PFD = !PF2 ? 1 : 2;
    constexpr int TILE_FLOATS = (BF16 ? (BM + BN) * LS / 2 : (BM + BN) * LS) * (X3 ? 2 : 1);   // one buffer of both tiles (and planes), in floats
    constexpr int LO = 2 * (BM + BN) * LS;             // X3: the lo planes of both buffers sit this many halfs behind the hi planes

    HIP_DYNAMIC_SHARED(float, smem_all)
    const int kg = KG > 1 ? (int)(threadIdx.x >> 8) : 0;          // K group of this wave
    float* const smem = smem_all + kg * (2 * TILE_FLOATS);         // this group's tiles (and, later, its C staging)
    float* const As = smem;                            // fp32: [2][BM*LS] floats
    float* const Bs = smem + 2 * BM * LS;              //       [2][BN*LS]
    unsigned short* const Ah = reinterpret_cast<unsigned short*>(smem);          // bf16: [2][BM*LS] halfs
    unsigned short* const Bh = Ah + 2 * BM * LS;                                //       [2][BN*LS]
    int* const tap_dy = reinterpret_cast<int*>(smem_all + KG * 2 * TILE_FLOATS);
    int* const tap_dx = tap_dy + 64;
    // forward B item j of this thread -> (row kk of the K-tile, 4-column group n4, live)
    auto b_item = [&](int j, int& kk, int& n4) -> bool {
        if (BT) {
            const int u = (threadIdx.x & 255) + 256 * (j >> 2);
            n4 = u % (BN / 4);
            kk = (u / (BN / 4)) * 4 + (j & 3);
            return u < UN;
        }
        const int q = (threadIdx.x & 255) + 256 * j;
        kk = q / (BN / 4); n4 = q % (BN / 4);
        return q < BVEC;
    };

    const int tid = threadIdx.x & 255;                 // index within the K group
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    const int nwg = p.mtiles * p.ntiles;
    const int lin = mh_xcd_remap(blockIdx.x, nwg);
    const int tile_n = lin % p.ntiles;
    const int tile_m = lin / p.ntiles;
    const int n0 = tile_n * BN;
    int* const tap_id = tap_dx + 64;                   // real tap index (weight slice) of the t-th tap this workgroup walks
    int* const row_m = tap_id + 64;                    // [BM] output pixel of tile row r (-1: none), parity-class mode only

    // parity-class mode (DGRAD, stride 2): this workgroup's class, its tap subset and its (qy, qx) pixel grid
    const bool pcm = DGRAD && p.ncls > 0;
    int cls = 0;
    if (pcm) { while (cls + 1 < p.ncls && p.cls[cls + 1].tile0 <= tile_m) ++cls; }
    const int ntaps = pcm ? p.cls[cls].ntaps : p.taps;
    const int Hq = pcm ? p.cls[cls].Hq : p.Ho, Wq = pcm ? p.cls[cls].Wq : p.Wo, Mq = pcm ? p.cls[cls].M : p.M;
    const int m0 = (tile_m - (pcm ? p.cls[cls].tile0 : 0)) * BM;
    const int sshift = pcm ? 0 : p.sshift;             // class coordinates make it a stride-1 gather

    if (tid < ntaps) {
        if (pcm) {
            tap_dy[tid] = -(int)p.cls[cls].dy[tid];    // DGRAD gathers at a_by - tap_dy
            tap_dx[tid] = -(int)p.cls[cls].dx[tid];
            tap_id[tid] = p.cls[cls].id[tid];
        } else {
            const int ky = tid / p.kw, kx = tid - ky * p.kw;
            tap_dy[tid] = ky * p.dil;
            tap_dx[tid] = kx * p.dil;
            tap_id[tid] = tid;
        }
    }
    if (pcm && tid < BM) {
        const int m = m0 + tid;
        int v = -1;
        if (m < Mq) {
            const int qx = m % Wq, t2 = m / Wq;
            v = ((t2 / Hq) * p.Ho + 2 * (t2 % Hq) + p.cls[cls].py) * p.Wo + 2 * qx + p.cls[cls].px;
        }
        row_m[tid] = v;
    }

    // ---- per-thread A-row geometry (constant over the K loop) --------------------------
    const int ga = tid % GPT;          // which channel group of a K-tile
    const int ra = tid / GPT;          // row within a pass
    int a_by[AROWS], a_bx[AROWS];
    int a_img[AROWS];
    bool a_rowok[AROWS];
#pragma unroll
    for (int j = 0; j < AROWS; ++j) {
        const int r = ra + RPP * j;
        const int m = m0 + r;
        const bool ok = (r < BM) && (m < Mq);
        const int mm = ok ? m : 0;
        const int ox = mm % Wq;
        const int t2 = mm / Wq;
        const int oy = t2 % Hq;
        const int b = t2 / Hq;
        a_rowok[j] = ok;
        a_img[j] = b * p.Hi * p.Wi;
        if (!DGRAD) {
            a_by[j] = oy * p.stride - p.pad_t;
            a_bx[j] = ox * p.stride - p.pad_l;
        } else {
            a_by[j] = pcm ? oy : oy + p.pad_t;            // class mode: (qy, qx), the padding is folded into the tap offsets
            a_bx[j] = pcm ? ox : ox + p.pad_l;
        }
    }
    int a_tap = 0, a_c4 = ga + kg * GPT;          // group cursor: g = tile*GPT + ga -> (tap, c4); first tile = kg
    while (a_c4 >= p.G) { a_c4 -= p.G; ++a_tap; }

    // ---- per-thread B-item geometry ------------------------------------------------------
    // w_trans == 1 (k contiguous in memory): item q -> g = q % GPT, n = q / GPT     (float4 along k)
    // w_trans == 0 (n contiguous in memory): item q -> n4 = q % (BN/4), kk = q / (BN/4) (float4 along n)
    int b_tap[BITEMS], b_c4[BITEMS];
#pragma unroll
    for (int j = 0; j < BITEMS; ++j) {
        const int q = tid + 256 * j;
        int kk0, n40;
        b_item(j, kk0, n40);
        const int gb = DGRAD ? (q % GPT) : (kk0 >> 2);
        int t = 0, c = gb + kg * GPT;
        while (c >= p.G) { c -= p.G; ++t; }
        b_tap[j] = t; b_c4[j] = c;
    }

    // UNI fast path: thread-constant byte offsets + wave-uniform tile cursor
    int a_off[AROWS];           // ((img + by*Wi + bx)*in_ld + ga*4)*4
    int b_off[BITEMS];          // fwd: (kk*N + n)*4 ; dgrad: (n*K + g*4)*4   (MH_OOB when the item is dead)
    if (UNI) {
#pragma unroll
        for (int j = 0; j < AROWS; ++j) a_off[j] = ((a_img[j] + a_by[j] * p.Wi + a_bx[j]) * p.in_ld + ga * 4) * 4;
#pragma unroll
        for (int j = 0; j < BITEMS; ++j) {
            const int q = tid + 256 * j;
            int n, o;
            bool live;
            if (!DGRAD) { int kk, n4; live = b_item(j, kk, n4); n = n0 + n4 * 4; o = (kk * p.N + n) * 4; }
            else { live = q < BVEC; n = n0 + q / GPT; o = (n * p.K + (q % GPT) * 4) * 4; }
            b_off[j] = (live && n < p.N) ? o : MH_OOB;
        }
    }
    const int Kc = RAG ? (p.K + KT - 1) / KT * KT : p.K;       // channels one tap occupies in the K walk
    int u_tap = 0, u_c0 = kg * KT;    // wave-uniform cursor of the NEXT tile to load (UNI); Kc % KT == 0
    while (UNI && u_c0 >= Kc) { u_c0 -= Kc; ++u_tap; }

    // Register stages.  Small tiles (PF2) are latency bound -- one exposed load -> LDS -> MFMA round trip per
    // K-tile -- and run with prefetch distance 2: K-tiles t+1 and t+2 are in flight while tile t is multiplied.
    float4 ra0[AROWS], ra1[PF2 ? AROWS : 1], ra2[PFD == 4 ? AROWS : 1], ra3[PFD == 4 ? AROWS : 1];
    float4 rb0[BITEMS], rb1[PF2 ? BITEMS : 1], rb2[PFD == 4 ? BITEMS : 1], rb3[PFD == 4 ? BITEMS : 1];
    int kb0 = 0, kb1 = 0, kb2 = 0, kb3 = 0;     // kbase of the A group held by each stage (padding select at store time)

    __syncthreads();   // tap tables visible
    MH_STAMP(ts1);

    const __amdgpu_buffer_rsrc_t rs_in = mh_make_rsrc(p.in, p.in_bytes);
    const __amdgpu_buffer_rsrc_t rs_w = mh_make_rsrc(p.w, p.w_bytes);

    auto load_tile = [&](auto& ra_v, auto& rb_v, int& a_kb_st) {
        if (UNI) {
            // tile = channels [u_c0, u_c0+KT) of tap u_tap -- all scalar
            const bool tok = u_tap < ntaps;
            const int tapc = tok ? u_tap : 0;
            const int dy = __builtin_amdgcn_readfirstlane(tap_dy[tapc]);
            const int dx = __builtin_amdgcn_readfirstlane(tap_dx[tapc]);
            const int sdy = DGRAD ? -dy : dy, sdx = DGRAD ? -dx : dx;
            const int toff = ((sdy * p.Wi + sdx) * p.in_ld + u_c0) * 4;
            a_kb_st = RAG ? u_c0 + ga * 4 : 0;                   // K % KT == 0: no channel padding inside a tile
            const bool cok = !RAG || u_c0 + ga * 4 < p.K;        // ragged: this thread's channel group lies behind K in the tap's last tile
#pragma unroll
            for (int j = 0; j < AROWS; ++j) {
                const int iy = a_by[j] + sdy, ix = a_bx[j] + sdx;
                const bool ok = tok && cok && a_rowok[j] && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
                ra_v[j] = mh_buf_load4(rs_in, ok ? a_off[j] + toff : MH_OOB);
            }
            const int wt = pcm ? __builtin_amdgcn_readfirstlane(tap_id[tapc]) : u_tap;
            const int woff = tok ? (!DGRAD ? (wt * p.K + u_c0) * p.N : wt * p.N * p.K + u_c0) * 4 : MH_OOB;
            // ragged: row kk of the tile is real iff u_c0 + kk < K  <=>  b_off + u_c0 * N * 4 < K * N * 4   (b_off = (kk * N + n) * 4, n < N)
            const unsigned rel = RAG ? (unsigned)(u_c0 * p.N * 4) : 0u, lim = (unsigned)(p.K * p.N * 4);
#pragma unroll
            for (int j = 0; j < BITEMS; ++j) {
                // MH_OOB + anything stays out of range (offsets are < 2^31 and num_records < 2^31)
                const bool dead = b_off[j] == MH_OOB || !tok || (RAG && (unsigned)b_off[j] + rel >= lim);
                rb_v[j] = mh_buf_load4(rs_w, dead ? MH_OOB : b_off[j] + woff);
            }
            u_c0 += KT * KG;
            while (u_c0 >= Kc) { u_c0 -= Kc; ++u_tap; }
            return;
        }
        {
            const bool gok = a_tap < ntaps;
            const int tapc = gok ? a_tap : 0;
            const int dy = tap_dy[tapc], dx = tap_dx[tapc];
            const int kbase = a_c4 * 4;
            a_kb_st = kbase;
#pragma unroll
            for (int j = 0; j < AROWS; ++j) {
                bool ok = gok && a_rowok[j];
                int iy, ix;
                if (!DGRAD) {
                    iy = a_by[j] + dy; ix = a_bx[j] + dx;
                } else {
                    const int ny = a_by[j] - dy, nx = a_bx[j] - dx;
                    iy = ny >> sshift; ix = nx >> sshift;                // stride is a power of two
                    ok = ok && ((iy << sshift) == ny) && ((ix << sshift) == nx);
                }
                ok = ok && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
                const int off = ((a_img[j] + iy * p.Wi + ix) * p.in_ld + kbase) * 4;      // bytes (< 2 GiB)
                float4 v;
                if (VEC) {
                    v = mh_buf_load4(rs_in, ok ? off : MH_OOB);   // channel padding is zeroed at store time
                } else {
                    v.x = mh_buf_load1(rs_in, ok ? off : MH_OOB);
                    v.y = mh_buf_load1(rs_in, (ok && kbase + 1 < p.K) ? off + 4 : MH_OOB);
                    v.z = mh_buf_load1(rs_in, (ok && kbase + 2 < p.K) ? off + 8 : MH_OOB);
                    v.w = mh_buf_load1(rs_in, (ok && kbase + 3 < p.K) ? off + 12 : MH_OOB);
                }
                ra_v[j] = v;
            }
        }
#pragma unroll
        for (int j = 0; j < BITEMS; ++j) {
            const int q = tid + 256 * j;
            bool live;
            int k, n;
            if (!DGRAD) {
                int kk, n4;
                live = b_item(j, kk, n4) && (b_tap[j] < ntaps);
                k = b_c4[j] * 4 + (kk & 3);
                n = n0 + n4 * 4;
            } else {
                live = (q < BVEC) && (b_tap[j] < ntaps);
                n = n0 + q / GPT;
                k = b_c4[j] * 4;
            }
            const bool ok = live && k < p.K && n < p.N;
            // forward: w[tap][k][n] (float4 along n) ; dgrad: w[tap][n][k] (float4 along k)
            const int wt = pcm ? tap_id[live ? b_tap[j] : 0] : b_tap[j];
            const int off = (!DGRAD ? (wt * p.K + k) * p.N + n : (wt * p.N + n) * p.K + k) * 4;
            const int lim = !DGRAD ? p.N - n : p.K - k;      // valid elements of the run starting at off
            float4 v;
            if (VEC) {
                v = mh_buf_load4(rs_w, ok ? off : MH_OOB);
            } else {
                v.x = mh_buf_load1(rs_w, ok ? off : MH_OOB);
                v.y = mh_buf_load1(rs_w, (ok && lim > 1) ? off + 4 : MH_OOB);
                v.z = mh_buf_load1(rs_w, (ok && lim > 2) ? off + 8 : MH_OOB);
                v.w = mh_buf_load1(rs_w, (ok && lim > 3) ? off + 12 : MH_OOB);
            }
            rb_v[j] = v;
        }
        a_c4 += GPT * KG;
        while (a_c4 >= p.G) { a_c4 -= p.G; ++a_tap; }
#pragma unroll
        for (int j = 0; j < BITEMS; ++j) {
            b_c4[j] += GPT * KG;
            while (b_c4[j] >= p.G) { b_c4[j] -= p.G; ++b_tap[j]; }
        }
    };

    auto store_tile = [&](int buf, auto& ra_v, auto& rb_v, int a_kb_st) {
#pragma unroll
        for (int j = 0; j < AROWS; ++j) {
            const int r = ra + RPP * j;
            float4 v = ra_v[j];
            if (VEC) {   // zero the channel padding of the last group HERE (after the MFMAs), not at the
                         // load: a select on the loaded value would force an early s_waitcnt vmcnt
                v.y = (a_kb_st + 1 < p.K) ? v.y : 0.f;
                v.z = (a_kb_st + 2 < p.K) ? v.z : 0.f;
                v.w = (a_kb_st + 3 < p.K) ? v.w : 0.f;
            }
            if (r < BM) {
                if constexpr (X3) {
                    uint2 hi, lo;
                    mh_split_bf16x2(v.x, v.y, hi.x, lo.x);
                    mh_split_bf16x2(v.z, v.w, hi.y, lo.y);
                    *reinterpret_cast<uint2*>(&Ah[buf * (BM * LS) + r * LS + ga * 4]) = hi;
                    *reinterpret_cast<uint2*>(&Ah[LO + buf * (BM * LS) + r * LS + ga * 4]) = lo;
                } else if (BF16) *reinterpret_cast<uint2*>(&Ah[buf * (BM * LS) + r * LS + ga * 4]) = make_uint2(mh_pack_bf16(v.x, v.y), mh_pack_bf16(v.z, v.w));
                else *reinterpret_cast<float4*>(&As[buf * (BM * LS) + r * LS + ga * 4]) = v;
            }
        }
        if (BT) {
            // unit = rows kk..kk+3 (items 4u..4u+3) x columns n4*4..+3 : store column c as 4 consecutive k
            unsigned short* Bb = Bh + buf * (BN * LS);
#pragma unroll
            for (int uj = 0; uj < BITEMS / 4; ++uj) {
                int kk, n4;
                if (b_item(4 * uj, kk, n4)) {
                    const float4 v0 = rb_v[4 * uj], v1 = rb_v[4 * uj + 1], v2 = rb_v[4 * uj + 2], v3 = rb_v[4 * uj + 3];
                    // 16-byte chunks XOR-swizzled by the row block: the 16 lanes of a ds_write_b64 group hold 16 different
                    // row blocks n4 at the same kk (rows 4*LS apart = 2 banks -> 8-way conflict unswizzled, 2-way with it)
                    unsigned short* d = Bb + (n4 * 4) * LS + ((((kk >> 3) ^ n4) & (KT / 8 - 1)) << 3) + (kk & 7);
                    if constexpr (X3) {
                        const float e[4][4] = {{v0.x, v1.x, v2.x, v3.x}, {v0.y, v1.y, v2.y, v3.y}, {v0.z, v1.z, v2.z, v3.z}, {v0.w, v1.w, v2.w, v3.w}};
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            uint2 hi, lo;
                            mh_split_bf16x2(e[c][0], e[c][1], hi.x, lo.x);
                            mh_split_bf16x2(e[c][2], e[c][3], hi.y, lo.y);
                            *reinterpret_cast<uint2*>(d + c * LS) = hi;
                            *reinterpret_cast<uint2*>(d + LO + c * LS) = lo;
                        }
                    } else {
                    *reinterpret_cast<uint2*>(d) = make_uint2(mh_pack_bf16(v0.x, v1.x), mh_pack_bf16(v2.x, v3.x));
                    *reinterpret_cast<uint2*>(d + LS) = make_uint2(mh_pack_bf16(v0.y, v1.y), mh_pack_bf16(v2.y, v3.y));
                    *reinterpret_cast<uint2*>(d + 2 * LS) = make_uint2(mh_pack_bf16(v0.z, v1.z), mh_pack_bf16(v2.z, v3.z));
                    *reinterpret_cast<uint2*>(d + 3 * LS) = make_uint2(mh_pack_bf16(v0.w, v1.w), mh_pack_bf16(v2.w, v3.w));
                    }
                }
            }
            return;
        }
        float* Bb = Bs + buf * (BN * LS);
#pragma unroll
        for (int j = 0; j < BITEMS; ++j) {
            const int q = tid + 256 * j;
            if (q < BVEC) {
                if (!DGRAD) {
                    const int kk = q / (BN / 4), n4 = q % (BN / 4);
                    // rows n4*4 .. n4*4+3 share (row>>2) = n4 -> one swizzled column for all four
                    float* d = &Bb[(n4 * 4) * LS + swz_group<GPT>(n4 * 4, kk >> 2) * 4 + (kk & 3)];
                    d[0] = rb_v[j].x; d[LS] = rb_v[j].y; d[2 * LS] = rb_v[j].z; d[3 * LS] = rb_v[j].w;
                } else {
                    const int g = q % GPT, n = q / GPT;
                    if (BF16) *reinterpret_cast<uint2*>(&Bh[buf * (BN * LS) + n * LS + g * 4]) = make_uint2(mh_pack_bf16(rb_v[j].x, rb_v[j].y), mh_pack_bf16(rb_v[j].z, rb_v[j].w));
                    else *reinterpret_cast<float4*>(&Bb[n * LS + swz_group<GPT>(n, g) * 4]) = rb_v[j];
                }
            }
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int ntile = RAG ? (ntaps * (Kc / KT) + KG - 1) / KG
                          : ((ntaps * p.G + GPT - 1) / GPT + KG - 1) / KG;     // K-tiles per group (the same count for every group:
                                                                                // tiles past the end load zeros)
    const int li = lane & 15, lq = lane >> 4;

    auto compute_tile = [&](int buf) {
        if (BF16) {
            const unsigned short* Ab = Ah + buf * (BM * LS) + (wm * MT * 16 + li) * LS + lq * 8;
            const unsigned short* Bb = Bh + buf * (BN * LS) + (wn * NT * 16 + li) * LS + (BT ? 0 : lq * 8);
#pragma unroll
            for (int s = 0; s < KT / 32; ++s) {
                u32x4 a[MT], b[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const u32x4*>(Ab + i * 16 * LS + s * 32);
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if (BT) b[j] = *reinterpret_cast<const u32x4*>(Bb + j * 16 * LS + ((((s * 4 + lq) ^ ((wn * NT * 16 + j * 16 + li) >> 2)) & (KT / 8 - 1)) << 3));
                    else b[j] = *reinterpret_cast<const u32x4*>(Bb + j * 16 * LS + s * 32);
                }
                if constexpr (X3) {
                    u32x4 al[MT], bl[NT];
#pragma unroll
                    for (int i = 0; i < MT; ++i) al[i] = *reinterpret_cast<const u32x4*>(Ab + LO + i * 16 * LS + s * 32);
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        bl[j] = *reinterpret_cast<const u32x4*>(Bb + LO + j * 16 * LS + ((((s * 4 + lq) ^ ((wn * NT * 16 + j * 16 + li) >> 2)) & (KT / 8 - 1)) << 3));
#pragma unroll
                    for (int t = 0; t < 3; ++t)             // t outermost: consecutive MFMAs hit different accumulators
#pragma unroll
                        for (int i = 0; i < MT; ++i)
#pragma unroll
                            for (int j = 0; j < NT; ++j) acc[i][j] = mh_mfma_bf16(t == 0 ? al[i] : a[i], t == 1 ? bl[j] : b[j], acc[i][j]);
                } else {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = mh_mfma_bf16(a[i], b[j], acc[i][j]);
                }
            }
        } else {
        const float* Ab = As + buf * (BM * LS) + (wm * MT * 16 + li) * LS + lq * 4;
        const float* Bb = Bs + buf * (BN * LS) + (wn * NT * 16 + li) * LS;
#pragma unroll
        for (int s = 0; s < KT / 16; ++s) {
            float4 a[MT], b[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const float4*>(Ab + i * 16 * LS + s * 16);
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
        }
    };

    if constexpr (PFD == 4) {
        // Distance 4: tiles t+1 .. t+4 are in flight / parked in four register stages while tile t is multiplied.  The small
        // tiles are bound by ONE exposed global-load round trip per K-tile (~1 us at these grid sizes: 10-13 us for a layer of
        // 5-6 K-steps whatever its pixel count); with four tiles requested up front the round trip is paid once.
        load_tile(ra0, rb0, kb0);            // tile 0
        load_tile(ra1, rb1, kb1);            // tile 1
        load_tile(ra2, rb2, kb2);            // tile 2
        load_tile(ra3, rb3, kb3);            // tile 3
        store_tile(0, ra0, rb0, kb0);
        __syncthreads();
        for (int t = 0; t < ntile; t += 4) {
            load_tile(ra0, rb0, kb0);        // tile t+4
            compute_tile(0);                 // tile t
            store_tile(1, ra1, rb1, kb1);    // tile t+1
            __syncthreads();
            load_tile(ra1, rb1, kb1);        // tile t+5
            if (t + 1 < ntile) compute_tile(1);
            store_tile(0, ra2, rb2, kb2);    // tile t+2
            __syncthreads();
            load_tile(ra2, rb2, kb2);        // tile t+6
            if (t + 2 < ntile) compute_tile(0);
            store_tile(1, ra3, rb3, kb3);    // tile t+3
            __syncthreads();
            load_tile(ra3, rb3, kb3);        // tile t+7
            if (t + 3 < ntile) compute_tile(1);
            store_tile(0, ra0, rb0, kb0);    // tile t+4
            __syncthreads();
        }
    } else if constexpr (PF2) {
        // While tile t is multiplied out of LDS buffer t&1, tile t+1 sits in one register stage (stored to the
        // other LDS buffer after the MFMAs) and tile t+2 is being loaded into the other stage.  Loads past the
        // last tile are issued anyway (tap index out of range => out-of-range buffer offsets => zeros, no
        // memory traffic): unconditional loads keep hipcc's s_waitcnt vmcnt(N) counted.
        load_tile(ra0, rb0, kb0);            // tile 0
        load_tile(ra1, rb1, kb1);            // tile 1
        store_tile(0, ra0, rb0, kb0);
        __syncthreads();
        MH_STAMP(ts2);
        for (int t = 0; t < ntile; t += 2) {
            load_tile(ra0, rb0, kb0);        // tile t+2
            compute_tile(0);                 // tile t
            store_tile(1, ra1, rb1, kb1);    // tile t+1
            __syncthreads();
            load_tile(ra1, rb1, kb1);        // tile t+3
            if (t + 1 < ntile) compute_tile(1);   // tile t+1 (uniform branch, no global loads inside)
            store_tile(0, ra0, rb0, kb0);    // tile t+2
            __syncthreads();
        }
    } else {
        load_tile(ra0, rb0, kb0);
        store_tile(0, ra0, rb0, kb0);
        __syncthreads();
        for (int t = 0; t < ntile; ++t) {
            const int buf = t & 1;
            if (t + 1 < ntile) load_tile(ra0, rb0, kb0);
            compute_tile(buf);
            if (t + 1 < ntile) store_tile(buf ^ 1, ra0, rb0, kb0);
            __syncthreads();
        }
    }

    MH_STAMP(ts3);
    // ---- epilogue: bias + leaky (+ accumulate) (+ fused leaky-grad mask) ------------------
    // Fast path: the accumulator tile goes through LDS (free after the K loop) so that every lane
    // owns 4 consecutive output channels: bias / old / mask are read and the result is written
    // with 16-byte accesses along full NHWC rows (coalesced), all loads of a pass issued together.
    if (p.vecC) {
        constexpr int CS = BN + 4;                         // Cs[BM][CS] fits in the tile LDS
        static_assert(BM * CS <= 2 * TILE_FLOATS, "C staging must fit the group's tile region");
        float* const Cs = smem;                            // KG > 1: every group stages its PARTIAL tile in its own region
        __syncthreads();                                   // everyone is done reading the K-loop tiles
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    Cs[(wm * MT * 16 + i * 16 + lq * 4 + r) * CS + wn * NT * 16 + j * 16 + li] = acc[i][j][r];
        __syncthreads();
        MH_STAMP(ts4);
        if (kg != 0) return;                               // group 0 sums the partial tiles while it reads them (no barrier below)
        constexpr int C4 = BN / 4;                         // float4 per tile row
        constexpr int RP = 256 / C4;                       // tile rows per pass (threads >= RP*C4 idle)
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
            const int m = pcm ? row_m[row < BM ? row : 0] : m0 + row;
            const bool ok = (tid < RP * C4) && (row < BM) && (pcm ? m >= 0 : m < p.M) && (n < p.N);
            float4 v = *reinterpret_cast<const float4*>(&Cs[(row < BM ? row : 0) * CS + c4 * 4]);
#pragma unroll
            for (int g = 1; g < KG; ++g) {
                const float4 w = *reinterpret_cast<const float4*>(&Cs[g * (2 * TILE_FLOATS) + (row < BM ? row : 0) * CS + c4 * 4]);
                v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
            }
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
            if (ok && p.shadow) {
                uint2 hi, lo;
                mh_split_bf16x2(v.x, v.y, hi.x, lo.x);
                mh_split_bf16x2(v.z, v.w, hi.y, lo.y);
                *reinterpret_cast<uint2*>(p.shadow + (int64_t)m * p.shadow_ld + n) = hi;
                if (p.shadow_lo) *reinterpret_cast<uint2*>(p.shadow_lo + (int64_t)m * p.shadow_ld + n) = lo;
            }
        }
#ifdef MH_PHASE_TIMING
        if (p.dbg && threadIdx.x == 0) {
            __builtin_amdgcn_s_waitcnt(0x0F70);            // the stores have been acknowledged
            unsigned long long* d = p.dbg + (size_t)blockIdx.x * 8;
            d[0] = ts0; d[1] = ts1; d[2] = ts2; d[3] = ts3; d[4] = ts4; d[5] = __builtin_amdgcn_s_memtime();
            d[6] = tr0; d[7] = __builtin_amdgcn_s_memrealtime();
        }
#endif
        return;
    }
    static_assert(KG == 1 || VEC, "the split-K variant needs the vector epilogue");
    if (KG > 1) return;                                    // (never dispatched without vecC)
    // generic path (odd channel counts / unaligned rows, e.g. the 1-channel disparity heads)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = pcm ? row_m[wm * MT * 16 + i * 16 + lq * 4 + r] : m0 + wm * MT * 16 + i * 16 + lq * 4 + r;
            if (pcm ? m < 0 : m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = n0 + wn * NT * 16 + j * 16 + li;
                if (n >= p.N) continue;
                float v = acc[i][j][r];
                if (p.bias) v += p.bias[n];
                if (p.alpha != 1.0f) v = v > 0.f ? v : p.alpha * v;
                float* dst = p.out + (int64_t)m * p.out_ld + n;
                if (p.accumulate) v += *dst;
                if (p.mask_ref && n >= p.mask_c0 && n < p.mask_c1) {
                    const float y = p.mask_ref[(int64_t)m * p.mask_ld + n];
                    v *= (y > 0.f) ? 1.0f : p.mask_alpha;
                }
                *dst = v;
            }
        }
    }
}

// ---- single-output-channel forward conv (the disparity heads, Cin -> 1) ---------------------------------------------
// out[p] = act(sum_{tap,k} x[p + tap][k] * w[tap][k] + b) (+ old): a per-pixel dot product of length taps*K, HBM/L2
// bound.  N = 1 makes the HWIO weights k-contiguous, so both operands are read with 16-byte loads here (the tiled
// kernel would fall back to its scalar path and waste 15/16 of a 16-column MFMA tile).  LPP lanes per pixel split the
// channel groups, the weights are staged in LDS once per workgroup, partial sums meet in a shuffle butterfly.
template <int LPP>
__global__ __launch_bounds__(256) void conv_n1_fwd_kernel(ConvArgs p) {
    HIP_DYNAMIC_SHARED(float, wsm)                 // [taps][K]
    const int tid = threadIdx.x;
    const int nw4 = p.taps * p.K / 4;
    constexpr int PPB = 256 / LPP;
    const int sub = tid % LPP;
    const int G4 = p.K >> 2;
    const __amdgpu_buffer_rsrc_t rs_in = mh_make_rsrc(p.in, p.in_bytes);
    const float bias = p.bias ? p.bias[0] : 0.f;
    // The disparity heads (3x3, one channel group per lane, one pixel per lane group: every launch of the MADNet step): all nine taps are requested
    // before the filter bank goes to LDS -- the tap loop below waits for each load before it issues the next, nine dependent memory round trips for
    // 2 KB of work per pixel (round 4: ~7 us per launch at every level)
    if (p.taps == 9 && p.kw == 3 && G4 <= LPP && (int64_t)gridDim.x * PPB >= p.M) {
        const int m = blockIdx.x * PPB + tid / LPP;
        const bool live = m < p.M;
        const int mm = live ? m : 0;
        const int ox = mm % p.Wo;
        const int t2 = mm / p.Wo;
        const int oy = t2 % p.Ho, b = t2 / p.Ho;
        float4 x[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ky = t / 3, kx = t - ky * 3;
            const int iy = oy * p.stride + ky * p.dil - p.pad_t, ix = ox * p.stride + kx * p.dil - p.pad_l;
            const bool ok = live && sub < G4 && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            x[t] = mh_buf_load4(rs_in, ok ? (((b * p.Hi + iy) * p.Wi + ix) * p.in_ld + sub * 4) * 4 : MH_OOB);
        }
        const float old = (live && sub == 0 && p.accumulate) ? p.out[(int64_t)m * p.out_ld] : 0.f;
        for (int i = tid; i < nw4; i += 256) reinterpret_cast<float4*>(wsm)[i] = reinterpret_cast<const float4*>(p.w)[i];
        __syncthreads();
        float acc = 0.f;
        if (sub < G4) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float4 w = *reinterpret_cast<const float4*>(wsm + t * p.K + sub * 4);
                acc += (x[t].x * w.x + x[t].y * w.y) + (x[t].z * w.z + x[t].w * w.w);
            }
        }
#pragma unroll
        for (int o = LPP >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (live && sub == 0) {
            float v = acc + bias;
            if (p.alpha != 1.0f) v = v > 0.f ? v : p.alpha * v;
            v += old;
            p.out[(int64_t)m * p.out_ld] = v;
            if (p.out2) p.out2[(int64_t)m * p.out2_ld] = v;
            if (p.out3) p.out3[(int64_t)m * p.out3_ld] = v;
        }
        return;
    }
    for (int i = tid; i < nw4; i += 256) reinterpret_cast<float4*>(wsm)[i] = reinterpret_cast<const float4*>(p.w)[i];
    __syncthreads();
    for (int m = blockIdx.x * PPB + tid / LPP; m - tid / LPP < p.M; m += gridDim.x * PPB) {     // uniform trip count per wave
        const bool live = m < p.M;
        const int mm = live ? m : 0;
        const int ox = mm % p.Wo;
        const int t2 = mm / p.Wo;
        const int oy = t2 % p.Ho, b = t2 / p.Ho;
        float acc = 0.f;
        if (p.taps == 9 && p.kw == 3) {
            // wide 3x3 heads (DispNet's predictions: 64 .. 1024 channels): the nine taps of a channel group are requested together -- tap-major with one load
            // per iteration this was taps x groups DEPENDENT round trips (144 at K = 1024: 22 us for a 6x20 map).  Another summation order than the loop below.
            int offs[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ky = t / 3, kx = t - ky * 3;
                const int iy = oy * p.stride + ky * p.dil - p.pad_t, ix = ox * p.stride + kx * p.dil - p.pad_l;
                const bool ok = live && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
                offs[t] = ok ? (((b * p.Hi + iy) * p.Wi + ix) * p.in_ld) * 4 : MH_OOB;
            }
            for (int g = sub; g < G4; g += LPP) {
                float4 x[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) x[t] = mh_buf_load4(rs_in, offs[t] == MH_OOB ? MH_OOB : offs[t] + g * 16);
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const float4 w = *reinterpret_cast<const float4*>(wsm + t * p.K + g * 4);
                    acc += (x[t].x * w.x + x[t].y * w.y) + (x[t].z * w.z + x[t].w * w.w);
                }
            }
        } else
        for (int t = 0; t < p.taps; ++t) {
            const int ky = t / p.kw, kx = t - ky * p.kw;
            const int iy = oy * p.stride + ky * p.dil - p.pad_t, ix = ox * p.stride + kx * p.dil - p.pad_l;
            const bool ok = live && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            const int base = ((b * p.Hi + iy) * p.Wi + ix) * p.in_ld;
            for (int g = sub; g < G4; g += LPP) {
                const float4 x = mh_buf_load4(rs_in, ok ? (base + g * 4) * 4 : MH_OOB);
                const float4 w = *reinterpret_cast<const float4*>(wsm + t * p.K + g * 4);
                acc += (x.x * w.x + x.y * w.y) + (x.z * w.z + x.w * w.w);
            }
        }
#pragma unroll
        for (int o = LPP >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (live && sub == 0) {
            float v = acc + bias;
            if (p.alpha != 1.0f) v = v > 0.f ? v : p.alpha * v;
            float* dst = p.out + (int64_t)m * p.out_ld;
            if (p.accumulate) v += *dst;
            *dst = v;
            if (p.out2) p.out2[(int64_t)m * p.out2_ld] = v;
            if (p.out3) p.out3[(int64_t)m * p.out3_ld] = v;
        }
    }
}

static bool conv_n1_ok(const ConvArgs& a) {
    return a.mode == 0 && a.N == 1 && a.vecA && (a.K % 4 == 0) && mh_aligned16(a.w) && !a.mask_ref &&
           (size_t)a.taps * a.K * 4 <= 64 * 1024 && a.M > 0;
}

static int launch_conv_n1(ConvArgs& a, hipStream_t s) {
    const size_t lds = (size_t)a.taps * a.K * sizeof(float);
    const int g4 = a.K / 4;
#define MH_N1(LPPv)                                                                                     \
    {   int grid = mh_cdiv(a.M, 256 / LPPv); if (grid > 256 * 32) grid = 256 * 32;                       \
        hipLaunchKernelGGL((conv_n1_fwd_kernel<LPPv>), dim3(grid), dim3(256), lds, s, a); }
    if (g4 <= 4) MH_N1(4) else if (g4 <= 8) MH_N1(8) else MH_N1(16)
#undef MH_N1
    mh_note_kernel("conv_n1_fwd_kernel K=%d", a.K);
    return mh_check_launch("conv_n1_fwd");
}

// ---- input gradient of a single-output-channel conv (the disparity heads, mode 1 with K = 1) ---------------------------------------
// dx[p][n] = mask(sum_t dz[p + pad - tap_t] * w[t][n]) (+ old): one float of dz per tap, taps*N weights -- no reduction axis worth a
// tile, the launch is bound by the dx store.  A thread owns 4 channels of a pixel (consecutive threads = consecutive channel groups:
// full-line stores), the weights sit in LDS, exact fp32 whatever the mode.  The tiled kernel ran its scalar fp32 path here (9-11 us
// against ~5 us for this one at every level).
__global__ __launch_bounds__(256) void conv_k1_dgrad_kernel(ConvArgs p) {
    HIP_DYNAMIC_SHARED(float, wsm)                 // [taps][N]
    const int tid = threadIdx.x;
    const int nw4 = p.taps * p.N / 4;
    for (int i = tid; i < nw4; i += 256) reinterpret_cast<float4*>(wsm)[i] = reinterpret_cast<const float4*>(p.w)[i];
    __syncthreads();
    const int G4 = p.N >> 2;
    const int total = p.M * G4;
    for (int idx = blockIdx.x * 256 + tid; idx < total; idx += gridDim.x * 256) {
        const int m = idx / G4, n = (idx - m * G4) * 4;
        const int ox = m % p.Wo;
        const int t2 = m / p.Wo;
        const int oy = t2 % p.Ho, b = t2 / p.Ho;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int t = 0; t < p.taps; ++t) {
            const int ky = t / p.kw, kx = t - ky * p.kw;
            const int iy = oy + p.pad_t - ky * p.dil, ix = ox + p.pad_l - kx * p.dil;
            if ((unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi) {
                const float z = p.in[(int64_t)((b * p.Hi + iy) * p.Wi + ix) * p.in_ld];
                const float4 w = *reinterpret_cast<const float4*>(wsm + t * p.N + n);
                v.x += z * w.x; v.y += z * w.y; v.z += z * w.z; v.w += z * w.w;
            }
        }
        float* dst = p.out + (int64_t)m * p.out_ld + n;
        if (p.accumulate) { const float4 o = *reinterpret_cast<const float4*>(dst); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        if (p.mask_ref) {
            const float4 mk = *reinterpret_cast<const float4*>(p.mask_ref + (int64_t)m * p.mask_ld + n);
            v.x *= (mk.x > 0.f || n + 0 < p.mask_c0 || n + 0 >= p.mask_c1) ? 1.0f : p.mask_alpha;
            v.y *= (mk.y > 0.f || n + 1 < p.mask_c0 || n + 1 >= p.mask_c1) ? 1.0f : p.mask_alpha;
            v.z *= (mk.z > 0.f || n + 2 < p.mask_c0 || n + 2 >= p.mask_c1) ? 1.0f : p.mask_alpha;
            v.w *= (mk.w > 0.f || n + 3 < p.mask_c0 || n + 3 >= p.mask_c1) ? 1.0f : p.mask_alpha;
        }
        *reinterpret_cast<float4*>(dst) = v;
        if (p.shadow) *reinterpret_cast<uint2*>(p.shadow + (int64_t)m * p.shadow_ld + n) = make_uint2(mh_pack_bf16(v.x, v.y), mh_pack_bf16(v.z, v.w));
    }
}

static bool conv_k1_dgrad_ok(const ConvArgs& a) {
    return a.mode == 1 && a.K == 1 && a.stride == 1 && a.vecC && mh_aligned16(a.w) && (a.N % 4 == 0) && !a.bias && a.alpha == 1.0f &&
           (size_t)a.taps * a.N * 4 <= 64 * 1024 && a.M > 0 && (int64_t)a.M * (a.N / 4) < (1ll << 31);
}

static int launch_conv_k1_dgrad(ConvArgs& a, hipStream_t s) {
    const size_t lds = (size_t)a.taps * a.N * sizeof(float);
    int grid = mh_cdiv(a.M * (a.N / 4), 256);
    if (grid > 256 * 16) grid = 256 * 16;
    hipLaunchKernelGGL(conv_k1_dgrad_kernel, dim3(grid), dim3(256), lds, s, a);
    mh_note_kernel("conv_k1_dgrad_kernel N=%d", a.N);
    return mh_check_launch("conv_k1_dgrad");
}

// ---- thin full-resolution layers (3x3, Cin <= 32, Cout 16 / 32, many pixels): weights-stationary direct conv ----------
// The tiled kernel spends its time on LDS staging and barriers for K-walks of 1-5 tiles (35 us for 1.1 GFLOP).  Here a
// wave keeps the WHOLE filter bank as bf16 MFMA B operands in registers (taps*Cin <= 288 k-values = 9 steps x NT x 4
// VGPRs), streams 16-pixel tiles, and gathers each A operand straight from global memory into the MFMA layout: lane
// (pixel i, quarter q) needs 8 consecutive k = two 4-channel groups of the flattened (tap, channel) axis = two 16-byte
// loads.  No LDS, no barrier; the 9x tap re-reads are L1/L2 hits.  bf16 mode only (the fp32 parity path keeps the
// tiled exact-fp32 kernel).  DGRAD = stride-1 input gradient (flipped gather, transposed weights, fused mask).
template <int NT, bool DGRAD>
__global__ __launch_bounds__(256) void conv_thin_kernel(ConvArgs p) {
    constexpr int MAXS = 9;
    const int lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
    const int G = p.G;                                   // power of two (1, 2, 4, 8)
    int gshift = 0;
    while ((1 << gshift) < G) ++gshift;
    const int ngroups = p.taps * G;
    const int nsteps = (ngroups + 7) >> 3;
    // filter bank -> registers (B operand: lane (n = li, q) holds k = 8q .. 8q+7 of every 32-k step)
    u32x4 bw[MAXS][NT];
#pragma unroll
    for (int s = 0; s < MAXS; ++s)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float w8[8];
            const int n = j * 16 + li;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int g = 8 * s + 2 * lq + (e >> 2);
                const int tap = g >> gshift, k = (g & (G - 1)) * 4 + (e & 3);
                const bool ok = s < nsteps && g < ngroups && k < p.K && n < p.N;
                w8[e] = ok ? p.w[DGRAD ? (tap * p.N + n) * p.K + k : (tap * p.K + k) * p.N + n] : 0.f;
            }
            bw[s][j] = (u32x4){mh_pack_bf16(w8[0], w8[1]), mh_pack_bf16(w8[2], w8[3]), mh_pack_bf16(w8[4], w8[5]), mh_pack_bf16(w8[6], w8[7])};
        }
    float bias[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bias[j] = (p.bias && j * 16 + li < p.N) ? p.bias[j * 16 + li] : 0.f;
    const __amdgpu_buffer_rsrc_t rs_in = mh_make_rsrc(p.in, p.in_bytes);
    const bool kpad = (p.K & 3) != 0;                    // Cin = 3: the 4th lane of a group is padding
    const int ntiles = (p.M + 15) >> 4;
    const int nwaves = gridDim.x * 4;
    for (int tile = blockIdx.x * 4 + (threadIdx.x >> 6); tile < ntiles; tile += nwaves) {
        const int m = tile * 16 + li;
        const bool rowok = m < p.M;
        const int mm = rowok ? m : 0;
        const int ox = mm % p.Wo, t2 = mm / p.Wo;
        const int oy = t2 % p.Ho, b = t2 / p.Ho;
        const int by = DGRAD ? oy + p.pad_t : oy * p.stride - p.pad_t;
        const int bx = DGRAD ? ox + p.pad_l : ox * p.stride - p.pad_l;
        const int img = b * p.Hi * p.Wi;
        f32x4 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < MAXS; ++s) {
            if (s < nsteps) {                            // uniform
                float4 v[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int g = 8 * s + 2 * lq + h;
                    const int tap = g >> gshift, c4 = g & (G - 1);
                    const int ky = (tap * 11) >> 5, kx = tap - ky * 3;          // tap / 3, tap % 3 for tap < 9 (+ past-the-end groups)
                    const int iy = DGRAD ? by - ky : by + ky, ix = DGRAD ? bx - kx : bx + kx;
                    const bool ok = rowok && g < ngroups && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
                    v[h] = mh_buf_load4(rs_in, ok ? ((img + iy * p.Wi + ix) * p.in_ld + c4 * 4) * 4 : MH_OOB);
                }
                if (kpad) { v[0].w = 0.f; v[1].w = 0.f; v[0].z = p.K > 2 ? v[0].z : 0.f; v[1].z = p.K > 2 ? v[1].z : 0.f;
                            v[0].y = p.K > 1 ? v[0].y : 0.f; v[1].y = p.K > 1 ? v[1].y : 0.f; }
                const u32x4 a = (u32x4){mh_pack_bf16(v[0].x, v[0].y), mh_pack_bf16(v[0].z, v[0].w), mh_pack_bf16(v[1].x, v[1].y), mh_pack_bf16(v[1].z, v[1].w)};
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[j] = mh_mfma_bf16(a, bw[s][j], acc[j]);
            }
        }
        // epilogue: lane holds column n = j*16 + li of the tile rows 4*lq + r
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int mo = tile * 16 + 4 * lq + r;
            if (mo >= p.M) continue;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = j * 16 + li;
                if (n >= p.N) continue;
                float val = acc[j][r] + bias[j];
                if (p.alpha != 1.0f) val = val > 0.f ? val : p.alpha * val;
                float* dst = p.out + (int64_t)mo * p.out_ld + n;
                if (p.accumulate) val += *dst;
                if (p.mask_ref && n >= p.mask_c0 && n < p.mask_c1) val *= (p.mask_ref[(int64_t)mo * p.mask_ld + n] > 0.f) ? 1.0f : p.mask_alpha;
                *dst = val;
            }
        }
    }
}

static std::atomic<int> g_thin_min_m{16384};   // tuning hook (mh_tune_conv_thin): pixel count from which the weights-stationary kernel is used; < 0 = never

static bool conv_thin_ok(const ConvArgs& a) {
    const bool g_pow2 = a.G == 1 || a.G == 2 || a.G == 4 || a.G == 8;
    // Measured (MADNet pyramid, us per launch, tiled kernel -> this one): 3->16 s2 @ 245760 px 44 -> 21, 16->16 33 -> 35,
    // 16->32 s2 @ 61440 px 13.5 -> 40, 32->32 14.6 -> 58: every wave rebuilds the filter bank (8 scalar loads per lane per
    // step), which only amortises for the 2-step image layer.  The default therefore takes Cin <= 4 forward layers only;
    // mh_tune_conv_thin(n > 0) lifts the restriction for tests / experiments.
    const bool wide_ok = g_thin_min_m != 16384;
    return g_thin_min_m >= 0 && a.bf16 && a.vecA && a.kh == 3 && a.kw == 3 && a.dil == 1 && g_pow2 && (a.N == 16 || a.N == 32) &&
           ((a.mode == 0 && a.stride <= 2) || (a.mode == 1 && a.stride == 1)) && a.M >= g_thin_min_m &&
           (wide_ok || (a.G == 1 && a.mode == 0));
}

static int launch_conv_thin(ConvArgs& a, hipStream_t s) {
    int grid = mh_cdiv(mh_cdiv(a.M, 16), 4);
    if (grid > 256 * 8) grid = 256 * 8;
    if (a.mode == 0) {
        if (a.N <= 16) hipLaunchKernelGGL((conv_thin_kernel<1, false>), dim3(grid), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv_thin_kernel<2, false>), dim3(grid), dim3(256), 0, s, a);
    } else {
        if (a.N <= 16) hipLaunchKernelGGL((conv_thin_kernel<1, true>), dim3(grid), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv_thin_kernel<2, true>), dim3(grid), dim3(256), 0, s, a);
    }
    mh_note_kernel("conv_thin_kernel<NT=%d,%s> grid %d", a.N <= 16 ? 1 : 2, a.mode == 0 ? "fwd" : "dgrad", grid);
    return mh_check_launch("conv_thin");
}

// split-bf16 on the tiled kernel: OFF by default.  Measured (profiles/r02_experiments.txt #27): the forward layers that have no patch / bank instance
// are latency bound on their small tiles, and the doubled LDS planes + split conversions cost more than the exact-fp32 MFMAs they replace
// (MADNet FULL 2.02 -> 2.17 ms, DispNet 4.39 -> 4.35 ms).  mh_tune_conv_x3_igemm(1) switches it on (parity-tested).
static std::atomic<int> g_x3_igemm{-1};
static bool conv_x3_igemm_on() {
    int v = g_x3_igemm.load(std::memory_order_relaxed);
    return v > 0;
}
extern "C" int mh_tune_conv_x3_igemm(int on) { g_x3_igemm = on < 0 ? -1 : (on != 0); return 0; }
static std::atomic<bool> g_no_uni{false};    // tuning hook: disable the uniform-tap fast path (A/B experiments)
static std::atomic<bool> g_ragged_uni{true}; // tuning hook: ragged-K layers on the uniform-tap loader (bit 19 of mh_tune_conv_tile's bm clears it)
static std::atomic<bool> g_split_k{true};   // tuning hook: intra-workgroup split-K of the small tiles
static std::atomic<bool> g_parity_classes{true};   // tuning hook: stride-2 dgrad as 4 parity-class sub-problems

template <int WM, int WN, int MT, int NT, int KT, bool DGRAD, bool VEC, bool UNI, bool BF16, int KG = 1, bool X3 = false, bool RAG = false>
int launch_one(ConvArgs& a, hipStream_t s) {
    constexpr int BM = WM * MT * 16, BN = WN * NT * 16;
    constexpr size_t tiles = (BF16 ? (size_t)(2 * (BM + BN) * (KT + 16)) * 2 : (size_t)(2 * (BM + BN) * (KT + 4)) * sizeof(float)) * (X3 ? 2 : 1);
    constexpr size_t lds = KG * tiles + 1536;          // + tap_dy / tap_dx / tap_id [64] and row_m [128]
    static_assert(lds <= 160 * 1024, "tiles do not fit the 160 KiB LDS");
    static std::atomic<uint64_t> attr_done{0};       // LDS > 64 KiB needs the opt-in once per instantiation
    const uint64_t attr_dev = mh_device_bit();
    if (!(attr_done.load(std::memory_order_relaxed) & attr_dev)) {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<WM, WN, MT, NT, KT, DGRAD, VEC, UNI, BF16, KG, X3, RAG>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) { mh_set_error("conv: hipFuncSetAttribute(%d B LDS): %s", (int)lds, hipGetErrorString(e)); return (int)e; }
        }
        attr_done.fetch_or(attr_dev);
    }
    if (a.M < 0) return 0;               // mh_init(): attribute set-up only
    a.mtiles = mh_cdiv(a.M, BM);
    a.ntiles = mh_cdiv(a.N, BN);
    if (DGRAD && a.ncls > 0) {           // parity classes: tiles never straddle two classes
        int t0 = 0;
        for (int c = 0; c < a.ncls; ++c) { a.cls[c].tile0 = t0; t0 += mh_cdiv(a.cls[c].M, BM); }
        a.mtiles = t0;
    }
    const int nwg = a.mtiles * a.ntiles;
    mh_note_kernel("conv_igemm_kernel<%d,%d,%d,%d,KT=%d,%s,%s,%s,%s,KG=%d> tile %dx%d grid %d", WM, WN, MT, NT, KT, DGRAD ? "dgrad" : "fwd",
                   VEC ? "vec" : "scalar", RAG ? "uni-ragged" : UNI ? "uni" : "gen", X3 ? "bf16x3" : BF16 ? "bf16" : "f32", KG, BM, BN, nwg);
    hipLaunchKernelGGL((conv_igemm_kernel<WM, WN, MT, NT, KT, DGRAD, VEC, UNI, BF16, KG, X3, RAG>), dim3(nwg), dim3(256 * KG), lds, s, a);
    return mh_check_launch("conv_igemm");
}

// small latency-bound tile on a grid smaller than the chip with a long K walk: 4 K groups per workgroup
template <int WM, int WN, int MT, int NT, int KT, bool DGRAD, bool UNI, bool BF16, bool X3 = false, bool RAG = false>
int launch_vec(ConvArgs& a, hipStream_t s) {
    constexpr int BM = WM * MT * 16, BN = WN * NT * 16;
    constexpr bool SMALL = (BM * BN <= 32 * 64) && (BM <= 64) && KT == 64;
    constexpr size_t tiles = (BF16 ? (size_t)(2 * (BM + BN) * (KT + 16)) * 2 : (size_t)(2 * (BM + BN) * (KT + 4)) * sizeof(float)) * (X3 ? 2 : 1);
    constexpr int KGV = (4 * tiles + 1536 <= 160 * 1024) ? 4 : 2;       // K groups that fit the LDS
    if constexpr (SMALL) {
        const bool all = a.M < 0;
        const int64_t nwg = all ? 0 : (int64_t)mh_cdiv(a.M, BM) * mh_cdiv(a.N, BN);
        const int ktiles = all ? 0 : mh_cdiv(a.taps * a.G * 4, KT);
        if (all || (g_split_k && a.vecC && nwg <= 256 && ktiles >= 8)) {
            const int rc = launch_one<WM, WN, MT, NT, KT, DGRAD, true, UNI, BF16, KGV, X3, RAG>(a, s);
            if (!all || rc) return rc;
        }
    }
    return launch_one<WM, WN, MT, NT, KT, DGRAD, true, UNI, BF16, 1, X3, RAG>(a, s);
}

// F32 / B16: which arithmetic variants of this (tile, KT) are instantiated (bf16 runs KT = 64 only)
template <int WM, int WN, int MT, int NT, int KT, bool F32, bool B16>
int launch_cfg(ConvArgs& a, hipStream_t s) {
    const bool all = a.M < 0;
    const bool dg = a.mode == 1, vec = a.vecA && a.vecB;
    // uniform-tap fast path: whole K-tiles per tap, no channel padding, unit-stride gather
    const bool uni = vec && (a.K % KT == 0) && (!dg || a.sshift == 0 || a.ncls > 0) && !g_no_uni;
    const bool bf = a.bf16 && vec;                 // the scalar (odd-shape) path stays fp32
    constexpr bool X3FIT = B16 && ((size_t)(2 * (WM * MT * 16 + WN * NT * 16) * (KT + 16)) * 2 * 2 + 1536 <= 160 * 1024);
    const bool x3 = X3FIT && a.x3 && vec && !dg && conv_x3_igemm_on();       // split-bf16 forward on the tiled kernel
    int rc = 0;
    if constexpr (X3FIT) {
        if (all || (x3 && uni)) { rc = launch_vec<WM, WN, MT, NT, KT, false, true, true, true>(a, s); if (!all || rc) return rc; }
        if (all || (x3 && !uni)) { rc = launch_vec<WM, WN, MT, NT, KT, false, false, true, true>(a, s); if (!all || rc) return rc; }
    }
    if constexpr (F32) {
        if (all || (!bf && !dg && vec && uni)) { rc = launch_vec<WM, WN, MT, NT, KT, false, true, false>(a, s); if (!all || rc) return rc; }
        if (all || (!bf && !dg && vec && !uni)) { rc = launch_vec<WM, WN, MT, NT, KT, false, false, false>(a, s); if (!all || rc) return rc; }
        if (all || (!dg && !vec)) { rc = launch_one<WM, WN, MT, NT, KT, false, false, false, false>(a, s); if (!all || rc) return rc; }
        if (all || (!bf && dg && vec && uni)) { rc = launch_vec<WM, WN, MT, NT, KT, true, true, false>(a, s); if (!all || rc) return rc; }
        if (all || (!bf && dg && vec && !uni)) { rc = launch_vec<WM, WN, MT, NT, KT, true, false, false>(a, s); if (!all || rc) return rc; }
        if (all || (dg && !vec)) { rc = launch_one<WM, WN, MT, NT, KT, true, false, false, false>(a, s); if (!all || rc) return rc; }
    }
    if constexpr (B16) {
        // ragged K on the uniform-tap loader: from 4 K-tiles per tap on (the padding of the last tile is then < 25 % of the walk)
        const bool rag = bf && !dg && !uni && !x3 && (a.K % KT != 0) && a.K > 4 * KT && !g_no_uni && g_ragged_uni;
        if (all || rag) { rc = launch_vec<WM, WN, MT, NT, KT, false, true, true, false, true>(a, s); if (!all || rc) return rc; }
        if (all || (bf && !dg && uni)) { rc = launch_vec<WM, WN, MT, NT, KT, false, true, true>(a, s); if (!all || rc) return rc; }
        if (all || (bf && !dg && !uni)) { rc = launch_vec<WM, WN, MT, NT, KT, false, false, true>(a, s); if (!all || rc) return rc; }
        if (all || (bf && dg && uni)) { rc = launch_vec<WM, WN, MT, NT, KT, true, true, true>(a, s); if (!all || rc) return rc; }
        if (all || (bf && dg && !uni)) { rc = launch_vec<WM, WN, MT, NT, KT, true, false, true>(a, s); if (!all || rc) return rc; }
    }
    if (!all) { mh_set_error("conv: no kernel variant (bf16=%d vec=%d KT=%d)", (int)bf, (int)vec, KT); return MH_ERR_UNSUPPORTED; }
    return rc;
}

}  // namespace

// ---- tile selection -------------------------------------------------------------------------
// Measured on MI355X (scripts/microbench.py, profiles/r01_microbench.txt): at batch 1 the layers are
// small relative to 256 CUs, so the best tile is the LARGEST one that still yields >= ~480
// workgroups (about 2 per CU, i.e. 2 waves per SIMD to overlap one wave's load/LDS phase with
// another's MFMAs); when no tile reaches that, take the one with the most workgroups.  Big tiles
// use KT=32; the small, latency-bound ones KT=64/128 (fewer barriers, more bytes in flight).
// mh_tune_conv_tile(bm, bn) forces a tile for experiments.
static std::atomic<int> g_force_bm{-1}, g_force_bn{0}, g_force_kt{0};
static int forced_bm() {
    const int v = g_force_bm.load(std::memory_order_relaxed);
    return v < 0 ? 0 : v;
}
extern "C" int mh_tune_conv_thin(int min_pixels) { g_thin_min_m = min_pixels == 0 ? 16384 : min_pixels; return 0; }
extern "C" int mh_tune_conv_tile(int bm, int bn) {
    g_force_bm = bm & 0xffff; g_force_bn = bn & 0xffff; g_force_kt = bn >> 16;
    g_no_uni = ((bm >> 16) & 1) != 0;      // bit 16 of bm: disable the uniform-tap fast path
    g_split_k = ((bm >> 17) & 1) == 0;     // bit 17 of bm: disable the intra-workgroup split-K
    g_parity_classes = ((bm >> 18) & 1) == 0;   // bit 18 of bm: disable the stride-2 parity classes
    g_ragged_uni = ((bm >> 19) & 1) == 0;       // bit 19 of bm: ragged-K layers back on the generic loader
    return 0;
}

struct TileCfg { int bm, bn, kt; };
static const TileCfg kTiles[] = {
    {128, 128, 32}, {128, 96, 32}, {128, 64, 32}, {64, 128, 32}, {64, 96, 32}, {64, 64, 32}, {128, 32, 32},
    {32, 128, 64}, {32, 96, 64}, {32, 64, 64}, {64, 32, 64}, {32, 32, 64}, {128, 16, 32}, {64, 16, 64},
    {32, 64, 128}, {64, 32, 128}, {32, 32, 128},
};

static int conv_dispatch(ConvArgs& a, hipStream_t s) {
    const bool all = a.M < 0;               // mh_init(): touch every instantiation
    int bm = 0, bn = 0, kt = 0;
    if (!all) {
        const int N = a.N;
        const int64_t M = a.M;
        const int ktot = a.taps * a.G * 4;
        if (forced_bm() > 0 && g_force_bn > 0) {
            bm = g_force_bm; bn = g_force_bn; kt = g_force_kt;
        }
        if (bm == 0) {
            // admissible N tiles: the smallest tile >= N (no wasted columns beyond rounding), or 32/64
            // column slices of a wider layer
            const int bn_full = N > 96 ? 128 : (N > 64 ? 96 : (N > 32 ? 64 : (N > 16 ? 32 : 16)));
            int64_t best_w = -1; int best_area = 0;
            for (const TileCfg& t : kTiles) {
                if (t.kt == 128) continue;                                   // only via the latency rule below
                const bool n_ok = (t.bn == bn_full) || (t.bn < bn_full && t.bn >= 32 && bn_full % t.bn == 0 && bn_full != 96) ||
                                  (bn_full == 96 && t.bn == 32);
                if (!n_ok) continue;
                if (forced_bm() > 0 && t.bm != g_force_bm) continue;
                const int64_t w = (int64_t)mh_cdiv(M, t.bm) * mh_cdiv(N, t.bn);
                const int area = t.bm * t.bn;
                const bool enough = w >= 480, best_enough = best_w >= 480;
                bool take;
                if (best_w < 0) take = true;
                else if (enough != best_enough) take = enough;
                else if (enough) take = area > best_area;                    // both fill the chip: bigger tile
                else take = (w > best_w) || (w == best_w && area > best_area);
                if (take) { best_w = w; best_area = area; bm = t.bm; bn = t.bn; kt = t.kt; }
            }
            // latency-bound (few workgroups, long K): deepen the K-tile
            // ... unless the intra-workgroup split-K variant (KT = 64 instances) takes the layer anyway
            const bool split_k = g_split_k && a.vecC && a.vecA && a.vecB && bm * bn <= 32 * 64 && bm <= 64 && best_w <= 256;
            if (best_w < 240 && ktot >= 512 && !split_k)
                for (const TileCfg& t : kTiles) if (t.bm == bm && t.bn == bn && t.kt == 128) kt = 128;
        }
        if (kt == 0) for (const TileCfg& t : kTiles) if (t.bm == bm && t.bn == bn) { kt = t.kt; break; }
        const bool x3f = a.x3 && a.mode == 0 && a.vecA && a.vecB && conv_x3_igemm_on();
        if (x3f && bm == 128 && bn == 128) bn = 64;     // split-bf16: both planes of a 128x128 tile pair do not fit the LDS
        if ((a.bf16 || x3f) && a.vecA && a.vecB) kt = 64;        // bf16 tiles: 64 k-values (128 B) per LDS row
        else if (kt == 64) {                            // fp32 has no KT=64 instance of the big (KT=32) tiles
            bool has = false;
            for (const TileCfg& t : kTiles) if (t.bm == bm && t.bn == bn && t.kt == 64) has = true;
            if (!has) kt = 32;
        }
    }
    int rc = 0;
#define MH_CFG(BMv, BNv, KTv, F32v, B16v, ...)                                      \
    if (all || (bm == BMv && bn == BNv && kt == KTv)) {                             \
        rc = launch_cfg<__VA_ARGS__, KTv, F32v, B16v>(a, s);                        \
        if (!all || rc) return rc;                                                  \
    }
    // fp32 instances
    MH_CFG(128, 128, 32, true, false, 2, 2, 4, 4) MH_CFG(128, 96, 32, true, false, 2, 2, 4, 3) MH_CFG(128, 64, 32, true, false, 2, 2, 4, 2)
    MH_CFG(64, 128, 32, true, false, 1, 4, 4, 2)  MH_CFG(64, 96, 32, true, false, 2, 2, 2, 3)  MH_CFG(64, 64, 32, true, false, 2, 2, 2, 2)
    MH_CFG(128, 32, 32, true, false, 4, 1, 2, 2)  MH_CFG(128, 16, 32, true, false, 4, 1, 2, 1)
    MH_CFG(32, 64, 128, true, false, 2, 2, 1, 2)  MH_CFG(64, 32, 128, true, false, 4, 1, 1, 2) MH_CFG(32, 32, 128, true, false, 2, 2, 1, 1)
    // KT = 64: fp32 for the small tiles, bf16 for every tile
    MH_CFG(32, 128, 64, true, true, 1, 4, 2, 2)   MH_CFG(32, 96, 64, true, true, 2, 2, 1, 3)   MH_CFG(32, 64, 64, true, true, 2, 2, 1, 2)
    MH_CFG(64, 32, 64, true, true, 4, 1, 1, 2)    MH_CFG(32, 32, 64, true, true, 2, 2, 1, 1)   MH_CFG(64, 16, 64, true, true, 4, 1, 1, 1)
    MH_CFG(128, 128, 64, false, true, 2, 2, 4, 4) MH_CFG(128, 96, 64, false, true, 2, 2, 4, 3) MH_CFG(128, 64, 64, false, true, 2, 2, 4, 2)
    MH_CFG(64, 128, 64, false, true, 1, 4, 4, 2)  MH_CFG(64, 96, 64, false, true, 2, 2, 2, 3)  MH_CFG(64, 64, 64, false, true, 2, 2, 2, 2)
    MH_CFG(128, 32, 64, false, true, 4, 1, 2, 2)  MH_CFG(128, 16, 64, false, true, 4, 1, 2, 1)
#undef MH_CFG
    if (all) return 0;
    mh_set_error("conv_dispatch: no tile configuration for bm=%d bn=%d kt=%d", bm, bn, kt);
    return MH_ERR_UNSUPPORTED;
}

// one-time set-up of the >64 KiB dynamic-LDS opt-in for every instantiation (must not happen
// inside a hipGraph capture)
int mh_conv_init() {
    ConvArgs a{};
    a.M = -1; a.N = 1;
    int rc = mh_conv_patch_launch(a, nullptr);
    if (!rc) rc = mh_conv_bank_small_launch(a, nullptr);
    return rc ? rc : conv_dispatch(a, nullptr);
}

struct HeadOuts { float* out2; int out2_ld; float* out3; int out3_ld; const void* in_shadow; const void* mask_shadow = nullptr; int flags = 0; int query = 0; void* out_lo = nullptr; };
int mh_plane_split_one(const float* src, int src_ld, int C, void* hi, void* lo, int dst_ld, int64_t npix, hipStream_t s);      // conv_planes.hip
static int conv_entry(const mh_conv_desc* d, const float* in, const float* w, const float* wt, const void* wb, const float* bias,
                      float* out, const float* mask_ref, void* stream, void* out_shadow = nullptr, const HeadOuts* head = nullptr);
int mh_shadow_cast_one(const float* src, int src_ld, int C, void* dst, int dst_ld, int64_t npix, hipStream_t s);      // wgrad_stream.hip
#ifdef MH_PHASE_TIMING
static unsigned long long* g_conv_dbg = nullptr;
extern "C" int mh_tune_conv_dbg(void* buf) { g_conv_dbg = (unsigned long long*)buf; return 0; }
#endif
extern "C" int mh_conv2d(const mh_conv_desc* d, const float* in, const float* w, const float* bias,
                         float* out, const float* mask_ref, void* stream) {
    return conv_entry(d, in, w, nullptr, nullptr, bias, out, mask_ref, stream);
}
extern "C" int mh_conv2d_wb(const mh_conv_desc* d, const float* in, const float* w, const void* wb, const float* bias,
                            float* out, const float* mask_ref, void* stream) {
    return conv_entry(d, in, w, nullptr, wb, bias, out, mask_ref, stream);
}
extern "C" int mh_conv2d_sh(const mh_conv_desc* d, const float* in, const float* w, const void* wb, const float* bias,
                            float* out, const float* mask_ref, void* out_shadow, void* stream) {
    MH_REQUIRE(!out_shadow || (((uintptr_t)out_shadow) & 15u) == 0, MH_ERR_ALIGN, "mh_conv2d_sh: out_shadow must be 16-byte aligned");
    return conv_entry(d, in, w, nullptr, wb, bias, out, mask_ref, stream, out_shadow);
}
extern "C" int mh_conv2d_sh2(const mh_conv_desc* d, const float* in, const void* in_shadow, const float* w, const void* wb, const float* bias,
                             float* out, const float* mask_ref, void* out_shadow, void* stream) {
    MH_REQUIRE(!out_shadow || (((uintptr_t)out_shadow) & 15u) == 0, MH_ERR_ALIGN, "mh_conv2d_sh2: out_shadow must be 16-byte aligned");
    MH_REQUIRE(!in_shadow || (((uintptr_t)in_shadow) & 15u) == 0, MH_ERR_ALIGN, "mh_conv2d_sh2: in_shadow must be 16-byte aligned");
    const HeadOuts h{nullptr, 0, nullptr, 0, in_shadow};
    return conv_entry(d, in, w, nullptr, wb, bias, out, mask_ref, stream, out_shadow, &h);
}
extern "C" int mh_conv2d_sh3(const mh_conv_desc* d, const float* in, const void* in_shadow, const float* w, const void* wb, const float* bias,
                             float* out, const float* mask_ref, const void* mask_shadow, void* out_shadow, int32_t flags, void* stream) {
    MH_REQUIRE(!out_shadow || (((uintptr_t)out_shadow) & 15u) == 0, MH_ERR_ALIGN, "mh_conv2d_sh3: out_shadow must be 16-byte aligned");
    MH_REQUIRE(!in_shadow || (((uintptr_t)in_shadow) & 15u) == 0, MH_ERR_ALIGN, "mh_conv2d_sh3: in_shadow must be 16-byte aligned");
    MH_REQUIRE(!mask_shadow || (((uintptr_t)mask_shadow) & 15u) == 0, MH_ERR_ALIGN, "mh_conv2d_sh3: mask_shadow must be 16-byte aligned");
    MH_REQUIRE(!(flags & 1) || (out_shadow && !d->accumulate), MH_ERR_ARG, "mh_conv2d_sh3: MH_CONV_SHADOW_ONLY needs out_shadow and no accumulation");
    HeadOuts h{nullptr, 0, nullptr, 0, in_shadow};
    h.mask_shadow = mask_shadow; h.flags = flags;
    return conv_entry(d, in, w, nullptr, wb, bias, out, mask_ref, stream, out_shadow, &h);
}
extern "C" int mh_conv2d_sh4(const mh_conv_desc* d, const float* in, const float* w, const void* wb, const float* bias, float* out, const float* mask_ref,
                             void* out_hi, void* out_lo, void* stream) {
    MH_REQUIRE(out_hi && out_lo, MH_ERR_ARG, "mh_conv2d_sh4: both output planes");
    MH_REQUIRE(mh_aligned16(out_hi) && mh_aligned16(out_lo), MH_ERR_ALIGN, "mh_conv2d_sh4: output planes must be 16-byte aligned");
    HeadOuts h{nullptr, 0, nullptr, 0, nullptr};
    h.out_lo = out_lo;
    return conv_entry(d, in, w, nullptr, wb, bias, out, mask_ref, stream, out_hi, &h);
}
extern "C" int mh_conv2d_takes_shadows(const mh_conv_desc* d, const float* in, const float* w, const void* wb, float* out, const float* mask_ref) {
    HeadOuts h{nullptr, 0, nullptr, 0, nullptr};
    h.query = 1;
    return conv_entry(d, in, w, nullptr, wb, nullptr, out, mask_ref, nullptr, nullptr, &h);
}
extern "C" int mh_conv2d_head(const mh_conv_desc* d, const float* in, const float* w, const float* bias, float* out,
                              float* out2, int32_t out2_ld, float* out3, int32_t out3_ld, void* stream) {
    MH_REQUIRE(d && d->N == 1 && d->mode == 0, MH_ERR_ARG, "mh_conv2d_head: a forward conv with ONE output channel");
    MH_REQUIRE((!out2 || out2_ld >= 1) && (!out3 || out3_ld >= 1), MH_ERR_ARG, "mh_conv2d_head: pixel strides of the extra outputs must be >= 1");
    const HeadOuts h{out2, out2_ld, out3, out3_ld, nullptr};
    return conv_entry(d, in, w, nullptr, nullptr, bias, out, nullptr, stream, nullptr, &h);
}
static int conv_entry(const mh_conv_desc* d, const float* in, const float* w, const float* wt, const void* wb, const float* bias,
                      float* out, const float* mask_ref, void* stream, void* out_shadow, const HeadOuts* head) {
    MH_REQUIRE(d && in && w && out, MH_ERR_ARG, "mh_conv2d: null argument");
    MH_REQUIRE(d->B > 0 && d->Hi > 0 && d->Wi > 0 && d->Ho > 0 && d->Wo > 0 && d->K > 0 && d->N > 0,
               MH_ERR_ARG, "mh_conv2d: non-positive dimension");
    MH_REQUIRE(d->kh > 0 && d->kw > 0 && d->kh * d->kw <= 64, MH_ERR_ARG, "mh_conv2d: kernel %dx%d unsupported", d->kh, d->kw);
    MH_REQUIRE(d->stride >= 1 && d->dil >= 1, MH_ERR_ARG, "mh_conv2d: stride/dilation must be >= 1");
    MH_REQUIRE(d->in_ld >= d->K && d->out_ld >= d->N, MH_ERR_ARG, "mh_conv2d: ld smaller than channel count");
    MH_REQUIRE((int64_t)d->B * d->Ho * d->Wo < (1ll << 31), MH_ERR_ARG, "mh_conv2d: too many output pixels");
    ConvArgs a;
    a.in = in; a.w = w; a.bias = bias; a.out = out; a.mask_ref = mask_ref;
    a.out2 = a.out3 = nullptr; a.out2_ld = a.out3_ld = 0;
    a.in_shadow = nullptr; a.in_shadow_bytes = 0;
    a.mask_shadow = nullptr; a.mask_shadow_bytes = 0; a.mask_shadow_ld = 0; a.no_f32_out = 0;
#ifdef MH_PHASE_TIMING
    a.dbg = g_conv_dbg;
#endif
    // bank contract: forward = mh_pack_weights(trans 0, planes = 2 for precision 2, 1 for precision 1); mode 1 = (trans 1, 1 plane)
    a.wb = (wb && mh_aligned16(wb) && (d->precision == 2 ? d->mode == 0 : d->precision == 1)) ? wb : nullptr;
    a.wb_bytes = a.wb ? (unsigned)mh_pack_bytes(d->kh * d->kw, d->K, d->N, d->precision == 2 ? 2 : 1) : 0u;
    a.in_ld = d->in_ld; a.out_ld = d->out_ld; a.mask_ld = d->mask_ld;
    a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.Ho = d->Ho; a.Wo = d->Wo;
    a.K = d->K; a.N = d->N; a.G = (d->K + 3) / 4; a.taps = d->kh * d->kw;
    a.kh = d->kh; a.kw = d->kw; a.stride = d->stride; a.dil = d->dil; a.pad_t = d->pad_t; a.pad_l = d->pad_l;
    a.mode = d->mode; a.w_trans = d->w_trans; a.accumulate = d->accumulate; a.bf16 = (d->precision == 1); a.x3 = (d->precision == 2);
    MH_REQUIRE(d->precision >= 0 && d->precision <= 2, MH_ERR_ARG, "mh_conv2d: precision must be 0 (fp32), 1 (bf16 MFMA) or 2 (split-bf16)");
    MH_REQUIRE(d->mode == d->w_trans && (d->mode == 0 || d->mode == 1), MH_ERR_UNSUPPORTED,
               "mh_conv2d: supported combinations are mode=0/w_trans=0 (forward) and mode=1/w_trans=1 (dgrad, conv2d_transpose)");
    {
        const int64_t inb = (((int64_t)d->B * d->Hi * d->Wi - 1) * d->in_ld + (int64_t)((d->K + 3) / 4) * 4) * 4;
        const int64_t wb = (int64_t)d->kh * d->kw * d->K * d->N * 4;
        MH_REQUIRE(inb < (1ll << 31) - 64 && wb < (1ll << 31) - 64, MH_ERR_UNSUPPORTED, "mh_conv2d: tensors must be < 2 GiB (32-bit buffer offsets)");
        a.in_bytes = (unsigned)inb; a.w_bytes = (unsigned)wb;
        const int64_t ob = (((int64_t)d->B * d->Ho * d->Wo - 1) * d->out_ld + d->N) * 4;
        const int64_t mb = mask_ref ? (((int64_t)d->B * d->Ho * d->Wo - 1) * d->mask_ld + d->N) * 4 : 0;
        a.vecC = (d->N % 4 == 0) && (d->out_ld % 4 == 0) && mh_aligned16(out) && (!bias || mh_aligned16(bias)) &&
                 (!mask_ref || (d->mask_ld % 4 == 0 && mh_aligned16(mask_ref))) && ob < (1ll << 31) - 64 && mb < (1ll << 31) - 64;
        a.out_bytes = (unsigned)(a.vecC ? ob : 0); a.mask_bytes = (unsigned)(a.vecC ? mb : 0);
        // (the estimators' first layers: 38 / 70 / ... input channels in rows of 40 / 72 / ...: see ConvArgs::vecCpad)
        a.vecCpad = 0;
        const int n4 = (d->N + 3) / 4 * 4;
        if (!a.vecC && d->mode == 1 && d->N % 4 != 0 && !bias && d->out_ld % 4 == 0 && d->out_ld >= n4 && mh_aligned16(out) &&
            (!mask_ref || (d->mask_ld % 4 == 0 && d->mask_ld >= n4 && mh_aligned16(mask_ref)))) {
            const int64_t obp = (((int64_t)d->B * d->Ho * d->Wo - 1) * d->out_ld + n4) * 4;
            const int64_t mbp = mask_ref ? (((int64_t)d->B * d->Ho * d->Wo - 1) * d->mask_ld + n4) * 4 : 0;
            if (obp < (1ll << 31) - 64 && mbp < (1ll << 31) - 64) { a.vecCpad = 1; a.out_bytes = (unsigned)obp; a.mask_bytes = (unsigned)mbp; }
        }
    }
    a.sshift = 0;
    while ((1 << a.sshift) < d->stride) ++a.sshift;
    a.ncls = 0;
    if (d->mode == 1 && d->stride == 2 && g_parity_classes) {
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px) {
                ConvArgs::Cls c{};
                c.py = py; c.px = px;
                c.Hq = (d->Ho - py + 1) / 2; c.Wq = (d->Wo - px + 1) / 2;
                c.M = d->B * c.Hq * c.Wq;
                for (int ky = 0; ky < d->kh; ++ky)
                    for (int kx = 0; kx < d->kw; ++kx) {
                        const int ny = py + d->pad_t - ky * d->dil, nx = px + d->pad_l - kx * d->dil;
                        if ((ny & 1) || (nx & 1)) continue;             // (two's complement: & 1 is the parity of negatives too)
                        if (c.ntaps < 16 && ny / 2 >= -127 && ny / 2 <= 127 && nx / 2 >= -127 && nx / 2 <= 127) {
                            c.dy[c.ntaps] = (signed char)(ny >> 1); c.dx[c.ntaps] = (signed char)(nx >> 1);
                            c.id[c.ntaps] = (unsigned char)(ky * d->kw + kx);
                        }
                        ++c.ntaps;
                    }
                if (c.M > 0 && c.ntaps > 0) a.cls[a.ncls++] = c;
                else if (c.M > 0) { a.ncls = -1; py = px = 2; }          // a class without taps would leave its pixels unwritten
            }
        bool fits = a.ncls > 0;
        for (int c = 0; fits && c < a.ncls; ++c) fits = a.cls[c].ntaps <= 16;
        if (!fits) a.ncls = 0;                                              // fall back to the lattice-mask walk
    }
    MH_REQUIRE(d->mode == 0 || (1 << a.sshift) == d->stride, MH_ERR_UNSUPPORTED, "mh_conv2d: mode 1 needs a power-of-two stride");
    a.M = d->B * d->Ho * d->Wo;
    a.alpha = d->alpha; a.mask_alpha = d->mask_alpha;
    a.mask_c0 = d->mask_c0; a.mask_c1 = (d->mask_c0 == 0 && d->mask_c1 == 0) ? d->N : d->mask_c1;
    // 16-byte vector loads of A need every group start 16B aligned and the full group in-bounds
    a.vecA = mh_aligned16(in) && (d->in_ld % 4 == 0) && (d->in_ld >= a.G * 4);
    a.vecB = mh_aligned16(w) && (d->w_trans ? (d->K % 4 == 0) : (d->N % 4 == 0));
    // bf16 shadow of the output (operand of mh_wgrad_stream): written by the epilogue of the kernel families that have the store, by a cast
    // launch behind the others
    a.shadow = nullptr; a.shadow_ld = (d->N + 31) / 32 * 32; a.shadow_done = 0;
    a.shadow_lo = nullptr; a.shadow_lo_done = 0;
    void* const out_lo = head ? head->out_lo : nullptr;
    hipStream_t hs = (hipStream_t)stream;
    int rc;
    if (head && head->query) {
        // mh_conv2d_takes_shadows: would this launch stage the bf16 shadow of its input (and honour the shadow-only options)?  Same dispatch order
        // as below: only the patch-staged input-gradient kernel does.
        if (conv_n1_ok(a) || conv_k1_dgrad_ok(a) || mh_conv_rows_ok(a) || conv_thin_ok(a) || mh_conv_bank_small_ok(a)) return 0;
        if (!(mh_conv_patch_ok(a) && a.mode == 1 && a.bf16)) return 0;
        // (the same byte bounds as the launch below: a shadow of 2 GiB or more is not staged)
        const int64_t sbq = (int64_t)d->B * d->Hi * d->Wi * ((d->K + 31) / 32 * 32) * 2, mbq = (int64_t)d->B * d->Ho * d->Wo * ((d->N + 31) / 32 * 32) * 2;
        if (sbq >= (1ll << 31) - 64) return 0;
        const bool mask_sh = mask_ref && mbq < (1ll << 31) - 64 && d->mask_c0 == 0 && (d->mask_c1 == 0 || d->mask_c1 == d->N);
        return 1 | (mask_sh ? 2 : 0);
    }
    // MH_CONV_IN_F32_STALE / MH_CONV_MASK_F32_STALE: the caller elided the fp32 tensor (only its bf16 shadow is valid): a launch whose kernel would read the
    // fp32 tensor is refused -- the dispatch is re-decided on every eager replay from process-wide tuning hooks, so a plan recorded under other
    // settings must fail loudly instead of reading a tensor nobody stored
    auto stale_ok = [&]() -> bool {
        if (!head) return true;
        if ((head->flags & MH_CONV_IN_F32_STALE) && !a.in_shadow) { mh_set_error("mh_conv2d_sh3: the fp32 input was elided (MH_CONV_IN_F32_STALE) but the dispatched kernel does not stage in_shadow"); return false; }
        if ((head->flags & MH_CONV_MASK_F32_STALE) && mask_ref && !a.mask_shadow) { mh_set_error("mh_conv2d_sh3: the fp32 mask was elided (MH_CONV_MASK_F32_STALE) but the dispatched kernel does not read mask_shadow"); return false; }
        return true;
    };
    if (head && (head->out2 || head->out3)) {
        MH_REQUIRE(conv_n1_ok(a), MH_ERR_UNSUPPORTED, "mh_conv2d_head: the layer does not fit the single-output-channel kernel (Cin %% 4, aligned operands, <= 64 KB of weights)");
        a.out2 = head->out2; a.out2_ld = head->out2_ld; a.out3 = head->out3; a.out3_ld = head->out3_ld;
    }
    const bool stale_any = head && (head->flags & (MH_CONV_IN_F32_STALE | MH_CONV_MASK_F32_STALE));
    if (stale_any && (conv_n1_ok(a) || conv_k1_dgrad_ok(a) || (!mh_conv_rows_ok(a) && (conv_thin_ok(a) || mh_conv_bank_small_ok(a))))) { stale_ok(); rc = MH_ERR_UNSUPPORTED; }
    else if (conv_n1_ok(a)) rc = launch_conv_n1(a, hs);
    else if (conv_k1_dgrad_ok(a)) { a.shadow = (unsigned short*)out_shadow; a.shadow_done = 1; rc = launch_conv_k1_dgrad(a, hs); }
    else if (mh_conv_rows_ok(a)) {
        a.shadow = (unsigned short*)out_shadow; a.shadow_done = 1;
        if (head && head->in_shadow && head->mask_shadow && mask_ref && a.mode == 1 && a.bf16 && d->mask_c0 == 0 && (d->mask_c1 == 0 || d->mask_c1 == d->N)) {
            // input AND mask from bf16 shadows (the row kernel takes both or neither)
            const int64_t sb = (int64_t)d->B * d->Hi * d->Wi * ((d->K + 31) / 32 * 32) * 2, mb = (int64_t)d->B * d->Ho * d->Wo * ((d->N + 31) / 32 * 32) * 2;
            if (sb < (1ll << 31) - 64 && mb < (1ll << 31) - 64) {
                a.in_shadow = (const unsigned short*)head->in_shadow; a.in_shadow_bytes = (unsigned)sb;
                a.mask_shadow = (const unsigned short*)head->mask_shadow; a.mask_shadow_bytes = (unsigned)mb; a.mask_shadow_ld = (d->N + 31) / 32 * 32;
            }
        }
        rc = stale_ok() ? mh_conv_rows_launch(a, hs) : MH_ERR_UNSUPPORTED;
    }
    else if (conv_thin_ok(a)) rc = launch_conv_thin(a, hs);
    else if (mh_conv_bank_small_ok(a)) { a.shadow = (unsigned short*)out_shadow; a.shadow_done = 1; rc = mh_conv_bank_small_launch(a, hs); }
    else if (mh_conv_patch_ok(a)) {
        a.shadow = (unsigned short*)out_shadow; a.shadow_done = 1;
        if (head && head->in_shadow && a.mode == 1 && a.bf16) {          // (only this family stages a shadow; the others read the fp32 tensor)
            const int64_t sb = (int64_t)d->B * d->Hi * d->Wi * ((d->K + 31) / 32 * 32) * 2;
            if (sb < (1ll << 31) - 64) { a.in_shadow = (const unsigned short*)head->in_shadow; a.in_shadow_bytes = (unsigned)sb; }
        }
        if (head && a.mode == 1 && a.bf16) {
            const int sld = (d->N + 31) / 32 * 32;
            const int64_t mb = (int64_t)d->B * d->Ho * d->Wo * sld * 2;
            if (head->mask_shadow && mask_ref && mb < (1ll << 31) - 64 && d->mask_c0 == 0 && (d->mask_c1 == 0 || d->mask_c1 == d->N)) {
                a.mask_shadow = (const unsigned short*)head->mask_shadow; a.mask_shadow_bytes = (unsigned)mb; a.mask_shadow_ld = sld;
            }
            if ((head->flags & 1) && out_shadow && !d->accumulate) a.no_f32_out = 1;
        }
        rc = stale_ok() ? mh_conv_patch_launch(a, hs) : MH_ERR_UNSUPPORTED;
    }
    else if (!stale_ok()) rc = MH_ERR_UNSUPPORTED;
    else {
        if (a.vecC) {      // the tiled kernel's vector epilogue has the stores (hi and, where asked for, lo)
            a.shadow = (unsigned short*)out_shadow; a.shadow_done = 1;
            if (out_shadow && out_lo) { a.shadow_lo = (unsigned short*)out_lo; a.shadow_lo_done = 1; }
        }
        rc = conv_dispatch(a, hs);
    }
    if (!rc && out_shadow && out_lo && !a.shadow_lo_done)       // a family without the lo store: one split launch behind it (hi rewritten with the same bits)
        rc = mh_plane_split_one(out, d->out_ld, d->N, out_shadow, out_lo, a.shadow_ld, (int64_t)d->B * d->Ho * d->Wo, hs);
    else if (!rc && out_shadow && !a.shadow_done)
        rc = mh_shadow_cast_one(out, d->out_ld, d->N, out_shadow, a.shadow_ld, (int64_t)d->B * d->Ho * d->Wo, hs);
    return rc;
}
