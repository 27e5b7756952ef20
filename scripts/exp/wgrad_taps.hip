// REJECTED EXPERIMENT (round 2, removed from the product in round 3): the all-taps filter-gradient kernel -- nine taps per workgroup, fp32 operands
// converted by loader waves, transposing LDS reads, one barrier per 32-pixel segment.  8-21 % faster than the tiled kernel stand-alone, 3.5 % slower in
// the step (profiles/r02_microbench_wgrad_taps.txt, profiles/r03_pmc_wgrad_tiled_vs_taps.txt).  Superseded by csrc/wgrad_stream.hip (bf16 shadows through
// LDS DMA, wave-private rings, no barrier).  Kept for the record; it needs the definitions of csrc/wgrad.hip (WgradArgs, mh_lds_read_tr16) to build.
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

