// EXPERIMENT RECORD (not built, not shipped): filter gradient fed from planar bf16 shadows -- correct (it passed the parity tests that went with it:
// emulator + MI355X, 6 shapes) but 2-3x SLOWER than wgrad_bf16_kernel: profiles/r02_microbench_wgrad_planar_rejected.txt, profiles/r02_experiments.txt #24.

// wgrad_planar.hip -- filter gradient of the 3x3 (dilated) stride-1 layers from PLANAR bf16 shadows of its two operands.
//
// dW[tap][ci][co] = sum_p x[p + tap][ci] * dz[p][co] is a GEMM whose reduction index is the PIXEL, and an MFMA lane wants 8 consecutive
// reduction elements of one row: with NHWC tensors that is an 8x8 transposition of every operand element, which wgrad_bf16_kernel pays in
// registers + LDS on every pass (10 % MFMA busy, 2-4 waves per SIMD waiting on the load -> convert -> LDS -> barrier chain;
// profiles/r02_pmc_roofline.json).  Here the transposition is paid ONCE per tensor by planar_kernel -- fp32 [B][H][W][C] -> bf16
// [B][C][H + 2 pad][Wp], zero border, x contiguous -- and the filter-gradient kernel has no LDS at all: lane (i, g) of an A fragment
// loads the 8 pixels (y + dy, x0 + 8 g + dx .. + 7) of channel ci0 + i with ONE 16-byte load (two + a funnel shift when dx is odd: bf16
// pixels are 2 bytes), a B fragment the same from the dz shadow, three register stages run ahead of the MFMAs, no barrier.
// Same operand rounding as wgrad_bf16_kernel (RNE to bf16), another summation order.  The bias gradient stays with mh_bias_grad.
#include "mh_common.h"
#include <stdlib.h>

namespace {

struct PlanarWgradArgs {
    const void* xT; const void* dzT; float* ws;
    unsigned x_bytes, dz_bytes;
    int B, H, W, K, N, dil, pad;
    int Hp, Wp;                 // padded plane height / row pitch (pixels)
    int ktiles, ntiles, splits, nchunk, chunk_per_split, cpr;   // cpr = 32-pixel chunks per row
};

// one workgroup = (tap, 128-row ci tile, 128-column co tile, split): 8 waves of 32 (ci) x 64 (co)
__global__ __launch_bounds__(512) void wgrad_planar_kernel(PlanarWgradArgs p) {
    constexpr int WN = 2, MT = 2, NT = 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, lq = lane >> 4;
    int bid = mh_xcd_remap(blockIdx.x, gridDim.x);
    const int tap = bid % 9; bid /= 9;
    const int tn = bid % p.ntiles; bid /= p.ntiles;
    const int tk = bid % p.ktiles; bid /= p.ktiles;
    const int split = bid;
    const int ky = tap / 3, kx = tap - ky * 3;
    const int dy = (ky - 1) * p.dil, dx = (kx - 1) * p.dil;
    const __amdgpu_buffer_rsrc_t rs_x = mh_make_rsrc(p.xT, p.x_bytes);
    const __amdgpu_buffer_rsrc_t rs_z = mh_make_rsrc(p.dzT, p.dz_bytes);
    const int plane = p.Hp * p.Wp;                                      // pixels per (b, channel) plane
    // lane-constant parts of the byte offsets (batch 0): channel plane + this lane's 8-pixel group; out-of-range channels read zeros
    int a_off[MT], b_off[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int ci = tk * 128 + (wm * MT + i) * 16 + li;
        a_off[i] = ci < p.K ? (ci * plane + lq * 8) * 2 : MH_OOB;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int co = tn * 128 + (wn * NT + j) * 16 + li;
        b_off[j] = co < p.N ? (co * plane + lq * 8) * 2 : MH_OOB;
    }
    const bool odd = (dx & 1) != 0;
    const unsigned sh = odd ? 16u : 0u;
    const int c0 = split * p.chunk_per_split;
    const int c1 = min(p.nchunk, c0 + p.chunk_per_split);

    u32x4 fa[3][MT], fb[3][NT];
    unsigned fx[3][MT];                                                 // odd shifts: the dword behind the aligned 16 bytes
    // chunk c = (image b, row y, 32-pixel segment s): scalar byte offsets of its first pixel in the two shadows
    auto issue = [&](int st, int c) {
        const bool in = c < c1;
        const int cc = in ? c : c0;
        const int s = cc % p.cpr, r = cc / p.cpr;
        const int y = r % p.H, b = r / p.H;
        const int zrow = ((b * p.N * p.Hp + (y + p.pad)) * p.Wp + p.pad + s * 32) * 2;
        const int xpix = (b * p.K * p.Hp + (y + p.pad + dy)) * p.Wp + p.pad + s * 32 + dx;
        const int xrow = (odd ? xpix - 1 : xpix) * 2;
#pragma unroll
        for (int j = 0; j < NT; ++j)
            fb[st][j] = __builtin_amdgcn_raw_buffer_load_b128(rs_z, (in && b_off[j] != MH_OOB) ? b_off[j] + zrow : MH_OOB, 0, 0);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int o = (in && a_off[i] != MH_OOB) ? a_off[i] + xrow : MH_OOB;
            fa[st][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, o, 0, 0);
            fx[st][i] = __builtin_amdgcn_raw_buffer_load_b32(rs_x, o == MH_OOB ? MH_OOB : o + 16, 0, 0);     // (always: a branch in the loop costs a vmcnt(0) per stage)
        }
    };
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    issue(0, c0);
    issue(1, c0 + 1);
    issue(2, c0 + 2);
    __builtin_amdgcn_s_waitcnt(0x0F70);      // nothing in flight at the loop header the first time: hipcc's waitcnt pass then counts inside the loop (conv_bank_kernel)
    for (int c = c0; c < c1; c += 3) {
#pragma unroll
        for (int st = 0; st < 3; ++st) {
            u32x4 a[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                // odd shift: pixels (x-1 .. x+8) sit in 5 dwords and the window moves by one bf16 (sh = 16); even: sh = 0 returns the low word
                const u32x4 w = fa[st][i];
                a[i][0] = __builtin_amdgcn_alignbit(w[1], w[0], sh);
                a[i][1] = __builtin_amdgcn_alignbit(w[2], w[1], sh);
                a[i][2] = __builtin_amdgcn_alignbit(w[3], w[2], sh);
                a[i][3] = __builtin_amdgcn_alignbit(fx[st][i], w[3], sh);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = mh_mfma_bf16(a[i], fb[st][j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
            issue(st, c + st + 3);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float* const dwb = p.ws + (int64_t)split * (9 * (int64_t)p.K * p.N) + (int64_t)tap * p.K * p.N;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ci = tk * 128 + (wm * MT + i) * 16 + lq * 4 + r;
            if (ci >= p.K) continue;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int co = tn * 128 + (wn * NT + j) * 16 + li;
                if (co < p.N) dwb[(int64_t)ci * p.N + co] = acc[i][j][r];
            }
        }
}

// fp32 NHWC -> bf16 planar shadow (interior only: the zero border is the caller's, written once).  One workgroup = 32 pixels of a row x 64
// channels: coalesced float4 reads along the channels, fp32 tile through LDS, 16-byte stores of 8 pixels of one channel.
__global__ __launch_bounds__(256) void planar_kernel(const mh_planar_seg* __restrict__ segs, int nseg) {
    int lo = 0, hi = nseg - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (segs[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const mh_planar_seg sg = segs[lo];
    __shared__ float tile[32][65];
    const int tid = threadIdx.x;
    const int cblk = (sg.C + 63) >> 6, cpr = (sg.W + 31) >> 5;
    int bid = (int)blockIdx.x - sg.blk0;
    const int cb = bid % cblk; bid /= cblk;
    const int s = bid % cpr; bid /= cpr;
    const int y = bid % sg.H, b = bid / sg.H;
    const int Hp = sg.H + 2 * sg.pad, Wp = (sg.W + 2 * sg.pad + 7) & ~7;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int e = tid + 256 * q;
        const int px = e >> 4, c4 = e & 15;
        const int x = s * 32 + px, c = cb * 64 + c4 * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (x < sg.W && c < sg.C) {
            const float* src = sg.src + ((int64_t)(b * sg.H + y) * sg.W + x) * sg.ld + c;
            if (c + 3 < sg.C && ((sg.ld & 3) == 0) && ((((uintptr_t)sg.src) & 15) == 0)) v = *reinterpret_cast<const float4*>(src);
            else { v.x = src[0]; if (c + 1 < sg.C) v.y = src[1]; if (c + 2 < sg.C) v.z = src[2]; if (c + 3 < sg.C) v.w = src[3]; }
        }
        tile[px][c4 * 4 + 0] = v.x; tile[px][c4 * 4 + 1] = v.y; tile[px][c4 * 4 + 2] = v.z; tile[px][c4 * 4 + 3] = v.w;
    }
    __syncthreads();
    const int ch = tid >> 2, g = tid & 3;
    const int c = cb * 64 + ch;
    if (c < sg.C) {
        unsigned w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = mh_pack_bf16(tile[g * 8 + 2 * k][ch], tile[g * 8 + 2 * k + 1][ch]);
        unsigned short* dst = reinterpret_cast<unsigned short*>(sg.dst) + ((int64_t)(b * sg.C + c) * Hp + y + sg.pad) * Wp + sg.pad + s * 32 + g * 8;
        *reinterpret_cast<u32x4*>(dst) = (u32x4){w[0], w[1], w[2], w[3]};       // past-W pixels of the last segment are zeros (tile) inside the right border
    }
}

}  // namespace

extern "C" int64_t mh_planar_bytes(int32_t B, int32_t H, int32_t W, int32_t C, int32_t pad) {
    const int64_t Wp = (W + 2 * pad + 7) & ~7;
    return (int64_t)B * C * (H + 2 * pad) * Wp * 2 + 64;         // + slack: the odd-shift loads read one dword past the last row
}

extern "C" int mh_to_planar(const mh_planar_seg* segs_device, int32_t nseg, int32_t nblocks, void* stream) {
    MH_REQUIRE(segs_device && nseg > 0 && nblocks > 0, MH_ERR_ARG, "mh_to_planar: empty segment table");
    hipLaunchKernelGGL(planar_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, segs_device, nseg);
    return mh_check_launch("to_planar");
}

extern "C" int mh_conv2d_wgrad_planar(const mh_conv_desc* d, const void* xT, const void* dzT, int32_t pad, float* ws, int32_t* splits, void* stream) {
    MH_REQUIRE(d && splits, MH_ERR_ARG, "mh_conv2d_wgrad_planar: null argument");
    MH_REQUIRE(d->mode == 0 && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad_t == d->dil && d->pad_l == d->dil && d->Hi == d->Ho && d->Wi == d->Wo,
               MH_ERR_UNSUPPORTED, "mh_conv2d_wgrad_planar: stride-1 SAME 3x3 layers only");
    MH_REQUIRE(pad >= d->dil && pad >= 16 && (pad & 7) == 0 && d->Wo + pad >= ((d->Wo + 31) & ~31),
               MH_ERR_ARG, "mh_conv2d_wgrad_planar: pad must be a multiple of 8, >= 16, >= dilation and cover the last 32-pixel segment");
    PlanarWgradArgs a;
    a.xT = xT; a.dzT = dzT; a.ws = ws;
    a.B = d->B; a.H = d->Ho; a.W = d->Wo; a.K = d->K; a.N = d->N; a.dil = d->dil; a.pad = pad;
    a.Hp = a.H + 2 * pad; a.Wp = (a.W + 2 * pad + 7) & ~7;
    const int64_t xb = mh_planar_bytes(a.B, a.H, a.W, a.K, pad), zb = mh_planar_bytes(a.B, a.H, a.W, a.N, pad);
    MH_REQUIRE(xb < (1ll << 31) - 64 && zb < (1ll << 31) - 64, MH_ERR_UNSUPPORTED, "mh_conv2d_wgrad_planar: shadows must be < 2 GiB");
    a.x_bytes = (unsigned)xb; a.dz_bytes = (unsigned)zb;
    a.ktiles = mh_cdiv(a.K, 128); a.ntiles = mh_cdiv(a.N, 128);
    a.cpr = mh_cdiv(a.W, 32);
    a.nchunk = a.B * a.H * a.cpr;
    const int base = 9 * a.ktiles * a.ntiles;
    int sp = mh_cdiv(384, base);
    const int maxs = mh_cdiv(a.nchunk, 6);
    if (sp > maxs) sp = maxs;
    if (sp > 192) sp = 192;
    if (sp < 1) sp = 1;
    a.chunk_per_split = mh_cdiv(a.nchunk, sp);
    a.splits = mh_cdiv(a.nchunk, a.chunk_per_split);
    if (!ws) { *splits = a.splits; return 0; }                      // query
    MH_REQUIRE(*splits == a.splits, MH_ERR_ARG, "mh_conv2d_wgrad_planar: split count %d does not match this geometry (%d)", *splits, a.splits);
    MH_REQUIRE(xT && dzT && mh_aligned16(xT) && mh_aligned16(dzT), MH_ERR_ARG, "mh_conv2d_wgrad_planar: shadows must be 16-byte aligned");
    mh_note_kernel("wgrad_planar_kernel tile 128x128 splits %d grid %d", a.splits, base * a.splits);
    hipLaunchKernelGGL(wgrad_planar_kernel, dim3(base * a.splits), dim3(512), 0, (hipStream_t)stream, a);
    return mh_check_launch("wgrad_planar");
}
