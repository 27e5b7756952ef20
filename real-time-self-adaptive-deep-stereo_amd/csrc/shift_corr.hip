// shift_corr.hip -- the reference's native operator pair under its LITERAL launcher signature, for a caller that binds
// the custom op exactly like sharedLayers.correlation_native does (Nets/sharedLayers.py:31-39):
//     ShiftCorrKernelLauncher(in0, in1, max_disp, batch, in_h, in_w_padded, channels, out)        Nets/Native/shift_corr.cc:22-23
//     ShiftCorrGradKernelLauncher(in0, in1, grad, max_disp, batch, h, padded_w, channels, o0, o1) Nets/Native/shift_corr.cc:58-60
// Layout contract of those launchers (shift_corr.cc:36-56, shift_corr.cu.cc:193-233): in0 / in1 are NHWC with W ALREADY
// zero-padded by max_disp on both sides (the tf.pad calls of correlation_native), `out` / `grad` are NCHW
// [batch, 2*max_disp+1, in_h, W] with W = in_w_padded - 2*max_disp (the caller transposes, sharedLayers.py:37):
//     out[b][d][y][x] = mean_c in0[b][y][x + max_disp][c] * in1[b][y][x + d][c]                   (shift_corr.cu.cc:27-66)
// The engines do not use this file: they call mh_corr_fwd / mh_corr_bwd (un-padded NHWC in, NHWC out written straight
// into the concat buffer).  The gradient here is the gradient of that formula with respect to the PADDED inputs, laid
// out like the inputs (NHWC, padded) -- not the arithmetic of CorrelateDataBackward0/1, which reads in0 for both
// operands and writes NCHW offsets into NHWC tensors.
#include "mh_common.h"

namespace {

struct ShiftArgs {
    const float* in0; const float* in1; const float* grad;
    float* out; float* out0; float* out1;
    int B, H, Wp, W, C, md, D;
    unsigned in_bytes, g_bytes;
    float inv_c;
};

// One workgroup = 64 consecutive pixels of one image row, 4 lanes per pixel split the 4-channel groups; the D shifts are
// walked in chunks of DT accumulators (the left pixel is re-read per chunk from L1).  Stores: for a fixed d the 16 pixels
// of a wave are consecutive floats of one NCHW row.
template <int DT, bool VEC>
__global__ __launch_bounds__(256) void shift_corr_fwd_kernel(ShiftArgs p) {
    const int tid = threadIdx.x;
    const int sub = tid & 3, px = tid >> 2;
    const int segs = (p.W + 63) / 64;
    int bid = blockIdx.x;
    const int seg = bid % segs; bid /= segs;
    const int y = bid % p.H;
    const int b = bid / p.H;
    const int x = seg * 64 + px;
    const bool live = x < p.W;
    const __amdgpu_buffer_rsrc_t r0 = mh_make_rsrc(p.in0, p.in_bytes);
    const __amdgpu_buffer_rsrc_t r1 = mh_make_rsrc(p.in1, p.in_bytes);
    const int row = ((b * p.H + y) * p.Wp) * p.C;                 // element offset of padded pixel 0 of this row
    const int G = (p.C + 3) >> 2;
    for (int d0 = 0; d0 < p.D; d0 += DT) {
        float acc[DT];
#pragma unroll
        for (int j = 0; j < DT; ++j) acc[j] = 0.f;
        for (int g = sub; g < G; g += 4) {
            const int c = g * 4;
            float4 l;
            const int lo = (row + (x + p.md) * p.C + c) * 4;
            if (VEC) l = mh_buf_load4(r0, live ? lo : MH_OOB);
            else {
                l.x = mh_buf_load1(r0, live ? lo : MH_OOB);
                l.y = mh_buf_load1(r0, (live && c + 1 < p.C) ? lo + 4 : MH_OOB);
                l.z = mh_buf_load1(r0, (live && c + 2 < p.C) ? lo + 8 : MH_OOB);
                l.w = mh_buf_load1(r0, (live && c + 3 < p.C) ? lo + 12 : MH_OOB);
            }
#pragma unroll
            for (int j = 0; j < DT; ++j) {
                const bool ok = live && d0 + j < p.D;
                const int ro = (row + (x + d0 + j) * p.C + c) * 4;
                float4 r;
                if (VEC) r = mh_buf_load4(r1, ok ? ro : MH_OOB);
                else {
                    r.x = mh_buf_load1(r1, ok ? ro : MH_OOB);
                    r.y = mh_buf_load1(r1, (ok && c + 1 < p.C) ? ro + 4 : MH_OOB);
                    r.z = mh_buf_load1(r1, (ok && c + 2 < p.C) ? ro + 8 : MH_OOB);
                    r.w = mh_buf_load1(r1, (ok && c + 3 < p.C) ? ro + 12 : MH_OOB);
                }
                acc[j] += (l.x * r.x + l.y * r.y) + (l.z * r.z + l.w * r.w);
            }
        }
#pragma unroll
        for (int j = 0; j < DT; ++j) {
            float v = acc[j];
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            if (live && sub == 0 && d0 + j < p.D)
                p.out[(((int64_t)b * p.D + d0 + j) * p.H + y) * p.W + x] = v * p.inv_c;
        }
    }
}

// thread = (padded pixel, 4-channel group): both input gradients in one pass.
//   d_in0[b][y][x+md][c] = (1/C) sum_d grad[b][d][y][x]      * in1[b][y][x+d][c]          (0 in the pad columns)
//   d_in1[b][y][xp][c]   = (1/C) sum_d grad[b][d][y][xp-d]   * in0[b][y][xp-d+md][c]      (0 <= xp-d < W)
template <bool VEC>
__global__ __launch_bounds__(256) void shift_corr_grad_kernel(ShiftArgs p) {
    const int G = (p.C + 3) >> 2;
    const int64_t total = (int64_t)p.B * p.H * p.Wp * G;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int g = (int)(idx % G);
    int64_t t = idx / G;
    const int xp = (int)(t % p.Wp); t /= p.Wp;
    const int y = (int)(t % p.H);
    const int b = (int)(t / p.H);
    const int c = g * 4;
    const __amdgpu_buffer_rsrc_t r0 = mh_make_rsrc(p.in0, p.in_bytes);
    const __amdgpu_buffer_rsrc_t r1 = mh_make_rsrc(p.in1, p.in_bytes);
    const int row = ((b * p.H + y) * p.Wp) * p.C;
    auto ld = [&](const __amdgpu_buffer_rsrc_t& r, int xq) -> float4 {
        const int o = (row + xq * p.C + c) * 4;
        float4 v;
        if (VEC) v = mh_buf_load4(r, o);
        else {
            v.x = mh_buf_load1(r, o);
            v.y = mh_buf_load1(r, c + 1 < p.C ? o + 4 : MH_OOB);
            v.z = mh_buf_load1(r, c + 2 < p.C ? o + 8 : MH_OOB);
            v.w = mh_buf_load1(r, c + 3 < p.C ? o + 12 : MH_OOB);
        }
        return v;
    };
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    const int x = xp - p.md;
    const int64_t gplane = (int64_t)p.H * p.W;
    const float* gb = p.grad + ((int64_t)b * p.D * p.H + y) * p.W;
    for (int d = 0; d < p.D; ++d) {
        if (x >= 0 && x < p.W) {
            const float gv = gb[d * gplane + x];
            const float4 r = ld(r1, x + d);
            a0.x += gv * r.x; a0.y += gv * r.y; a0.z += gv * r.z; a0.w += gv * r.w;
        }
        const int xs = xp - d;
        if (xs >= 0 && xs < p.W) {
            const float gv = gb[d * gplane + xs];
            const float4 l = ld(r0, xs + p.md);
            a1.x += gv * l.x; a1.y += gv * l.y; a1.z += gv * l.z; a1.w += gv * l.w;
        }
    }
    float* o0 = p.out0 + (int64_t)row + (int64_t)xp * p.C + c;
    float* o1 = p.out1 + (int64_t)row + (int64_t)xp * p.C + c;
    const float s = p.inv_c;
    if (VEC) {
        *reinterpret_cast<float4*>(o0) = make_float4(a0.x * s, a0.y * s, a0.z * s, a0.w * s);
        *reinterpret_cast<float4*>(o1) = make_float4(a1.x * s, a1.y * s, a1.z * s, a1.w * s);
    } else {
        const float v0[4] = {a0.x, a0.y, a0.z, a0.w}, v1[4] = {a1.x, a1.y, a1.z, a1.w};
        for (int e = 0; e < 4 && c + e < p.C; ++e) { o0[e] = v0[e] * s; o1[e] = v1[e] * s; }
    }
}

int check_common(const char* who, const void* a, const void* b, int max_disp, int batch, int h, int wp, int c, ShiftArgs& p) {
    MH_REQUIRE(a && b, MH_ERR_ARG, "%s: null argument", who);
    MH_REQUIRE(max_disp >= 0 && batch > 0 && h > 0 && c > 0 && wp > 2 * max_disp, MH_ERR_ARG,
               "%s: need batch, height, channels > 0 and padded width > 2*max_disp", who);
    const int64_t bytes = (int64_t)batch * h * wp * c * 4;
    MH_REQUIRE(bytes < (1ll << 31) - 64, MH_ERR_UNSUPPORTED, "%s: tensors must be < 2 GiB (32-bit buffer offsets)", who);
    p.B = batch; p.H = h; p.Wp = wp; p.W = wp - 2 * max_disp; p.C = c; p.md = max_disp; p.D = 2 * max_disp + 1;
    p.in_bytes = (unsigned)bytes; p.inv_c = 1.0f / (float)c;
    return 0;
}

}  // namespace

extern "C" int mh_shift_corr(const float* in0, const float* in1, int32_t max_disp, int32_t batch, int32_t in_h,
                             int32_t in_w_padded, int32_t channels, float* out, void* stream) {
    ShiftArgs p{};
    if (int rc = check_common("mh_shift_corr", in0, in1, max_disp, batch, in_h, in_w_padded, channels, p)) return rc;
    MH_REQUIRE(out, MH_ERR_ARG, "mh_shift_corr: null output");
    p.in0 = in0; p.in1 = in1; p.out = out;
    const bool vec = (channels % 4 == 0) && mh_aligned16(in0) && mh_aligned16(in1);
    const int grid = batch * in_h * ((p.W + 63) / 64);
    hipStream_t s = (hipStream_t)stream;
    if (p.D <= 5) {
        if (vec) hipLaunchKernelGGL((shift_corr_fwd_kernel<5, true>), dim3(grid), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((shift_corr_fwd_kernel<5, false>), dim3(grid), dim3(256), 0, s, p);
    } else {
        if (vec) hipLaunchKernelGGL((shift_corr_fwd_kernel<9, true>), dim3(grid), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((shift_corr_fwd_kernel<9, false>), dim3(grid), dim3(256), 0, s, p);
    }
    return mh_check_launch("shift_corr_fwd");
}

extern "C" int mh_shift_corr_grad(const float* in0, const float* in1, const float* grad, int32_t max_disp, int32_t batch,
                                  int32_t height, int32_t padded_width, int32_t channels, float* out0, float* out1,
                                  void* stream) {
    ShiftArgs p{};
    if (int rc = check_common("mh_shift_corr_grad", in0, in1, max_disp, batch, height, padded_width, channels, p)) return rc;
    MH_REQUIRE(grad && out0 && out1, MH_ERR_ARG, "mh_shift_corr_grad: null argument");
    p.in0 = in0; p.in1 = in1; p.grad = grad; p.out0 = out0; p.out1 = out1;
    const bool vec = (channels % 4 == 0) && mh_aligned16(in0) && mh_aligned16(in1) && mh_aligned16(out0) && mh_aligned16(out1);
    const int64_t total = (int64_t)batch * height * padded_width * ((channels + 3) / 4);
    const int grid = mh_cdiv(total, 256);
    hipStream_t s = (hipStream_t)stream;
    if (vec) hipLaunchKernelGGL((shift_corr_grad_kernel<true>), dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((shift_corr_grad_kernel<false>), dim3(grid), dim3(256), 0, s, p);
    return mh_check_launch("shift_corr_grad");
}
