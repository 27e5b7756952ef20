// lib.hip -- host side of libmadnet_hip.so: error reporting, the native plan executor that
// replays a compiled network (array of mh_op records) with one FFI call, hipGraph capture /
// replay of such a plan, and HIP-event timing helpers.
#include "mh_common.h"
#include <stdarg.h>
#include <string.h>
#include <mutex>
#include <vector>

static thread_local char g_err[512] = "";

void mh_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static thread_local char g_last_kernel[160] = "";
void mh_note_kernel(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_kernel, sizeof(g_last_kernel), fmt, ap);
    va_end(ap);
}
extern "C" const char* mh_last_kernel(void) { return g_last_kernel; }

int mh_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        mh_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

int mh_conv_init();
int mh_wgrad_init();
int mh_wgrad_stream_init();
int mh_conv_planes_init();
int mh_corr_init();
static int mh_lanes_init();

extern "C" const char* mh_last_error(void) { return g_err; }
// one-time, capture-unsafe set-up (dynamic-LDS opt-in of every kernel instantiation)
extern "C" int mh_init(void) {
    if (int e = mh_conv_init()) return e;
    if (int e = mh_wgrad_init()) return e;
    if (int e = mh_wgrad_stream_init()) return e;
    if (int e = mh_conv_planes_init()) return e;
    if (int e = mh_corr_init()) return e;
    return mh_lanes_init();          // side streams / events of the plan executor (not creatable inside a capture)
}
extern "C" int mh_abi_version(void) { return MH_ABI_VERSION; }

// ---- deterministic accumulation: the process-wide range table, mirrored into every translation unit that has atomics (mh_common.h) ----------------
extern "C" int mh_det_sync_corr(const void*);
extern "C" int mh_det_sync_ops(const void*);
extern "C" int mh_det_sync_wgrad(const void*);
extern "C" int mh_det_sync_wgrad_stream(const void*);
namespace {
std::mutex g_det_mutex;
mh_det_table g_det_host = {};
int g_det_device = -1;             // the device whose copy of the table holds the ranges (ONE per process: the table is a __device__ symbol, one copy per device)
int det_sync() {
    // upload + synchronise on the OWNING device whatever is current (an engine garbage-collected while another device was current left
    // the owner with a stale entry pointing at a freed twin)
    int cur = 0;
    if (hipError_t e = hipGetDevice(&cur)) { mh_set_error("mh_deterministic: %s", hipGetErrorString(e)); return (int)e; }
    const bool sw = g_det_device >= 0 && g_det_device != cur;
    if (sw) if (hipError_t e = hipSetDevice(g_det_device)) { mh_set_error("mh_deterministic: %s", hipGetErrorString(e)); return (int)e; }
    int rc = 0;
    if (hipError_t e = hipDeviceSynchronize()) { mh_set_error("mh_deterministic: %s (not callable while a stream capture is active)", hipGetErrorString(e)); rc = (int)e; }     // no launch in flight may see half a table
    if (!rc)
        for (auto fn : {mh_det_sync_corr, mh_det_sync_ops, mh_det_sync_wgrad, mh_det_sync_wgrad_stream})
            if (int e = fn(&g_det_host)) { mh_set_error("mh_deterministic: table upload failed (%d)", e); rc = e; break; }
    if (sw) hipSetDevice(cur);
    return rc;
}
}  // namespace
extern "C" int mh_deterministic_add(float* base, int64_t n, void* acc) {
    MH_REQUIRE(base && acc && n > 0, MH_ERR_ARG, "mh_deterministic_add: null range");
    MH_REQUIRE((((uintptr_t)acc) & 7u) == 0, MH_ERR_ALIGN, "mh_deterministic_add: the fixed-point twin must be 8-byte aligned");
    std::lock_guard<std::mutex> g(g_det_mutex);
    MH_REQUIRE(g_det_host.n < 8, MH_ERR_UNSUPPORTED, "mh_deterministic_add: at most 8 ranges");
    int cur = 0;
    if (hipError_t e = hipGetDevice(&cur)) { mh_set_error("mh_deterministic_add: %s", hipGetErrorString(e)); return (int)e; }
    MH_REQUIRE(g_det_host.n == 0 || cur == g_det_device, MH_ERR_UNSUPPORTED,
               "mh_deterministic_add: the ranges of a process live on ONE device (registered on device %d, current device %d)", g_det_device, cur);
    g_det_device = cur;
    const int i = g_det_host.n++;
    g_det_host.lo[i] = base; g_det_host.hi[i] = base + n; g_det_host.acc[i] = (long long*)acc;
    return det_sync();
}
extern "C" int mh_deterministic_remove(float* base) {
    std::lock_guard<std::mutex> g(g_det_mutex);
    int k = 0;
    for (int i = 0; i < g_det_host.n; ++i)
        if (g_det_host.lo[i] != base) { g_det_host.lo[k] = g_det_host.lo[i]; g_det_host.hi[k] = g_det_host.hi[i]; g_det_host.acc[k] = g_det_host.acc[i]; ++k; }
    if (k == g_det_host.n) return 0;
    g_det_host.n = k;
    return det_sync();
}
extern "C" int mh_deterministic_ranges(void) { std::lock_guard<std::mutex> g(g_det_mutex); return g_det_host.n; }
extern "C" int mh_det_ovf_corr(void);
extern "C" int mh_det_ovf_ops(void);
extern "C" int mh_det_ovf_wgrad(void);
extern "C" int mh_det_ovf_wgrad_stream(void);
// 1 if an addend outside the fixed-point twin's range (|v| >= 2^15, or NaN) was saturated since the previous call, 0 if not, < 0 = -hipError.  Synchronises the
// device the ranges live on; not callable inside a stream capture.
extern "C" int mh_deterministic_overflow(void) {
    std::lock_guard<std::mutex> g(g_det_mutex);
    int cur = 0;
    if (hipError_t e = hipGetDevice(&cur)) return -(int)e;
    const bool sw = g_det_device >= 0 && g_det_device != cur;
    if (sw) if (hipError_t e = hipSetDevice(g_det_device)) return -(int)e;
    int any = 0;
    for (auto fn : {mh_det_ovf_corr, mh_det_ovf_ops, mh_det_ovf_wgrad, mh_det_ovf_wgrad_stream}) {
        const int v = fn();
        if (v < 0) { any = v; break; }
        any |= v;
    }
    if (sw) (void)hipSetDevice(cur);
    return any;
}
// host utility for the TensorFlow-checkpoint importer (Data_utils/tf_checkpoint.py): CRC-32C (Castagnoli), bytewise table
extern "C" uint32_t mh_crc32c(const void* data, int64_t n, uint32_t crc) {
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1) ? 0x82F63B78u : 0u);
            table[i] = c;
        }
        ready = true;
    }
    const unsigned char* p = (const unsigned char*)data;
    uint32_t c = crc ^ 0xFFFFFFFFu;
    for (int64_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}
extern "C" int mh_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { mh_set_error("hipGetDeviceCount: %s", hipGetErrorString(e)); return -(int)e; }
    return n;
}

// ---- plan executor ------------------------------------------------------------------------
// Field packing of mh_op per kind (host side: madnet_hip/plan.py must match):
//  CONV      i[0..18] = mh_conv_desc ints in declaration order, i[19]=mask_c0 i[20]=mask_c1 i[22]=precision, f[0]=alpha f[1]=mask_alpha
//            p[0]=in p[1]=w p[2]=bias p[3]=out p[4]=mask_ref
//  WGRAD     same desc; i[21]=dout_ld ; p[0]=in p[1]=dout p[2]=dw p[3]=db
//  CORR_FWD  i: l_ld r_ld out_ld coff B H W C md stride copy_left zero_tail ; p: L R u out
//  CORR_BWD  i: g_ld coff l_ld r_ld dl_ld acc_l dr_ld acc_r acc_u B H W C md stride copy_left precision ; p: g L R dL dR du
//  WARP_FWD  i: img_ld out_ld B H W C ; p: img u out
//  WARP_BWD  i: g_ld img_ld dimg_ld acc_u B H W C ; p: g img u dimg du
//  RESIZE_*  i: B Hi Wi Hr Wr cy cx Ho Wo mode accumulate ; f[0]=mul ; FWD p: in out ; BWD p: g in din
//  PAD       i: B H W C Hp Wp pt pl out_ld ; f: div sub ; p: in out
//  LOSS      i: B H W ; f[0]=grad_scale ; p: left right disp ws result ddisp
//  METRICS   i: B H W ; f[0]=pixel_th ; p: disp gt ws result
//  MOMENTUM  n ; f: lr momentum grad_scale ; p: var accum grad
//  COPY_CH   i: src_ld dst_ld nch accumulate ; n=npix ; f[0]=scale ; p: src dst
//  LEAKY_BWD i: dy_ld y_ld nch ; n=npix ; f[0]=alpha ; p: dy y
//  FILL      n ; f[0]=v ; p: ptr
//  BIAS_GRAD i: dz_ld nch ; n=npix ; p: dz db
static void desc_from_op(const mh_op& o, mh_conv_desc& d) {
    const int32_t* i = o.i;
    d.B = i[0]; d.Hi = i[1]; d.Wi = i[2]; d.Ho = i[3]; d.Wo = i[4]; d.K = i[5]; d.N = i[6];
    d.kh = i[7]; d.kw = i[8]; d.stride = i[9]; d.dil = i[10]; d.pad_t = i[11]; d.pad_l = i[12];
    d.mode = i[13]; d.w_trans = i[14]; d.in_ld = i[15]; d.out_ld = i[16]; d.mask_ld = i[17]; d.accumulate = i[18];
    d.alpha = o.f[0]; d.mask_alpha = o.f[1]; d.mask_c0 = i[19]; d.mask_c1 = i[20]; d.precision = i[22];
}

static int run_op(const mh_op& o, void* s) {
    const int32_t* i = o.i;
    void* const* p = o.p;
    switch (o.kind) {
        case MH_OP_CONV: {
            mh_conv_desc d; desc_from_op(o, d);
            if (i[23] & 32) return mh_conv2d_sh4(&d, (const float*)p[0], (const float*)p[1], p[6], (const float*)p[2], (float*)p[3], (const float*)p[4], p[7], p[5], s);
            if (i[23] & 30) return mh_conv2d_sh3(&d, (const float*)p[0], (i[23] & 1) ? p[5] : nullptr, (const float*)p[1], p[6], nullptr, (float*)p[3], (const float*)p[4],
                                              (i[23] & 2) ? p[2] : nullptr, p[7],
                                              ((i[23] & 4) ? MH_CONV_SHADOW_ONLY : 0) | ((i[23] & 8) ? MH_CONV_IN_F32_STALE : 0) | ((i[23] & 16) ? MH_CONV_MASK_F32_STALE : 0), s);      // (input gradients carry no bias: p[2] = mask shadow)
            if (i[23]) return mh_conv2d_sh2(&d, (const float*)p[0], p[5], (const float*)p[1], p[6], (const float*)p[2], (float*)p[3], (const float*)p[4], p[7], s);
            if (p[7]) return mh_conv2d_sh(&d, (const float*)p[0], (const float*)p[1], p[6], (const float*)p[2], (float*)p[3], (const float*)p[4], p[7], s);
            if (p[6]) return mh_conv2d_wb(&d, (const float*)p[0], (const float*)p[1], p[6], (const float*)p[2], (float*)p[3], (const float*)p[4], s);
            return mh_conv2d(&d, (const float*)p[0], (const float*)p[1], (const float*)p[2], (float*)p[3], (const float*)p[4], s);
        }
        case MH_OP_WGRAD: {
            mh_conv_desc d; desc_from_op(o, d);
            return mh_conv2d_wgrad(&d, (const float*)p[0], (const float*)p[1], i[21], (float*)p[2], (float*)p[3], s);
        }
        case MH_OP_WGRAD_PARTIAL: {
            mh_conv_desc d; desc_from_op(o, d);
            int32_t splits = i[23];
            return mh_conv2d_wgrad_partial(&d, (const float*)p[0], (const float*)p[1], i[21], (float*)p[2], &splits, (float*)p[3], s);
        }
        case MH_OP_SUPERVISED_LOSS:
            return mh_supervised_loss((const float*)p[0], (const float*)p[1], (float*)p[2], (float*)p[3], (float*)p[4], o.f[0], o.f[1], o.f[2], i[0], i[1], i[2], s);
        case MH_OP_ADAM: {        // f = {lr, beta1, beta2, eps}; i[0] = bits of grad_scale
            float gs; memcpy(&gs, &i[0], sizeof gs);
            return mh_adam((float*)p[0], (float*)p[1], (float*)p[2], (const float*)p[3], o.n, (const float*)p[4], o.f[0], o.f[1], o.f[2], o.f[3], gs, s);
        }
        case MH_OP_ADAM_ADVANCE:
            return mh_adam_advance((float*)p[0], o.f[0], o.f[1], s);
        case MH_OP_PROXY_LOSS:
            return mh_proxy_loss((const float*)p[0], (const float*)p[1], (float*)p[2], (float*)p[3], (float*)p[4], o.f[0], o.f[1], i[0], i[1], i[2], s);
        case MH_OP_CORR_WARP_BWD:
            return mh_corr_warp_bwd((const float*)p[0], i[0], i[1], (const float*)p[1], i[2], (const float*)p[2], i[3], (const float*)p[3], i[4],
                                    (const float*)p[4], (float*)p[5], i[5], i[6], (float*)p[6], i[7], (float*)p[7],
                                    i[8], i[9], i[10], i[11], i[12], i[13], i[14], s);
        case MH_OP_SHADOW_CAST:
            return mh_shadow_cast((const mh_shadow_seg*)p[0], i[0], i[1], s);
        case MH_OP_WGRAD_STREAM:
            return mh_wgrad_stream((const mh_wgs_layer*)p[0], i[0], i[1], i[2], i[3], s);
        case MH_OP_HEAD_FWD: {
            mh_conv_desc d; desc_from_op(o, d);
            return mh_conv2d_head(&d, (const float*)p[0], (const float*)p[1], (const float*)p[2], (float*)p[3], (float*)p[4], i[23], (float*)p[5], i[24], s);
        }
        case MH_OP_HEAD_BWD: {
            mh_head_bwd_desc d;
            d.kind = i[0]; d.B = i[1]; d.H = i[2]; d.W = i[3]; d.N = i[4]; d.Hr = i[5]; d.Wr = i[6]; d.cy = i[7]; d.cx = i[8]; d.Ho = i[9]; d.Wo = i[10];
            d.src0_ld = i[11]; d.src1_ld = i[12]; d.dx_ld = i[13]; d.mask_ld = i[14]; d.accumulate_dx = i[15];
            d.mul = o.f[0]; d.mask_alpha = o.f[1];
            return mh_head_bwd(&d, (const float*)p[0], (const float*)p[1], (float*)p[2], p[3], (const float*)p[4], (float*)p[5], (const float*)p[6], p[7], s);
        }
        case MH_OP_CONV_PLANES: {  // i = mh_conv_desc ints as CONV, i[23] = in_pld, i[24] = out_pld ; p: in_hi in_lo bank bias out out_hi out_lo
            mh_conv_desc d; desc_from_op(o, d);
            return mh_conv2d_planes(&d, p[0], p[1], i[23], p[2], (const float*)p[3], (float*)p[4], p[5], p[6], i[24], s);
        }
        case MH_OP_CONV_PLANES_BWD: {   // i = the forward layer's mh_conv_desc ints, i[23] = dz_pld, i[24] = mask_pld, i[25] = dx_pld ; p: dz_hi bank mask_hi dx dx_hi
            mh_conv_desc d; desc_from_op(o, d);
            return mh_conv2d_planes_bwd(&d, p[0], i[23], p[1], p[2], i[24], (float*)p[3], p[4], i[25], s);
        }
        case MH_OP_CONV_IMAGE:       // i: NB H0 W0 C Hp Wp reflect_t reflect_l N stride pad_t pad_l out_ld shadow_ld ; f: div sub alpha ; p: frames w bias out shadow
            return mh_conv_image_fwd((const float*)p[0], i[0], i[1], i[2], i[3], i[4], i[5], i[6], i[7], o.f[0], o.f[1], (const float*)p[1], (const float*)p[2], i[8], i[9], i[10],
                                     i[11], o.f[2], (float*)p[3], i[12], p[4], i[13], s);
        case MH_OP_DET_FLUSH:        // p: dst, twin ; n
            return mh_det_flush((float*)p[0], p[1], o.n, s);
        case MH_OP_ALLREDUCE: {      // p[0] = comm, p[1 .. i[0]] = buffers ; i[1 .. i[0]] = counts
            float* bufs[MH_ALLREDUCE_MAX_BUFS]; int64_t counts[MH_ALLREDUCE_MAX_BUFS];
            const int nb = i[0];
            if (nb < 1 || nb > MH_ALLREDUCE_MAX_BUFS) { mh_set_error("MH_OP_ALLREDUCE: %d buffers", nb); return MH_ERR_ARG; }
            for (int k = 0; k < nb; ++k) { bufs[k] = (float*)p[1 + k]; counts[k] = i[1 + k]; }
            return mh_allreduce_sum(bufs, counts, nb, p[0], s);
        }
        case MH_OP_FETCH_INPUTS: {   // p[0] = table, p[1 .. i[0]] = destinations ; i[1 .. i[0]] = counts
            float* dst[MH_FETCH_MAX]; int64_t cnt[MH_FETCH_MAX];
            const int nb = i[0];
            if (nb < 1 || nb > MH_FETCH_MAX) { mh_set_error("MH_OP_FETCH_INPUTS: %d entries", nb); return MH_ERR_ARG; }
            for (int k = 0; k < nb; ++k) { dst[k] = (float*)p[1 + k]; cnt[k] = i[1 + k]; }
            return mh_fetch_inputs((const mh_input_table*)p[0], dst, cnt, nb, s);
        }
        case MH_OP_STAMP:
            return mh_stamp(p[0], s);
        case MH_OP_PLANE_SPLIT:
            return mh_plane_split((const mh_plane_seg*)p[0], i[0], i[1], s);
        case MH_OP_PACK_W:
            return mh_pack_weights((const mh_pack_seg*)p[0], i[0], i[1], s);
        case MH_OP_WGRAD_REDUCE:
            return mh_wgrad_reduce((const mh_wgrad_seg*)p[0], i[0], i[1], s);
        case MH_OP_CORR_FWD:
            return mh_corr_fwd_prec((const float*)p[0], i[0], (const float*)p[1], i[1], (const float*)p[2], (float*)p[3], i[2], i[3],
                                    i[4], i[5], i[6], i[7], i[8], i[9], i[10], i[11], i[12], s);
        case MH_OP_CORR_BWD:
            return mh_corr_bwd_prec((const float*)p[0], i[0], i[1], (const float*)p[1], i[2], (const float*)p[2], i[3],
                                    (float*)p[3], i[4], i[5], (float*)p[4], i[6], i[7], (float*)p[5], i[8],
                                    i[9], i[10], i[11], i[12], i[13], i[14], i[15], i[16], s);
        case MH_OP_WARP_FWD:
            return mh_warp_fwd((const float*)p[0], i[0], (const float*)p[1], (float*)p[2], i[1], i[2], i[3], i[4], i[5], s);
        case MH_OP_WARP_BWD:
            return mh_warp_bwd((const float*)p[0], i[0], (const float*)p[1], i[1], (const float*)p[2], (float*)p[3], i[2],
                               (float*)p[4], i[3], i[4], i[5], i[6], i[7], s);
        case MH_OP_RESIZE_FWD:
            return mh_resize_fwd((const float*)p[0], (float*)p[1], i[0], i[1], i[2], i[3], i[4], i[5], i[6], i[7], i[8], o.f[0], i[9], s);
        case MH_OP_RESIZE_BWD:
            return mh_resize_bwd((const float*)p[0], (const float*)p[1], (float*)p[2], i[10], i[0], i[1], i[2], i[3], i[4], i[5], i[6],
                                 i[7], i[8], o.f[0], i[9], s);
        case MH_OP_LEVEL_FRONT:
            if (p[8])           // the coarser level's disparity head in the same launch: X, head bank, head bias; i[14] = x_ld, i[15] = K
                return mh_level_front_head_fwd((const float*)p[8], i[14], i[15], (const float*)p[9], (const float*)p[10], (float*)p[0], i[0], i[1], o.f[0],
                                               (const float*)p[1], i[2], (const float*)p[2], i[3], (float*)p[3], i[4], i[5], (float*)p[4], i[6], (float*)p[5],
                                               i[7], i[8], i[9], i[10], i[11], i[12], p[6], p[7], i[13], s);
            return mh_level_front_fwd_planes((const float*)p[0], i[0], i[1], o.f[0], (const float*)p[1], i[2], (const float*)p[2], i[3], (float*)p[3],
                                             i[4], i[5], (float*)p[4], i[6], (float*)p[5], i[7], i[8], i[9], i[10], i[11], i[12], p[6], p[7], i[13], s);
        case MH_OP_RESIZE_IMAGE:
            return mh_resize_image_fwd((const float*)p[0], (float*)p[1], i[0], i[1], i[2], i[3], i[4], i[5], s);
        case MH_OP_PAD_REFLECT:
            return mh_pad_reflect((const float*)p[0], (float*)p[1], i[0], i[1], i[2], i[3], i[4], i[5], i[6], i[7], i[8], o.f[0], o.f[1], s);
        case MH_OP_LOSS:
            return mh_reprojection_loss_phase((const float*)p[0], (const float*)p[1], (const float*)p[2], (float*)p[3], (float*)p[4],
                                              (float*)p[5], o.f[0], i[0], i[1], i[2], i[3], s);
        case MH_OP_METRICS:
            return mh_metrics((const float*)p[0], (const float*)p[1], (float*)p[2], (float*)p[3], o.f[0], i[0], i[1], i[2], s);
        case MH_OP_MOMENTUM:
            return mh_momentum((float*)p[0], (float*)p[1], (const float*)p[2], o.n, o.f[0], o.f[1], o.f[2], s);
        case MH_OP_COPY_CH:
            return mh_copy_channels((const float*)p[0], i[0], (float*)p[1], i[1], o.n, i[2], o.f[0], i[3], s);
        case MH_OP_LEAKY_BWD:
            return mh_leaky_bwd((float*)p[0], i[0], (const float*)p[1], i[1], o.n, i[2], o.f[0], s);
        case MH_OP_FILL:
            return mh_fill((float*)p[0], o.n, o.f[0], s);
        case MH_OP_BIAS_GRAD:
            if (i[2] > 0) return mh_bias_grad_partial((const float*)p[0], i[0], o.n, i[1], (float*)p[1], i[2], s);      // p[1] = workspace [i[2]][nch]
            return mh_bias_grad((const float*)p[0], i[0], o.n, i[1], (float*)p[1], s);
        default:
            mh_set_error("mh_plan_run: unknown op kind %d", o.kind);
            return MH_ERR_ARG;
    }
}

#define MH_HIP(call)                                                        \
    do {                                                                    \
        hipError_t e_ = (call);                                             \
        if (e_ != hipSuccess) {                                             \
            mh_set_error("%s: %s", #call, hipGetErrorString(e_));           \
            return (int)e_;                                                 \
        }                                                                   \
    } while (0)

// Side lanes: library-owned non-blocking streams (per device) + a small round-robin pool of timing-less events.
// An op on lane L>0 is forked from the caller's stream right before it (event record on lane 0 -> lane L waits),
// so it is ordered after everything lane 0 has been given so far and runs concurrently with what follows; an op
// flagged MH_OP_JOIN makes lane 0 wait for every side lane first.  Under hipStreamBeginCapture on the caller's
// stream the same calls become fork/join edges of the captured graph (parallel branches).
namespace {
struct Lanes {
    hipStream_t aux[MH_MAX_LANES] = {};
    hipEvent_t ev[32] = {};
    int next = 0;
    bool ready = false;
};
// per host thread AND per device: two threads that replay plans on one GPU (the demo's grabber / worker pattern) never share a
// side stream or the event ring, so mh_plan_run is re-entrant without a lock ("stateless, re-entrant").  The streams and
// events of a thread go back to a per-device pool when the thread exits and the next new thread takes them over: a process that starts
// many short-lived worker threads (one RealTimeStereo thread per demo run, bench workers, prefetch threads) holds as many lane sets as it
// ever had threads ALIVE at once, not one per thread it ever started.  (Nothing is destroyed at thread exit: the HIP runtime may already be
// shutting down then.)
struct LanePool {
    std::mutex m;
    std::vector<Lanes*> free_[16];
};
LanePool& lane_pool() { static LanePool* p = new LanePool; return *p; }      // (never freed: thread-exit destructors may run after static destructors)
struct ThreadLanes {
    Lanes* l[16] = {};
    ~ThreadLanes() {
        LanePool& P = lane_pool();
        std::lock_guard<std::mutex> g(P.m);
        for (int d = 0; d < 16; ++d)
            if (l[d]) { P.free_[d].push_back(l[d]); l[d] = nullptr; }
    }
};
thread_local ThreadLanes t_lanes;

int lanes_get(Lanes** out) {
    int dev = 0;
    MH_HIP(hipGetDevice(&dev));
    MH_REQUIRE(dev >= 0 && dev < 16, MH_ERR_UNSUPPORTED, "device index %d out of range", dev);
    if (!t_lanes.l[dev]) {
        LanePool& P = lane_pool();
        Lanes* L = nullptr;
        {
            std::lock_guard<std::mutex> g(P.m);
            if (!P.free_[dev].empty()) { L = P.free_[dev].back(); P.free_[dev].pop_back(); }
        }
        if (!L) {
            L = new Lanes;
            for (int k = 1; k < MH_MAX_LANES; ++k) MH_HIP(hipStreamCreateWithFlags(&L->aux[k], hipStreamNonBlocking));
            for (auto& e : L->ev) MH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            L->ready = true;
        }
        t_lanes.l[dev] = L;
    }
    *out = t_lanes.l[dev];
    return 0;
}
}  // namespace
static int mh_lanes_init() { Lanes* L = nullptr; return lanes_get(&L); }
namespace {
int lane_edge(Lanes& L, hipStream_t from, hipStream_t to) {
    hipEvent_t e = L.ev[L.next];
    L.next = (L.next + 1) & 31;
    MH_HIP(hipEventRecord(e, from));
    MH_HIP(hipStreamWaitEvent(to, e, 0));
    return 0;
}
}  // namespace

// consecutive partial-filter-gradient ops of one lane (a batch of independent layers) go out as ONE grouped launch
static int run_wgrad_batch(const mh_op* ops, int m, void* s) {
    mh_wgrad_item items[16];
    for (int k = 0; k < m; ++k) {
        const mh_op& o = ops[k];
        desc_from_op(o, items[k].d);
        items[k].in = (const float*)o.p[0]; items[k].dout = (const float*)o.p[1];
        items[k].ws = (float*)o.p[2]; items[k].db = (float*)o.p[3];
        items[k].dout_ld = o.i[21]; items[k].splits = o.i[23];
        items[k].group_max_m = o.i[24]; items[k].reserved = 0;
    }
    return mh_conv2d_wgrad_partial_group(items, m, s);
}

// `fixed` != null: the side lanes of this run come from that lane set instead of the calling thread's (mh_plans_run: one set per plan)
static int plan_run_impl(const mh_op* ops, int32_t nops, void* stream, Lanes* fixed) {
    MH_REQUIRE(ops || nops == 0, MH_ERR_ARG, "mh_plan_run: null plan");
    Lanes* L = fixed;
    bool dirty[MH_MAX_LANES] = {};
    bool stale[MH_MAX_LANES];          // lane 0 has launched work since this lane's last fork edge
    for (bool& b : stale) b = true;
    hipStream_t main_s = (hipStream_t)stream;
    auto join = [&]() -> int {
        for (int l = 1; l < MH_MAX_LANES; ++l)
            if (dirty[l]) { if (int e = lane_edge(*L, L->aux[l], main_s)) return e; dirty[l] = false; }
        return 0;
    };
    // Side-lane launches are DEFERRED until the next lane-0 op has been launched (their fork edge is taken at the original point).  In a
    // captured graph the first node created under a parent inherits the parent's hardware queue and later children move to another
    // queue behind a cross-queue dependency (~15-20 us, seen as idle time on the critical path after every batch of filter gradients:
    // profiles/r02_experiments.txt #16); this way the critical-path successor is the first child and the side batch pays the hop.
    struct Deferred { int32_t k; int m, lane; };
    Deferred deferred[64];
    int ndef = 0;
    auto flush_deferred = [&]() -> int {
        for (int q = 0; q < ndef; ++q) {
            const Deferred& d = deferred[q];
            const int e = d.m > 1 ? run_wgrad_batch(ops + d.k, d.m, (void*)L->aux[d.lane]) : run_op(ops[d.k], (void*)L->aux[d.lane]);
            if (e) { ndef = 0; return e; }
        }
        ndef = 0;
        return 0;
    };
    for (int32_t k = 0; k < nops; ++k) {
        const int sched = ops[k].i[26];
        const int lane = sched & 0xff;
        int e = 0;
        if (lane >= MH_MAX_LANES) { mh_set_error("lane %d out of range", lane); e = MH_ERR_ARG; }
        if (!e && ndef && ((sched & MH_OP_JOIN) || ((sched >> 16) & 0xff) || ndef >= 60)) e = flush_deferred();
        if (!e && (sched & MH_OP_JOIN)) e = join();
        if (!e && ((sched >> 16) & 0xff)) {
            // join exactly these side lanes -- into lane 0 (an op of lane 0), or into the op's OWN side lane (round 6: the shared-model step's first all-reduce
            // waits for the filter-gradient lanes on its lane while lane 0 walks on into the pyramid's backward pass; the waited-for lanes stay dirty for lane 0)
            if (lane > 0 && !L) e = lanes_get(&L);
            hipStream_t tgt = lane > 0 ? L->aux[lane] : main_s;
            for (int l = 1; l < MH_MAX_LANES && !e; ++l)
                if ((((sched >> 16) >> l) & 1) && l != lane && dirty[l]) {
                    e = lane_edge(*L, L->aux[l], tgt);
                    if (lane == 0) dirty[l] = false;
                }
        }
        // batch = this op + the following partial-filter-gradient ops of the same lane (no join / lane change in between)
        int m = 1;
        if (ops[k].kind == MH_OP_WGRAD_PARTIAL)
            while (m < 16 && k + m < nops && ops[k + m].kind == MH_OP_WGRAD_PARTIAL && (ops[k + m].i[26] & ~MH_OP_NODEFER) == lane) ++m;
        auto run = [&](void* s) -> int { return m > 1 ? run_wgrad_batch(ops + k, m, s) : run_op(ops[k], s); };
        if (!e && lane > 0) {
            if (!L) e = lanes_get(&L);
            if (!e && stale[lane]) { e = lane_edge(*L, main_s, L->aux[lane]); stale[lane] = false; }
            if (!e) {
                dirty[lane] = true;
                // MH_OP_NODEFER is not honoured on an op that joins other side lanes into its own (round 6, scripts/exp/cut_update_probe.py): created at once, such a node is
                // the FIRST child of the fork in a captured graph -- it inherits lane 0's hardware queue and the critical chain moves to another one (+0.4 ms per step
                // measured) -- and a momentum update recorded that way gave weights 8e-6 off the plain step after four replays while the eager run and the deferred
                // form were bit-identical.  Deferred, the op is created behind the next lane-0 op like every other side-lane launch.
                const bool at_once = (sched & MH_OP_NODEFER) && !((sched >> 16) & 0xff);
                if (!at_once) { deferred[ndef].k = k; deferred[ndef].m = m; deferred[ndef].lane = lane; ++ndef; }
                else {
                    if (ndef) e = flush_deferred();          // keep the lane's order
                    if (!e) e = run((void*)L->aux[lane]);
                }
            }
        } else if (!e) {
            e = run(stream);
            for (bool& b : stale) b = true;
            if (!e && ndef) e = flush_deferred();
        }
        if (e != 0) {
            char tmp[400];
            strncpy(tmp, g_err, sizeof(tmp) - 1); tmp[sizeof(tmp) - 1] = 0;
            mh_set_error("plan op %d (kind %d): %s", k, ops[k].kind, tmp);
            ndef = 0;
            join();         // never leave a capture with an unjoined side stream
            return e;
        }
        k += m - 1;
    }
    if (ndef) { if (int e = flush_deferred()) { join(); return e; } }
    return join();
}

extern "C" int mh_plan_run(const mh_op* ops, int32_t nops, void* stream) { return plan_run_impl(ops, nops, stream, nullptr); }

// Several INDEPENDENT plans (private-model streams of one GPU: SURVEY 8(e) "several streams per GPU") as parallel branches: plan 0 runs on the
// caller's stream with the thread's own side lanes, plan i > 0 on a library-owned branch stream (forked from the caller's stream here, joined
// before the call returns) with a lane set of its own -- under mh_graph_begin / end the S step chains become S concurrent branches of ONE
// graph, which is what lets the latency-bound chains of small kernels share the chip (separate graph launches on separate streams do not
// overlap on this runtime: profiles/r02_experiments.txt #2).  Plans must keep to lanes 0 .. MH_MAX_LANES - 2.
namespace {
struct ThreadBranches {
    std::vector<Lanes*> sets[16];
    ~ThreadBranches() {
        LanePool& P = lane_pool();
        std::lock_guard<std::mutex> g(P.m);
        for (int d = 0; d < 16; ++d) { for (Lanes* L : sets[d]) P.free_[d].push_back(L); sets[d].clear(); }
    }
};
thread_local ThreadBranches t_branches;
int branch_get(int idx, Lanes** out) {
    int dev = 0;
    MH_HIP(hipGetDevice(&dev));
    MH_REQUIRE(dev >= 0 && dev < 16, MH_ERR_UNSUPPORTED, "device index %d out of range", dev);
    auto& v = t_branches.sets[dev];
    while ((int)v.size() <= idx) {
        LanePool& P = lane_pool();
        Lanes* L = nullptr;
        {
            std::lock_guard<std::mutex> g(P.m);
            if (!P.free_[dev].empty()) { L = P.free_[dev].back(); P.free_[dev].pop_back(); }
        }
        if (!L) {
            L = new Lanes;
            for (int k = 1; k < MH_MAX_LANES; ++k) MH_HIP(hipStreamCreateWithFlags(&L->aux[k], hipStreamNonBlocking));
            for (auto& e : L->ev) MH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            L->ready = true;
        }
        v.push_back(L);
    }
    *out = v[idx];
    return 0;
}
}  // namespace
extern "C" int mh_plans_prepare(int32_t nplans) {       // creates the branch streams / events (never inside a capture)
    MH_REQUIRE(nplans >= 1 && nplans <= 16, MH_ERR_ARG, "mh_plans_prepare: 1 .. 16 plans");
    for (int i = 1; i < nplans; ++i) { Lanes* L = nullptr; if (int e = branch_get(i - 1, &L)) return e; }
    return mh_lanes_init();
}
extern "C" int mh_plans_run(const mh_plan_ref* plans, int32_t nplans, void* stream) {
    MH_REQUIRE(plans && nplans >= 1 && nplans <= 16, MH_ERR_ARG, "mh_plans_run: 1 .. 16 plans");
    for (int i = 0; i < nplans; ++i) {
        MH_REQUIRE(plans[i].ops || plans[i].nops == 0, MH_ERR_ARG, "mh_plans_run: null plan %d", i);
        for (int k = 0; k < plans[i].nops; ++k)
            MH_REQUIRE((plans[i].ops[k].i[26] & 0xff) < MH_MAX_LANES - 1, MH_ERR_UNSUPPORTED, "mh_plans_run: plan %d uses lane %d (the last lane is the branch stream)", i,
                       plans[i].ops[k].i[26] & 0xff);
    }
    hipStream_t s0 = (hipStream_t)stream;
    Lanes* sets[16] = {};
    for (int i = 1; i < nplans; ++i) {
        if (int e = branch_get(i - 1, &sets[i])) return e;
        if (int e = lane_edge(*sets[i], s0, sets[i]->aux[MH_MAX_LANES - 1])) return e;          // fork
    }
    int rc = 0;
    for (int i = 0; i < nplans && !rc; ++i)
        rc = plan_run_impl(plans[i].ops, plans[i].nops, i == 0 ? (void*)s0 : (void*)sets[i]->aux[MH_MAX_LANES - 1], i == 0 ? nullptr : sets[i]);
    for (int i = 1; i < nplans; ++i)                                                          // join (also after an error: never leave a capture forked)
        if (int e = lane_edge(*sets[i], sets[i]->aux[MH_MAX_LANES - 1], s0)) { if (!rc) rc = e; }
    return rc;
}


extern "C" int mh_graph_begin(void* stream) {
    { Lanes* L = nullptr; if (int e = lanes_get(&L)) return e; }      // this thread's side streams exist BEFORE the capture starts
    MH_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
    return 0;
}
extern "C" int mh_graph_end(void* stream, void** graph_exec_out) {
    MH_REQUIRE(graph_exec_out, MH_ERR_ARG, "mh_graph_end: null output");
    hipGraph_t g = nullptr;
    MH_HIP(hipStreamEndCapture((hipStream_t)stream, &g));
    hipGraphExec_t ge = nullptr;
    hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) { mh_set_error("hipGraphInstantiate: %s", hipGetErrorString(e)); return (int)e; }
    *graph_exec_out = (void*)ge;
    return 0;
}
extern "C" int mh_graph_launch(void* graph_exec, void* stream) {
    MH_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
    return 0;
}
extern "C" int mh_graph_destroy(void* graph_exec) {
    if (graph_exec) MH_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
    return 0;
}
extern "C" int mh_event_create(void** ev) {
    MH_REQUIRE(ev, MH_ERR_ARG, "mh_event_create: null output");
    hipEvent_t e; MH_HIP(hipEventCreate(&e)); *ev = (void*)e; return 0;
}
extern "C" int mh_event_record(void* ev, void* stream) { MH_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream)); return 0; }
extern "C" int mh_event_elapsed_ms(void* a, void* b, float* ms) {
    MH_REQUIRE(ms, MH_ERR_ARG, "mh_event_elapsed_ms: null output");
    MH_HIP(hipEventSynchronize((hipEvent_t)b));
    MH_HIP(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
    return 0;
}
extern "C" int mh_event_destroy(void* ev) { if (ev) MH_HIP(hipEventDestroy((hipEvent_t)ev)); return 0; }
extern "C" int mh_stream_sync(void* stream) { MH_HIP(hipStreamSynchronize((hipStream_t)stream)); return 0; }
