/* madnet_hip.h -- C-ABI of libmadnet_hip.so (MI355X / gfx950, hand-written HIP).
 *
 * Drop-in boundary for the MADNet/DispNet forward + online-adaptation backward hot path
 * of CVLAB-Unibo/Real-time-self-adaptive-deep-stereo.  The reference's
 * native boundary is a TensorFlow custom op pair with C++ launchers
 *     void ShiftCorrKernelLauncher(const float*, const float*, int max_disp, int batch,
 *                                  int in_h, int in_w_padded, int channels, float* out)
 *                                              (Nets/Native/shift_corr.cc:22-23, .cu.cc:193)
 *     void ShiftCorrGradKernelLauncher(...)    (Nets/Native/shift_corr.cc:58-60, .cu.cc:235)
 * and everything else on the path is a TF-1.12 library kernel (tf.nn.conv2d, ...).  This
 * header declares the extern "C" entry points that replace BOTH: mh_corr_* replace the
 * ShiftCorr launchers (un-padded NHWC in, NHWC out, explicit stream, int status), the
 * rest replace the TF kernels the reference graph calls (call sites cited per function).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer owned by the caller
 *     (the library never allocates or frees tensor memory; borrowed until the stream op ends)
 *   - all tensors float32 NHWC; `*_ld` = element stride between consecutive pixels
 *     (lets a tensor be a channel-slice of a wider, concatenated buffer)
 *   - `stream` is a hipStream_t passed as void*; kernels are stateless and re-entrant
 *   - return 0 on success; negative = argument check failed; positive = hipError_t.
 *     mh_last_error() returns a thread-local message.  No exceptions cross the ABI.
 */
#ifndef MADNET_HIP_H
#define MADNET_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MH_ABI_VERSION 16

/* status codes of every int-returning entry point: 0 = launched; a NEGATIVE code means an argument check failed and
 * nothing was launched or written; a POSITIVE value is the hipError_t of a failed launch / runtime call.
 * (The reference launchers return void and never check cudaGetLastError, shift_corr.cu.cc:226-232.) */
#define MH_OK 0
#define MH_ERR_ARG (-1)          /* null pointer, non-positive size, ld smaller than the channel count, ... */
#define MH_ERR_ALIGN (-2)        /* a 16-byte alignment / multiple-of-4 requirement of the entry point is violated */
#define MH_ERR_COLLECTIVE (-4)   /* RCCL returned an error (mh_last_error carries its code and text) */
#define MH_ERR_UNSUPPORTED (-3)  /* valid arguments outside what the kernels implement (sizes >= 2 GiB, odd mode combinations) */

const char* mh_last_error(void);
/* name + template arguments + launch shape of the kernel instance the LAST conv / filter-gradient / correlation entry point of
 * this thread dispatched (thread-local, "" before the first call) -- lets a benchmark report the kernel it actually timed */
const char* mh_last_kernel(void);
int mh_abi_version(void);
/* number of visible HIP devices (<=0: none) -- lets the host fail loudly without torch */
int mh_device_count(void);
/* one-time set-up (opts every kernel instantiation into >64 KiB dynamic LDS); call once per
 * process before the first launch and never inside a hipGraph capture */
int mh_init(void);

/* ---- convolution family: tf.nn.conv2d / atrous_conv2d / conv2d_transpose + bias_add +
 *      leaky (Nets/sharedLayers.py:54-92), and their registered gradients -------------- */
typedef struct mh_conv_desc {
    int32_t B, Hi, Wi, Ho, Wo;      /* Hi/Wi: spatial size of `in`; Ho/Wo: of `out`              */
    int32_t K;                      /* channels contracted per tap (GEMM-K per tap)              */
    int32_t N;                      /* channels produced (GEMM-N)                                */
    int32_t kh, kw, stride, dil;    /* of the FORWARD convolution this call belongs to           */
    int32_t pad_t, pad_l;           /* TF 'SAME' pad_before of the forward convolution            */
    int32_t mode;                   /* 0: out[oy]<-in[oy*stride+ky*dil-pad]  (forward conv)
                                       1: out[y] <-in[(y+pad-ky*dil)/stride] (dgrad / conv2d_transpose) */
    int32_t w_trans;                /* 0: w is [tap][K][N]   1: w is [tap][N][K] (dgrad reads HWIO transposed) */
    int32_t in_ld, out_ld, mask_ld;
    int32_t accumulate;             /* out = result + out                                        */
    float alpha;                    /* leaky slope applied to (acc+bias); 1 = linear             */
    float mask_alpha;               /* if mask_ref: out *= (mask_ref>0 ? 1 : mask_alpha)  (fused leaky-grad) */
    int32_t mask_c0, mask_c1;       /* the mask applies to output channels [mask_c0, mask_c1); 0,0 = all
                                       (a concat gradient masks only the slice produced by an activation) */
    int32_t precision;              /* 0: exact fp32 (v_mfma_f32_16x16x4_f32) -- the parity path
                                       1: bf16 MFMA inputs (RNE) with fp32 accumulation -- throughput mode
                                       2: split-bf16: each fp32 operand as hi + lo bf16, three bf16 MFMAs per product, fp32
                                          accumulation (~2^-16 relative per product): forward layers with an x3 kernel instance
                                          (stride-1 3x3, >= 48 output channels, large images); every other call runs code 0.
                                          Filter gradients treat 2 as 0.
                                          Tensors stay fp32 in memory in every mode */
} mh_conv_desc;

int mh_conv2d(const mh_conv_desc* d, const float* in, const float* w, const float* bias,
              float* out, const float* mask_ref, void* stream);

/* mh_conv2d with the filter bank ALSO given as an "MFMA fragment bank" (forward, stride-1 3x3 layers in split-bf16 mode, precision 2):
 * wb = the image mh_pack_weights writes -- bank[(tap * ceil(K/32) + chunk)][16-column tile][plane hi, lo][lane 0..63][8 bf16], lane l
 * holding w[tap][32*chunk + 8*(l>>4) .. +7][16*tile + (l&15)], zero padded -- mh_pack_bytes(9, K, N, 2) bytes, 16-byte aligned.  The
 * kernel then streams its weight operand straight from global memory into registers (no LDS staging, no barrier in the K walk).
 * Also taken by the small-layer bank kernel (<= 4096 output pixels, K <= 224: 16 waves split the reduction of one 32x32 tile), which
 * additionally runs precision 1 (bank with planes = 1) and the input gradient (mode 1: bank packed with trans = 1, planes = 1).
 * Layers the bank kernels do not cover, and wb = NULL, behave exactly like mh_conv2d.  The bank must be re-packed whenever w changes
 * (the engines do it once per step, one launch for every layer).  Same arithmetic as the LDS-staged split-bf16 kernel. */
int mh_conv2d_wb(const mh_conv_desc* d, const float* in, const float* w, const void* wb, const float* bias,
                 float* out, const float* mask_ref, void* stream);
/* mh_conv2d_wb that ALSO writes out_shadow = bf16(out) as [pixel][N rounded up to 32] (round to nearest even; the padding channels are not
 * touched: allocate the shadow zeroed) -- the operand layout of mh_wgrad_stream, produced where the value is computed instead of by a
 * cast pass (+2 bytes per element written, nothing re-read).  wb and out_shadow may be NULL (then exactly mh_conv2d_wb / mh_conv2d). */
int mh_conv2d_sh(const mh_conv_desc* d, const float* in, const float* w, const void* wb, const float* bias,
                 float* out, const float* mask_ref, void* out_shadow, void* stream);
typedef struct mh_pack_seg {
    const float* src;     /* HWIO bank [taps][K][N] */
    void* dst;            /* fragment bank, mh_pack_bytes(taps, K, N, planes) bytes */
    int32_t taps, K, N;
    int32_t planes;       /* 2: hi + lo (split-bf16); 1: hi only */
    int32_t blk0;         /* exclusive prefix sum of ceil(taps*ceil(K/32)*ceil(N/16)*64 / 256) over the table */
    int32_t trans;        /* 0: src[tap][K][N] (forward: K = Cin, N = Cout); 1: src[tap][N][K] (the same HWIO bank seen by the input
                             gradient: K = Cout is the reduction, N = Cin the output column); 2: forward bank in the 32x32x16 register image
                             mh_conv2d_planes reads (planes = 2, mh_pack32_bytes bytes; blk0 then counts ceil(taps*ceil(K/16)*ceil(N/32)*64 / 256));
                             3: the input gradient's bank in that image (mh_conv2d_planes_bwd): planes = 1, K = Cout (reduction), N = Cin as for trans 1,
                             taps mirrored, mh_pack32_bytes(taps, K, N) / 2 bytes.  trans 2 with planes = 1: the one-plane (plain bf16) forward
                             bank of mh_conv2d_planes(precision 1), mh_pack32_bytes / 2 bytes */
    int32_t kc16;         /* trans 2 / 3: K-chunk of the image in 16-channel steps = mh_planes_kc16(K) (0: whole-K image; else the bank is
                             [chunk][tap][step] and K is padded to whole chunks -- blk0 counts the padded steps) */
    int32_t reserved;
} mh_pack_seg;
int64_t mh_pack_bytes(int32_t taps, int32_t K, int32_t N, int32_t planes);
int64_t mh_pack32_bytes(int32_t taps, int32_t K, int32_t N);      /* two-plane image, K padded per mh_planes_kc16 */
/* layout rule of the 32x32x16 images: 0 = whole-K (reductions up to 128 channels, and 97..112 / 193..208: DispNet's split-bf16 iconv layers),
 * else the chunk in 16-channel steps (4: every longer reduction -- DispNet's 256 .. 1056-channel layers and their input gradients) */
int mh_planes_kc16(int32_t K);
/* segs_device: table in DEVICE memory; nblocks = the sum the blk0 fields prefix. */
int mh_pack_weights(const mh_pack_seg* segs_device, int32_t nseg, int32_t nblocks, void* stream);

/* ---- split-bf16 forward pass from PRE-SPLIT operands (round 4; csrc/conv_planes.hip): tf.nn.conv2d / atrous_conv2d + bias_add + leaky
 *      (Nets/sharedLayers.py:54-77) of the stride-1 3x3 layers, activations carried as TWO bf16 NHWC planes hi = bf16(x), lo = bf16(x - hi)
 *      (pixel stride `*_pld` halfs >= the channel count rounded up to 16, a multiple of 8; the padding channels must be ZERO) -- the hi plane is
 *      the bf16 shadow mh_wgrad_stream and the input gradients read.  Same products as precision code 2 of mh_conv2d (lo*hi + hi*lo + hi*hi).
 *   d         : the forward descriptor (mode 0, stride 1, 'SAME', kh = kw = 3, any dilation; in_ld / precision / accumulate are ignored / must be 0)
 *   wb32      : mh_pack_weights(trans = 2) image of the HWIO bank, mh_pack32_bytes(9, K, N) bytes
 *   out       : fp32 result [pixel][d->out_ld] or NULL (only where a consumer without a plane path exists)
 *   out_hi/lo : result planes [pixel][out_pld] or NULL
 *   d->precision 1: plain bf16 -- ONE MFMA per product from the hi plane and a one-plane bank (in_lo ignored, may be NULL): DispNet's 'mixed' forward
 *   layers (Nets/DispNet.py:75-152).  Reductions over more than 128 channels run the K-chunked kernel (conv_planes_ck_kernel: a loader wave
 *   streams 64-channel patch chunks through three LDS buffers by LDS DMA while four waves walk the previous chunk).
 * mh_conv2d_planes_ok(d) = 1 if the layer has an instance (N a multiple of 8; MADNet: ceil(K/16) in {2,3,4,5,6,8}, N <= 128; DispNet: any K > 128,
 *   and 97 -> 32 / 193 -> 64 split-bf16).
 * mh_plane_split: fp32 NHWC -> hi (+ lo) planes for tensors no plane-writing kernel produces (one launch per table; lo may be NULL = mh_shadow_cast). */
int mh_conv2d_planes_ok(const mh_conv_desc* d);
int mh_conv2d_planes(const mh_conv_desc* d, const void* in_hi, const void* in_lo, int32_t in_pld, const void* wb32, const float* bias,
                     float* out, void* out_hi, void* out_lo, int32_t out_pld, void* stream);
/* Input gradient of such a layer from bf16 shadows, the same kernel with ONE plane (plain bf16, fp32 accumulate -- the arithmetic of precision code 1):
 *   dx = conv2d_backprop_input(dz, w) [* (mask > 0 ? 1 : d->mask_alpha)]        (Conv2DBackpropInput + the gradient of tf.maximum(alpha x, x))
 *   d      : the FORWARD descriptor of the layer (K = Cin, N = Cout, dil; in_ld = pixel stride of dx in floats)
 *   dz_hi  : bf16 plane of d loss / d output [pixel][dz_pld], padding channels zero;  wb32t: mh_pack_weights(trans = 3) image of the HWIO bank
 *   mask_hi: NULL, or the bf16 (hi) plane of the layer's INPUT activation [pixel][mask_pld] (only its sign is read)
 *   dx     : fp32 result or NULL;  dx_hi: its bf16 plane [pixel][dx_pld] or NULL (the next input gradient's dz_hi, the filter gradient's operand)
 * d->mask_c0 / mask_c1: the mask applies to output channels [c0, c1) only (one member of a concat: DispNet's up-sampling blocks); 0, 0 = all.
 * dx rows (d->in_ld) and dx_pld must hold Cin rounded up to 8 (the epilogue stores 8 channels per lane; the padding columns receive zeros).
 * Reductions over more than 128 output channels run the K-chunked kernel (bank packed with kc16 = mh_planes_kc16(Cout)).
 * STRIDE-2 layers (d->stride = 2, 3x3, 'SAME' on even sizes i.e. pad_t = pad_l = 0, Hi = 2 Ho, Wi = 2 Wo; Cout 32 or 64, Cin <= 32): the parity-class
 * kernel -- the four parities of (y, x) are four small convolutions over one dz patch; same bank (trans = 3), same operands, dz_hi is [B][Ho][Wo].
 * No accumulation: launches that need it stay on mh_conv2d (mode 1). */
int mh_conv2d_planes_bwd_ok(const mh_conv_desc* d);
int mh_conv2d_planes_bwd(const mh_conv_desc* d, const void* dz_hi, int32_t dz_pld, const void* wb32t, const void* mask_hi, int32_t mask_pld,
                         float* dx, void* dx_hi, int32_t dx_pld, void* stream);
typedef struct mh_plane_seg {
    const float* src;     /* fp32 [npix][src_ld], C valid channels */
    void* hi;             /* bf16 [npix][dst_ld]: bf16(src), channels >= C zero; 16-byte aligned */
    void* lo;             /* bf16 [npix][dst_ld]: bf16(src - hi), or NULL */
    int64_t npix;
    int32_t C, src_ld, dst_ld;
    int32_t blk0;         /* exclusive prefix sum of ceil(npix * dst_ld / 8 / 256) over the table */
    const float* src2;    /* NULL, or a second source whose C2 channels follow src's C (a fused tf.concat: [features | disparity], MadNet.py:123) */
    int32_t C2, src2_ld;
} mh_plane_seg;
int mh_plane_split(const mh_plane_seg* segs_device, int32_t nseg, int32_t nblocks, void* stream);

/* weight+bias gradient: dw[tap][K][N] += sum_pixels in(pixel,tap)[k] * dout[pixel][n] ;
 * db[n] += sum_pixels dout[pixel][n].  `d` describes the FORWARD conv (mode 0 geometry:
 * B,Hi,Wi = input, Ho,Wo = output, K = Cin, N = Cout).  dw/db are ACCUMULATED (fp32
 * atomics over pixel splits) -- zero them first.  db may be NULL. */
int mh_conv2d_wgrad(const mh_conv_desc* d, const float* in, const float* dout, int32_t dout_ld,
                    float* dw, float* db, void* stream);

/* Atomic-free form for a whole training step (the per-layer conv2d_backprop_filter calls TF issues for
 * Stereo_Online_Adaptation.py:114's minimize()): pixel split s writes its partial filter gradient to
 * ws[s][kh*kw*K*N] with plain stores; mh_wgrad_reduce then sums the splits of EVERY layer in one launch.
 *   query : ws == NULL -> *splits = number of pixel splits this geometry uses (nothing is launched);
 *   launch: ws != NULL, *splits = the queried value; ws must hold *splits * (kh*kw*K*N + (db ? N : 0)) floats, 16-byte
 *           aligned; it is fully overwritten (no zeroing needed).  Bias gradient (ABI 15): with *splits > 1 the bias partial sums of
 *           split s are STORED at ws[*splits * kh*kw*K*N + s * N + n] and db is not touched -- sum them with one more mh_wgrad_seg
 *           {ws + *splits * kh*kw*K*N, db, N, *splits} in the same mh_wgrad_reduce launch (fixed summation order: two replays of a step
 *           give bit-identical bias gradients; ABI <= 14 added one float atomic per workgroup and channel to db).  With *splits == 1 a
 *           channel has a single addend, which is atomically added to db as before (zero db first). */
int mh_conv2d_wgrad_partial(const mh_conv_desc* d, const float* in, const float* dout, int32_t dout_ld,
                            float* ws, int32_t* splits, float* db, void* stream);
/* Several layers' partial filter gradients in one launch (same contract per item as mh_conv2d_wgrad_partial with ws != NULL: `splits` from a
 * query call, the bias partial sums behind the filter partials).  bf16-mode layers share one grid; exact-fp32 layers are launched one by one.  Replaces the per-layer Conv2DBackpropFilter nodes
 * TF schedules for one tf.gradients() call (Stereo_Online_Adaptation.py:126-128). */
typedef struct mh_wgrad_item {
    mh_conv_desc d;
    const float* in; const float* dout;
    float* ws; float* db;
    int32_t dout_ld, splits;
    int32_t group_max_m;  /* > 0: group this layer only if it has at most this many reduction pixels (0 = library default 4096);
                             the largest value of a batch applies to the whole batch */
    int32_t reserved;
} mh_wgrad_item;
int mh_conv2d_wgrad_partial_group(const mh_wgrad_item* items, int32_t n, void* stream);
typedef struct mh_wgrad_seg {
    const float* ws;      /* [splits][size] partial sums of one layer */
    float* dst;           /* [size] filter gradient */
    int32_t size, splits;
    int32_t blk0;         /* exclusive prefix sum of ceil(size/1024) over the table */
    int32_t accumulate;   /* 0: dst = sum, 1: dst += sum (dst must then appear once per table) */
} mh_wgrad_seg;
/* segs_device: table in DEVICE memory; nblocks = sum of ceil(size/1024). */
int mh_wgrad_reduce(const mh_wgrad_seg* segs_device, int32_t nseg, int32_t nblocks, void* stream);

/* ---- filter gradients of the stride-1 3x3 (dilated) layers from bf16 "shadows", a whole backward batch per launch (csrc/wgrad_stream.hip) ----
 * The same Conv2DBackpropFilter / BiasAddGrad nodes as mh_conv2d_wgrad_partial (Stereo_Online_Adaptation.py:126-128), bf16 operands / fp32
 * accumulation, but the operands are read from bf16 NHWC copies ("shadows") of the layer input x and of d(loss)/d(output) dz whose channel
 * stride is the channel count rounded up to a multiple of 32 with the padding channels ZERO.
 *   mh_shadow_cast        : writes such shadows for a table of fp32 tensors (one launch; round to nearest even).
 *   mh_wgrad_stream_plan  : HOST-side planner.  Fills ktiles / ntiles / splits / blk0 of every layer record so that about `target_wgs` workgroups
 *                           of `nwaves` waves share the batch in proportion to the rows they stream; *nblocks_out = the grid.  The caller then
 *                           points layer.ws at splits * 9*K*N floats (16-byte aligned; or at dw itself when splits == 1) and uploads the table.
 *   mh_wgrad_stream       : the launch.  ws[split][9][K][N] is fully overwritten (sum the splits with mh_wgrad_reduce).  Bias gradient (db != NULL; ABI 15):
 *                           a layer with splits > 1 needs splits * N more floats behind its filter partials -- the bias partial sums are stored at
 *                           ws[splits * 9*K*N + split * N + n], db is not touched, one more reduction segment {.., db, N, splits} sums them in split order;
 *                           a layer with ONE split adds its single addend per channel to db atomically (zero db first).  max_dil = the largest dilation in the table (1 .. 16), -2 for a table of stride-2 layers, -3 for a table that mixes
 *                           stride-1 (dilation <= 8) and stride-2 layers. */
typedef struct mh_shadow_seg {
    const float* src;     /* fp32 [npix][src_ld], C valid channels */
    void* dst;            /* bf16 [npix][dst_ld], dst_ld % 8 == 0, channels >= C zero-filled; 16-byte aligned */
    int64_t npix;
    int32_t C, src_ld, dst_ld;
    int32_t blk0;         /* exclusive prefix sum of ceil(npix * dst_ld / 8 / 256) over the table */
} mh_shadow_seg;
int mh_shadow_cast(const mh_shadow_seg* segs_device, int32_t nseg, int32_t nblocks, void* stream);
typedef struct mh_wgs_layer {
    const void* x;        /* bf16 shadow of the layer input   [B][H][W][x_ld]  */
    const void* dz;       /* bf16 shadow of d loss / d output [B][H][W][dz_ld] */
    float* ws;            /* [splits][9][K][N] partial filter gradients */
    float* db;            /* [N] bias gradient (accumulated) or NULL */
    int32_t B, H, W, K, N, dil;
    int32_t x_ld, dz_ld;  /* >= K, N rounded up to 32; multiples of 8 */
    int32_t ktiles, ntiles, splits, blk0;      /* written by mh_wgrad_stream_plan */
    int32_t stride;       /* 1, or 2 (dilation 1, even sizes): then B, H, W are the OUTPUT (dz) size and x is [B][2H][2W][x_ld] */
    int32_t reserved;
} mh_wgs_layer;
int mh_wgrad_stream_plan(mh_wgs_layer* layers_host, int32_t n, int32_t target_wgs, int32_t nwaves, int32_t* nblocks_out);
int mh_wgrad_stream(const mh_wgs_layer* layers_device, int32_t nlayers, int32_t nblocks, int32_t nwaves, int32_t max_dil, void* stream);

/* ---- correlation / cost volume: sharedLayers.correlation (Nets/sharedLayers.py:23-51),
 *      replaces ShiftCorrKernelLauncher / ShiftCorrGradKernelLauncher ------------------- */
/* out[p][coff + j] = mean_c L[p][c]*R[p + (j*stride - max_disp)][c]   (R zero outside the row)
 * if copy_left: out[p][0..C) = L[p][:]  (fused tf.concat([reference, corr]), MadNet.py:370-375)
 * if u:         out[p][coff + D] = u[p] (fused tf.concat([costs, upsampled_disp]), MadNet.py:77-80)
 * channels [coff + D (+1), out_ld) are zero-filled when zero_tail != 0. */
int mh_corr_fwd(const float* L, int32_t l_ld, const float* R, int32_t r_ld, const float* u,
                float* out, int32_t out_ld, int32_t coff,
                int32_t B, int32_t H, int32_t W, int32_t C, int32_t max_disp, int32_t stride,
                int32_t copy_left, int32_t zero_tail, void* stream);
/* mh_corr_fwd with an arithmetic mode for the LARGE search range (D > 9, DispNet's 81-shift volume, stand-alone form): 0 = exact
 * fp32 MFMA (what mh_corr_fwd runs), 1 = bf16 operands / fp32 accumulate, 2 = split-bf16 (3 MFMAs per product, ~2^-16 relative).
 * D <= 9 and the fused-concat forms ignore the mode (pure bandwidth kernels, exact fp32). */
int mh_corr_fwd_prec(const float* L, int32_t l_ld, const float* R, int32_t r_ld, const float* u,
                     float* out, int32_t out_ld, int32_t coff,
                     int32_t B, int32_t H, int32_t W, int32_t C, int32_t max_disp, int32_t stride,
                     int32_t copy_left, int32_t zero_tail, int32_t precision, void* stream);
/* Fused front end of one MADNet pyramid level (Nets/MadNet.py:274-295 and the same lines of the other levels): in ONE launch
 *   u[p]        = mul * resize_bilinear(Vc[B,Hc,Wc] -> [H,W])[p]                (tf.image.resize_images, TF1 legacy; MadNet.py:274)
 *   Rw[p][c]    = linear warp of R at x + u[p] along the row, zero outside      (_build_indeces + _linear_warping, :378-436)
 *   out[p]      = [ L[p][0..C) | mean_c L[p][c]*Rw[p + j - max_disp][c], j < D | u[p] | 0 ... ]   (the mh_corr_fwd concat layout)
 * i.e. mh_resize_fwd(mode 0) + mh_warp_fwd + mh_corr_fwd(copy_left, u) with stride 1, D = 2*max_disp+1 <= 9.  Rw and u are
 * outputs too (the backward pass reads them). */
int mh_level_front_fwd(const float* Vc, int32_t Hc, int32_t Wc, float mul, const float* L, int32_t l_ld, const float* R, int32_t r_ld,
                       float* out, int32_t out_ld, int32_t coff, float* Rw, int32_t rw_ld, float* u,
                       int32_t B, int32_t H, int32_t W, int32_t C, int32_t max_disp, int32_t zero_tail, void* stream);
/* mh_level_front_fwd that ALSO writes the estimator input as bf16 planes (out_hi: the shadow the filter gradient reads; out_lo, may be NULL: the lo
 * plane mh_conv2d_planes needs), pixel stride out_pld halfs >= coff + D + 1, padding channels untouched (allocate zeroed). */
int mh_level_front_fwd_planes(const float* Vc, int32_t Hc, int32_t Wc, float mul, const float* L, int32_t l_ld, const float* R, int32_t r_ld,
                              float* out, int32_t out_ld, int32_t coff, float* Rw, int32_t rw_ld, float* u,
                              int32_t B, int32_t H, int32_t W, int32_t C, int32_t max_disp, int32_t zero_tail,
                              void* out_hi, void* out_lo, int32_t out_pld, void* stream);
/* mh_level_front_fwd_planes that ALSO runs the disparity head of the COARSER level (MadNet.py:118, the last layer of _stereo_estimator: 3x3, K -> 1 channels,
 * no activation) instead of reading its result: Vc[b][y][x] = sum_{tap,k} X[b][y + ky - 1][x + kx - 1][k] * hw[tap][k] + hb[0] (zero padding, fp32, the
 * arithmetic of mh_conv2d_fwd's single-output-channel kernel) is an OUTPUT here -- every element of Vc[B,Hc,Wc] is stored -- and u is interpolated from the
 * values as computed.  One launch less on the forward chain per level.  X: [B,Hc,Wc,x_ld] activations of the estimator's fifth layer; hw: the head's HWIO
 * bank [3][3][K][1] (16-byte aligned); hb: its bias or NULL.  Served shapes: mh_level_front_head_ok (K % 4 == 0, K <= 32, D <= 9, and an UP-scaling geometry
 * H >= Hc, W >= Wc: a coarse pixel is stored by the workgroup whose fine pixels interpolate from it) -- MH_ERR_UNSUPPORTED otherwise. */
int mh_level_front_head_fwd(const float* X, int32_t x_ld, int32_t K, const float* hw, const float* hb, float* Vc, int32_t Hc, int32_t Wc, float mul,
                            const float* L, int32_t l_ld, const float* R, int32_t r_ld, float* out, int32_t out_ld, int32_t coff, float* Rw, int32_t rw_ld,
                            float* u, int32_t B, int32_t H, int32_t W, int32_t C, int32_t max_disp, int32_t zero_tail,
                            void* out_hi, void* out_lo, int32_t out_pld, void* stream);
/* 1 when mh_level_front_head_fwd serves the shape, 0 when the head has to stay a launch of its own (no status, no error message) */
int mh_level_front_head_ok(int32_t Hc, int32_t Wc, int32_t H, int32_t W, int32_t C, int32_t K, int32_t max_disp);
/* g: gradient w.r.t. the buffer written by mh_corr_fwd (same ld / coff).
 * dL[p][c] (+)= [copy_left] g[p][c] + (1/C) sum_j g[p][coff+j] R[p+i_j][c]
 * dR[p][c] (+)=                      (1/C) sum_j g[p-i_j][coff+j] L[p-i_j][c]
 * du[p]    (+)= g[p][coff+D]     (du may be NULL).   acc_* select += vs =. */
int mh_corr_bwd(const float* g, int32_t g_ld, int32_t coff, const float* L, int32_t l_ld,
                const float* R, int32_t r_ld, float* dL, int32_t dl_ld, int32_t acc_l,
                float* dR, int32_t dr_ld, int32_t acc_r, float* du, int32_t acc_u,
                int32_t B, int32_t H, int32_t W, int32_t C, int32_t max_disp, int32_t stride,
                int32_t copy_left, void* stream);
/* The same gradient with a chosen arithmetic for the LARGE shift counts (D > 9, DispNet's 81-shift volume of Nets/DispNet.py:100-101; gradient of
 * Nets/sharedLayers.py:41-51): precision 0 = exact fp32 (mh_corr_bwd), 1 = operands (features and g) rounded to bf16, fp32 accumulation on
 * v_mfma_f32_16x16x32_bf16 -- the arithmetic of the other input gradients of the 'mixed' / 'bf16' engine modes -- 2 = runs as 0.  D <= 9 is pure
 * bandwidth and always exact fp32. */
int mh_corr_bwd_prec(const float* g, int32_t g_ld, int32_t coff, const float* L, int32_t l_ld,
                     const float* R, int32_t r_ld, float* dL, int32_t dl_ld, int32_t acc_l,
                     float* dR, int32_t dr_ld, int32_t acc_r, float* du, int32_t acc_u,
                     int32_t B, int32_t H, int32_t W, int32_t C, int32_t max_disp, int32_t stride,
                     int32_t copy_left, int32_t precision, void* stream);
/* One pyramid level's backward front end in ONE launch: mh_corr_bwd with the WARPED right features Rw as the right operand (fused cost volume +
 * concat form), followed by mh_warp_bwd of the resulting gradient -- which is never stored: dimg (+)= the bilinear scatter of it to the unwarped
 * right features' gradient (fp32 atomics: dimg must hold zeros or earlier contributions), du = g[.., coff + D] + the coordinate gradient of the
 * warp (img = the unwarped right features, u = the warp coordinates).  dimg or du may be NULL.  Replaces the gradients of
 * MadNet._linear_warping + correlation + concat of one level (MadNet.py:370-436, 77-80).
 * acc_l: bit 0 = dL accumulates; bit 1 (MH_CORR_WARP_OVERWRITE_DIMG, ABI 15) = this launch is the FIRST writer of dimg: it is overwritten, nothing of it is read
 * and it need not be zeroed (the row-owned gather kernel just stores; the scattering forms zero it themselves first). */
#define MH_CORR_WARP_OVERWRITE_DIMG 2
int mh_corr_warp_bwd(const float* g, int32_t g_ld, int32_t coff, const float* L, int32_t l_ld, const float* Rw, int32_t rw_ld,
                     const float* img, int32_t img_ld, const float* u, float* dL, int32_t dl_ld, int32_t acc_l,
                     float* dimg, int32_t dimg_ld, float* du,
                     int32_t B, int32_t H, int32_t W, int32_t C, int32_t max_disp, int32_t stride, int32_t copy_left, void* stream);

/* ---- the reference launchers under their LITERAL signature (+ an explicit stream, + a status) -- what a caller binds who
 *      keeps sharedLayers.correlation_native (Nets/sharedLayers.py:31-39) as it is: in0 / in1 NHWC with W already zero-padded by
 *      max_disp on both sides, out / grad NCHW [batch, 2*max_disp+1, in_h, in_w_padded - 2*max_disp].
 *   mh_shift_corr      replaces ShiftCorrKernelLauncher     (Nets/Native/shift_corr.cc:22-23, shift_corr.cu.cc:193-233)
 *   mh_shift_corr_grad replaces ShiftCorrGradKernelLauncher (Nets/Native/shift_corr.cc:58-60, shift_corr.cu.cc:235-288): out0 / out1
 *      = gradient w.r.t. the padded in0 / in1, laid out like them (NHWC) -- the gradient of the forward formula, not the
 *      reference kernels' defective arithmetic. */
int mh_shift_corr(const float* in0, const float* in1, int32_t max_disp, int32_t batch, int32_t in_h, int32_t in_w_padded,
                  int32_t channels, float* out, void* stream);
int mh_shift_corr_grad(const float* in0, const float* in1, const float* grad, int32_t max_disp, int32_t batch, int32_t height,
                       int32_t padded_width, int32_t channels, float* out0, float* out1, void* stream);

/* ---- MadNet._build_indeces + _linear_warping (Nets/MadNet.py:378-436) ---------------- */
int mh_warp_fwd(const float* img, int32_t img_ld, const float* u, float* out, int32_t out_ld,
                int32_t B, int32_t H, int32_t W, int32_t C, void* stream);
/* dimg += scatter (fp32 atomics; zero/initialise it first; NULL = coordinate gradient only); du (+)= coordinate gradient
 * (NULL when the coordinates are stop_gradient-ed, i.e. bulkhead / MAD, or when only the scatter is wanted). */
int mh_warp_bwd(const float* g, int32_t g_ld, const float* img, int32_t img_ld, const float* u,
                float* dimg, int32_t dimg_ld, float* du, int32_t acc_u,
                int32_t B, int32_t H, int32_t W, int32_t C, void* stream);

/* ---- tf.image.resize_images (TF1 legacy bilinear) fused with scale / relu / centre crop:
 *      MadNet._make_disp (MadNet.py:68-71), final prediction (:362-364), inter-level
 *      upsample (:274), preprocessing.rescale_image (preprocessing.py:269-277) ----------
 * virtual resize of in[B,Hi,Wi] to [Hr,Wr], then crop at (cy,cx) to out[B,Ho,Wo].
 * mode 0: out = mul*resize(in)   1: out = resize(relu(mul*in))   2: out = relu(mul*resize(in)) */
int mh_resize_fwd(const float* in, float* out, int32_t B, int32_t Hi, int32_t Wi,
                  int32_t Hr, int32_t Wr, int32_t cy, int32_t cx, int32_t Ho, int32_t Wo,
                  float mul, int32_t mode, void* stream);
/* din (+)= gradient (deterministic gather form).  `in` is the forward input (needed for
 * the relu masks of modes 1/2). */
int mh_resize_bwd(const float* g, const float* in, float* din, int32_t accumulate,
                  int32_t B, int32_t Hi, int32_t Wi, int32_t Hr, int32_t Wr, int32_t cy, int32_t cx,
                  int32_t Ho, int32_t Wo, float mul, int32_t mode, void* stream);

/* ---- the same legacy bilinear resize for NHWC images with C interleaved channels: Stereo_Online_Adaptation.scale_tensor
 *      (Stereo_Online_Adaptation.py:22-23,91-95 -> preprocessing.rescale_image, preprocessing.py:269-273) on the frames when
 *      --reprojectionScale != 1.  mh_resize_image_bwd: din = gradient (gather form, overwrites din). */
int mh_resize_image_fwd(const float* in, float* out, int32_t B, int32_t Hi, int32_t Wi, int32_t C, int32_t Ho, int32_t Wo, void* stream);
int mh_resize_image_bwd(const float* g, float* din, int32_t B, int32_t Hi, int32_t Wi, int32_t C, int32_t Ho, int32_t Wo, void* stream);

/* ---- preprocessing.bilinear_sampler, general form (Data_utils/preprocessing.py:121-199): out[b,y,x,:] = 4-tap bilinear sample of
 *      imgs[B,Hs,Ws,C] at coords[B,Ht,Wt,2] = (x, y); indices clamped to the border, weights un-masked, gather index computed in
 *      float32 like the reference (B*Hs*Ws < 2^24).  bwd: dcoords (may be NULL) overwritten; dimgs (may be NULL) accumulated with fp32
 *      atomics -- zero it first. */
int mh_bilinear_sampler_fwd(const float* imgs, const float* coords, float* out, int32_t B, int32_t Hs, int32_t Ws, int32_t C,
                            int32_t Ht, int32_t Wt, void* stream);
int mh_bilinear_sampler_bwd(const float* g, const float* imgs, const float* coords, float* dcoords, float* dimgs, int32_t B,
                            int32_t Hs, int32_t Ws, int32_t C, int32_t Ht, int32_t Wt, void* stream);

/* ---- input side: out[i] = (float)in[i] for uint8 frames -- the tf.cast(image, tf.float32) of the reference's reader
 *      (Data_utils/data_reader.py:98) executed after the host-to-device copy (1 byte per value over PCIe) */
int mh_u8_to_f32(const uint8_t* in, float* out, int64_t n, void* stream);

/* ---- the step's frames through a table the host rewrites between two replays of a captured step.  The reference feeds every sess.run through its input
 *      pipeline (Stereo_Online_Adaptation.py:73-80: the iterator's tensors ARE the graph's inputs); a captured hipGraph reads fixed addresses, and a prefetcher
 *      delivers frame t in slot t % depth -- so the step's first node looks the slot up: for k < count, src = table->src[k]: NULL or == dst[k]: nothing; else
 *      dst[k][0 .. n[k]) = (float)src[...] (table->u8[k] != 0: 8-bit source values, the tf.cast of Data_utils/data_reader.py:98; else float32).  `table` must
 *      be readable by the device when the kernel RUNS: device memory, or page-locked host memory through its device-side address (mh_host_device_pointer) --
 *      the host then rewrites it without an API call, after the previous replay has completed.  Plan op: MH_OP_FETCH_INPUTS. */
#define MH_FETCH_MAX 4
typedef struct mh_input_table { const void* src[MH_FETCH_MAX]; int32_t u8[MH_FETCH_MAX]; } mh_input_table;
int mh_fetch_inputs(const mh_input_table* table, float* const* dst, const int64_t* n, int32_t count, void* stream);
/* device-side address of a page-locked (hipHostMalloc / torch pin_memory) host allocation */
int mh_host_device_pointer(void* host, void** device);

/* ---- preprocessing.pad_image (REFLECT, preprocessing.py:7-29) fused with the float cast
 *      and the channel padding 3 -> out_ld (extra channels zero) ------------------------ */
/* out = in / div - sub  (MADNet: div=1, sub=0; DispNet._preprocess_inputs, DispNet.py:59-73: x/255 - 100/255) */
/* mh_conv2d_sh with the bf16 shadow of the INPUT tensor as well (pixel stride = K rounded up to 32 halfs, zero padded, as an earlier
 * mh_conv2d_sh / mh_shadow_cast wrote it): the patch-staged input-gradient kernel (mode 1, bf16) then stages the shadow as it is -- half the
 * bytes, no conversion, bit-identical result; every other kernel ignores it and reads `in`.  in_shadow may be NULL. */
int mh_conv2d_sh2(const mh_conv_desc* d, const float* in, const void* in_shadow, const float* w, const void* wb, const float* bias,
                  float* out, const float* mask_ref, void* out_shadow, void* stream);
/* mh_conv2d_sh2 with two more options of the patch-staged input-gradient kernel (every other kernel ignores them and behaves like mh_conv2d_sh2):
 * mask_shadow = the bf16 shadow of mask_ref (pixel stride = N rounded up to 32): the leaky mask tests only the sign, so 8 bytes of the shadow
 * replace 16 of the fp32 activation; flags & MH_CONV_SHADOW_ONLY: the fp32 result is NOT stored, only out_shadow (legal when every consumer of
 * the result takes its shadow -- ask mh_conv2d_takes_shadows for the consuming launch; needs out_shadow, no accumulation).
 * mh_conv2d_takes_shadows(d, ...) != 0 if the launch described by d / these pointers would stage in_shadow (and honour the options), else 0. */
#define MH_CONV_SHADOW_ONLY 1
/* flags & MH_CONV_IN_F32_STALE / MH_CONV_MASK_F32_STALE: the caller did not store the fp32 `in` / `mask_ref` (its producer ran with
 * MH_CONV_SHADOW_ONLY, or wrote planes only): the call FAILS (MH_ERR_UNSUPPORTED, nothing launched) unless the dispatched kernel stages in_shadow /
 * reads mask_shadow -- a plan recorded under one dispatch can never silently read a tensor nobody wrote when it is replayed under another.
 * mh_conv2d_takes_shadows returns a bit mask: 1 = in_shadow would be staged, 2 = the mask would be read from mask_shadow. */
#define MH_CONV_IN_F32_STALE 2
#define MH_CONV_MASK_F32_STALE 4
int mh_conv2d_sh3(const mh_conv_desc* d, const float* in, const void* in_shadow, const float* w, const void* wb, const float* bias,
                  float* out, const float* mask_ref, const void* mask_shadow, void* out_shadow, int32_t flags, void* stream);
int mh_conv2d_takes_shadows(const mh_conv_desc* d, const float* in, const float* w, const void* wb, float* out, const float* mask_ref);
/* mh_conv2d_sh that writes BOTH planes of its result, out_hi = bf16(out) and out_lo = bf16(out - out_hi) ([pixel][N rounded up to 32]): the producer side
 * of mh_conv2d_planes for layers that run from fp32 operands (the exact-fp32 stride-2 pyramid layers in front of conv4 / conv6).  The tiled kernel's
 * vector epilogue stores them itself; behind any other kernel family one split launch follows. */
int mh_conv2d_sh4(const mh_conv_desc* d, const float* in, const float* w, const void* wb, const float* bias, float* out, const float* mask_ref,
                  void* out_hi, void* out_lo, void* stream);
/* Forward pass of a disparity head (mh_conv2d with N = 1, mode 0) that also stores its result at up to two more places, each with its own
 * pixel stride (floats): a channel slot of a concatenated buffer (the context network's input, Nets/MadNet.py:155-157) and / or the buffer the
 * next stage accumulates into (final = V2 + context, MadNet.py:171).  out2 / out3 may be NULL.  Saves the copy launches behind the head. */
int mh_conv2d_head(const mh_conv_desc* d, const float* in, const float* w, const float* bias, float* out,
                   float* out2, int32_t out2_ld, float* out3, int32_t out3_ld, void* stream);
/* Backward front end of a disparity head (3x3 conv Cin -> 1, the last layer of a MADNet estimator: Nets/MadNet.py:95-117) in ONE launch:
 *   dV = [kind 0] gradient of u = resize_x2(V) * mul w.r.t. V, from the finer level's coordinate gradient src0 (= mh_resize_bwd mode 0, not accumulating)
 *        [kind 1] src0[pixel * src0_ld] + src1[pixel * src1_ld]   (either may be NULL)
 *   -> dV (fp32, [B,H,W]) and, if dV_shadow != NULL, its bf16 shadow (pixel stride 32 halfs, channel 0);
 *   dx (+)= conv2d_backprop_input(dV, w) * leaky'(mask_ref)  ([B,H,W,N], pixel stride dx_ld) and, if dx_shadow != NULL, its bf16 shadow
 *   (pixel stride round_up(N, 32)).  Replaces mh_resize_bwd / mh_copy_channels + mh_conv2d(mode 1, K = 1) of the reference's per-level
 *   gradient chain. */
typedef struct mh_head_bwd_desc {
    int32_t kind;                       /* 0 | 1 */
    int32_t B, H, W, N;                 /* head size, input channels of the head conv */
    int32_t Hr, Wr, cy, cx, Ho, Wo;     /* kind 0: the resize (as mh_resize_bwd: resized size, crop origin, size of the fine map src0) */
    float mul;                          /* kind 0 */
    int32_t src0_ld, src1_ld;           /* kind 1: pixel strides of the addends (floats) */
    int32_t dx_ld, mask_ld, accumulate_dx;
    float mask_alpha;
} mh_head_bwd_desc;
int mh_head_bwd(const mh_head_bwd_desc* d, const float* src0, const float* src1, float* dV, void* dV_shadow, const float* w,
                float* dx, const float* mask_ref, void* dx_shadow, void* stream);
int mh_pad_reflect(const float* in, float* out, int32_t B, int32_t H, int32_t W, int32_t C,
                   int32_t Hp, int32_t Wp, int32_t pad_t, int32_t pad_l, int32_t out_ld,
                   float div, float sub, void* stream);
/* The image layer's forward pass straight from the frames: out = leaky(conv3x3(reflect_pad(frames / div - sub), w) + bias), i.e. mh_pad_reflect followed by
 * mh_conv2d_fwd on the padded frames (Stereo_net._preprocess_inputs + the first conv2d of MadNet._pyramid_features, MadNet.py:56-60) WITHOUT reading or writing the
 * padded copy: frames [NB,H0,W0,C] fp32, padded frame Hp x Wp with the image at (reflect_t, reflect_l), SAME padding offsets pad_t / pad_l of the strided conv on
 * the padded frame (zeros outside it), w HWIO [3][3][C][N], out [NB,ceil(Hp/stride),ceil(Wp/stride),out_ld]; shadow (may be NULL): the bf16 copy of out the
 * streamed filter gradient of the next layer reads, pixel stride shadow_ld halfs.  Exact fp32 arithmetic.  Served: C = 3, N = 16, stride 1 / 2
 * (mh_conv_image_ok; MH_ERR_UNSUPPORTED otherwise). */
int mh_conv_image_fwd(const float* frames, int32_t NB, int32_t H0, int32_t W0, int32_t C, int32_t Hp, int32_t Wp, int32_t reflect_t, int32_t reflect_l,
                      float div, float sub, const float* w, const float* bias, int32_t N, int32_t stride, int32_t pad_t, int32_t pad_l, float alpha,
                      float* out, int32_t out_ld, void* shadow, int32_t shadow_ld, void* stream);
int mh_conv_image_ok(int32_t C, int32_t N, int32_t kh, int32_t kw, int32_t stride);       /* 1 / 0, no status */

/* ---- loss_factory.get_reprojection_loss('mean_SSIM_l1') forward + gradient w.r.t. the
 *      disparity (Losses/loss_factory.py:128-164,353-395; preprocessing.py:121-230) ------
 * left,right: [B,H,W,3] in 0..255;  disp: [B,H,W].  ws: workspace of mh_loss_ws_floats()
 * floats.  result[0] = loss, result[1] = mean SSIM term, result[2] = mean L1 term.
 * ddisp (may be NULL: forward only) = grad_scale * dLoss/ddisp. */
int64_t mh_loss_ws_floats(int32_t B, int32_t H, int32_t W);
int mh_reprojection_loss(const float* left, const float* right, const float* disp,
                         float* ws, float* result, float* ddisp, float grad_scale,
                         int32_t B, int32_t H, int32_t W, void* stream);

/* ---- validation ops (Stereo_Online_Adaptation.py:74-82): result[0]=EPE, result[1]=bad3,
 *      result[2]=#valid.  ws: >= mh_metrics_ws_floats() floats. ---------------------------- */
/* the same in two phases, so that the reduction of the loss VALUE (which no gradient needs) can leave the critical path of a step:
 * phase 1 = warp + SSIM maps + gradient (partial sums stay in ws), phase 2 = the final reduction into result; 0 = both. */
int mh_reprojection_loss_phase(const float* left, const float* right, const float* disp, float* ws, float* result, float* ddisp,
                               float grad_scale, int32_t B, int32_t H, int32_t W, int32_t phase, void* stream);
int64_t mh_metrics_ws_floats(int32_t B, int32_t H, int32_t W);
int mh_metrics(const float* disp, const float* gt, float* ws, float* result, float pixel_th,
               int32_t B, int32_t H, int32_t W, void* stream);

/* ---- proxy-label loss of the continual-adaptation variant: loss_factory.get_proxy_loss('mean_l1', weights=[w]*10)
 *      (Losses/loss_factory.py:304-351, mean_l1 :28-38; Stereo_Continual_Adaptation.py:75,112)
 * valid = !(proxy <= 0 || proxy >= 192);  result[0] = weight * sum(valid*|pred-proxy|) / sum(valid);  result[1] = sum(valid);
 * dpred (may be NULL) = grad_scale * weight * valid * sign(pred - proxy) / sum(valid).  ws: mh_proxy_ws_floats() floats. */
int64_t mh_proxy_ws_floats(int32_t B, int32_t H, int32_t W);
int mh_proxy_loss(const float* pred, const float* proxy, float* ws, float* result, float* dpred, float weight,
                  float grad_scale, int32_t B, int32_t H, int32_t W, void* stream);

/* ---- offline training (Train.py): supervised mean_l1 per predicted scale and the Adam update -------------------------------
 * mh_supervised_loss: loss_factory.get_supervised_loss('mean_l1', multiScale=True, max_disp=MAX_DISP) term of ONE scale
 *      (Losses/loss_factory.py:256-302, Train.py:100): valid = !(target == 0 || target >= max_disp); same outputs / workspace as
 *      mh_proxy_loss.
 * mh_adam: tf.train.AdamOptimizer(lr, beta1).apply (Train.py:95): state[0] / state[1] = beta1_power / beta2_power (initialised to
 *      beta1 / beta2 by the caller); lr_t = lr * sqrt(1 - state[1]) / (1 - state[0]); m, v, var updated in place.
 * mh_adam_advance: state[0] *= beta1, state[1] *= beta2 -- once per step, after the last mh_adam of the step. */
int mh_supervised_loss(const float* pred, const float* target, float* ws, float* result, float* dpred, float weight,
                       float grad_scale, float max_disp, int32_t B, int32_t H, int32_t W, void* stream);
int mh_adam(float* var, float* m, float* v, const float* grad, int64_t n, const float* state, float lr, float beta1, float beta2,
            float eps, float grad_scale, void* stream);
int mh_adam_advance(float* state, float beta1, float beta2, void* stream);

/* ---- tf.train.MomentumOptimizer apply (Stereo_Online_Adaptation.py:85):
 *      accum = momentum*accum + grad_scale*g ;  var -= lr*accum   (n contiguous floats) --- */
int mh_momentum(float* var, float* accum, const float* grad, int64_t n, float lr, float momentum,
                float grad_scale, void* stream);

/* ---- small glue ------------------------------------------------------------------------ */
/* dst[p][c] (+)= scale * src[p][c], c < nch  (tf.concat pieces, final_disp add, concat-grad splits) */
int mh_copy_channels(const float* src, int32_t src_ld, float* dst, int32_t dst_ld, int64_t npix,
                     int32_t nch, float scale, int32_t accumulate, void* stream);
/* dy[p][c] *= (y[p][c] > 0 ? 1 : alpha)   (gradient of tf.maximum(alpha*x,x)) */
int mh_leaky_bwd(float* dy, int32_t dy_ld, const float* y, int32_t y_ld, int64_t npix, int32_t nch,
                 float alpha, void* stream);
int mh_fill(float* p, int64_t n, float v, void* stream);
/* diagnostics: stores the device's constant-rate wall clock (ticks of mh_stamp_rate_khz() kHz) into the 8-byte slot -- recorded as a plan op
 * (MH_OP_STAMP) it times the REPLAYED graph from the inside: start of the side lane, end of the input-gradient chain, end of the step */
int mh_stamp(void* slot, void* stream);
int64_t mh_stamp_rate_khz(void);
/* ---- deterministic test mode (MH_DETERMINISTIC=1 in the Python engines).  The step's only order-dependent arithmetic is its float
 * atomics (bias gradients, the warp-gradient scatter, the sampler's image gradient, un-split filter gradients).  After
 * mh_deterministic_add(base, n, twin) every such atomic whose destination lies in [base, base + n) accumulates value * 2^48 into the 64-bit
 * integer twin[dst - base] instead (integer atomics are associative: any arrival order, any replay, the same bits); mh_det_flush adds the
 * twin into the float buffer and clears it -- the engines record it (MH_OP_DET_FLUSH) behind the warp-gradient scatter of every level and
 * in front of the optimizer.  twin: n int64, zero-initialised, owned by the caller.  At most 8 ranges per process; add / remove synchronise the
 * device (not inside a capture).  |sums| must stay below 2^15; resolution 2^-48. */
int mh_deterministic_add(float* base, int64_t n, void* twin);
int mh_deterministic_remove(float* base);
int mh_deterministic_ranges(void);
/* 1 if an addend outside the twin's range (|v| >= 2^15, or NaN: e.g. a huge grad_scale) was SATURATED since the previous call, 0 if not, < 0 = -hipError.
 * Device-synchronising; not inside a stream capture. */
int mh_deterministic_overflow(void);
int mh_det_flush(float* dst, void* twin, int64_t n, void* stream);
/* db[c] += sum_p dz[p][c]  (BiasAddGrad of conv2d_transpose, whose filter gradient runs with swapped operands) */
int mh_bias_grad(const float* dz, int32_t dz_ld, int64_t npix, int32_t nch, float* db, void* stream);
/* the same column sums without float atomics (ABI 15; bit-identical replays of a step): workgroup g stores its partial sums to ws[g][nch] (fully overwritten),
 * nblocks = mh_bias_grad_blocks(npix, nch) (1 .. 1024); sum them with an mh_wgrad_seg {ws, db, nch, nblocks} in the batch's mh_wgrad_reduce launch. */
int mh_bias_grad_blocks(int64_t npix, int32_t nch);
int mh_bias_grad_partial(const float* dz, int32_t dz_ld, int64_t npix, int32_t nch, float* ws, int32_t nblocks, void* stream);

/* ---- the collective of the shared-model mode (BASELINE config 5): RCCL all-reduce over xGMI behind the C-ABI ------------------------------------
 * Streams with PRIVATE models need no collective.  Streams of several GPUs that adapt ONE model sum their flat fp32 gradient buffers (+ the 4 loss floats behind
 * them) once per step, between the backward pass and the optimizer (the reference is single-GPU, Stereo_Online_Adaptation.py:39,114-128: every rank then applies
 * the update the reference applies).  One process per GPU; rank 0 calls mh_comm_unique_id and hands the MH_COMM_ID_BYTES bytes to the others by any means
 * (the host layer uses torch.distributed for exactly that), every rank calls mh_comm_init on its device (collective: returns when all `world` ranks have), then
 * mh_allreduce_sum on its stream -- or records it as a plan op (MH_OP_ALLREDUCE) so that a captured step is ONE hipGraph with the collective inside.
 * RCCL is resolved at run time (dlopen librccl.so; MADNET_HIP_RCCL = a full path): without it these entry points return MH_ERR_UNSUPPORTED and everything else
 * works.  RCCL errors: MH_ERR_COLLECTIVE + mh_last_error(). */
#define MH_COMM_ID_BYTES 128
#define MH_ALLREDUCE_MAX_BUFS 8
int mh_comm_available(void);                                    /* 1 when librccl.so was found and has the entry points (no status, no error message) */
int mh_comm_unique_id(void* id);                                /* ncclGetUniqueId: MH_COMM_ID_BYTES bytes, rank 0 only */
int mh_comm_init(const void* id, int32_t rank, int32_t world, void** comm);      /* ncclCommInitRank on the CURRENT device */
int mh_comm_destroy(void* comm);
int mh_comm_info(void* comm, int32_t* rank, int32_t* world, int32_t* rccl_version);       /* any out pointer may be NULL */
/* in-place fp32 sum over the ranks of n <= MH_ALLREDUCE_MAX_BUFS device buffers as ONE RCCL group (a MAD block's two gradient ranges + the loss tail travel together).
 * Every rank issues the same sequence of calls with the same counts.  Allowed inside a stream capture. */
int mh_allreduce_sum(float* const* bufs, const int64_t* counts, int32_t n, void* comm, void* stream);

/* (the process-wide tuning hooks of the benchmarks live in madnet_hip_tune.h: none of them is part of the reference interface) */

/* host utility: CRC-32C (Castagnoli) of a host buffer, chained through `crc` (0 to start) -- used by the TensorFlow
 * checkpoint importer that replaces tf.train.NewCheckpointReader (Data_utils/weights_utils.py:29). */
uint32_t mh_crc32c(const void* data, int64_t n, uint32_t crc);

/* ---- native plan executor: the host (Python) compiles the network into an array of op
 *      records once; one FFI call replays it (optionally captured into a hipGraph). ------ */
enum { MH_OP_CONV = 1, MH_OP_WGRAD, MH_OP_CORR_FWD, MH_OP_CORR_BWD, MH_OP_WARP_FWD, MH_OP_WARP_BWD,
       MH_OP_RESIZE_FWD, MH_OP_RESIZE_BWD, MH_OP_PAD_REFLECT, MH_OP_LOSS, MH_OP_METRICS,
       MH_OP_MOMENTUM, MH_OP_COPY_CH, MH_OP_LEAKY_BWD, MH_OP_FILL, MH_OP_BIAS_GRAD,
       MH_OP_WGRAD_PARTIAL, MH_OP_WGRAD_REDUCE, MH_OP_PROXY_LOSS, MH_OP_SUPERVISED_LOSS, MH_OP_ADAM, MH_OP_ADAM_ADVANCE,
       MH_OP_RESIZE_IMAGE, MH_OP_LEVEL_FRONT, MH_OP_RESERVED_25 /* (was: transposed filter banks of the retired LDS-free kernel) */, MH_OP_PACK_W, MH_OP_CORR_WARP_BWD, MH_OP_SHADOW_CAST, MH_OP_WGRAD_STREAM, MH_OP_HEAD_BWD, MH_OP_HEAD_FWD, MH_OP_CONV_PLANES, MH_OP_PLANE_SPLIT, MH_OP_STAMP, MH_OP_CONV_PLANES_BWD, MH_OP_DET_FLUSH, MH_OP_CONV_IMAGE,
       MH_OP_ALLREDUCE /* mh_allreduce_sum: p[0] = comm, p[1 .. i[0]] = buffers, i[1 .. i[0]] = counts (floats, < 2^31 each) */,
       MH_OP_FETCH_INPUTS /* mh_fetch_inputs: p[0] = table, p[1 .. i[0]] = destinations, i[1 .. i[0]] = counts (floats, < 2^31 each) */ };
/* i[26] of every op is its scheduling word: low byte = lane (0 = the caller's stream; 1..MH_MAX_LANES-1 = side
 * streams owned by the library: the op is forked from lane 0 right before it, i.e. ordered after everything recorded so
 * far, and runs concurrently with the lane-0 ops that follow); MH_OP_JOIN = lane 0 first waits for all side lanes.
 * mh_plan_run joins every side lane before it returns.  Under mh_graph_begin/end the lanes become parallel branches
 * of the captured hipGraph. */
#define MH_MAX_LANES 5
#define MH_OP_JOIN 0x100
/* side-lane op: launch it at once.  (Default: mh_plan_run takes the fork edge where the op stands but launches side-lane ops only after the
 * next lane-0 op, so that in a captured graph the lane-0 successor is the first child of the fork node and keeps its hardware queue; the
 * price is that the runtime starts such a side chain late.  A batch with a lot of work is worth the queue hop of the critical path.) */
#define MH_OP_NODEFER 0x200      /* (ignored on an op that carries a lane mask for its OWN side lane: such an op is always deferred) */
/* bits 16..23 of the scheduling word: lane 0 first waits for exactly the side lanes in this mask (bit l = lane l), leaving the others
 * running -- e.g. the scatter half of a warp gradient joined right before the pyramid backward while the filter gradients go on.  On an op of a
 * SIDE lane the mask makes that lane (not lane 0) wait for the named lanes: lane-to-lane edges, lane 0 is not held up */
#define MH_OP_JOIN_LANES(mask) (((mask) & 0xff) << 16)
typedef struct mh_op {
    int32_t kind;
    int32_t i[27];
    float f[4];
    void* p[12];
    int64_t n;
} mh_op;
int mh_plan_run(const mh_op* ops, int32_t nops, void* stream);
/* Several INDEPENDENT plans -- the step chains of private-model streams that share one GPU -- as parallel branches: plan 0 on
 * `stream`, plan i > 0 on a library-owned branch stream forked from `stream` and joined before the call returns, each with side lanes of its
 * own; captured between mh_graph_begin / mh_graph_end the S chains are S concurrent branches of ONE graph.  Plans may use lanes
 * 0 .. MH_MAX_LANES - 2.  mh_plans_prepare(nplans) creates the branch streams and must be called once per thread outside a capture. */
typedef struct mh_plan_ref { const mh_op* ops; int32_t nops; int32_t reserved; } mh_plan_ref;
int mh_plans_prepare(int32_t nplans);
int mh_plans_run(const mh_plan_ref* plans, int32_t nplans, void* stream);
/* Threading: every entry point is re-entrant; the side streams / events of mh_plan_run are per host thread and device, the
 * tuning hooks are process-wide atomics meant for benchmarks.
 * hipGraph wrappers: capture everything launched on `stream` between begin/end. */
int mh_graph_begin(void* stream);
int mh_graph_end(void* stream, void** graph_exec_out);
int mh_graph_launch(void* graph_exec, void* stream);
int mh_graph_destroy(void* graph_exec);
/* HIP-event timing helpers on the caller's stream (bench.py roofline leg) */
int mh_event_create(void** ev);
int mh_event_record(void* ev, void* stream);
int mh_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms);
int mh_event_destroy(void* ev);
int mh_stream_sync(void* stream);

#ifdef __cplusplus
}
#endif
#endif
