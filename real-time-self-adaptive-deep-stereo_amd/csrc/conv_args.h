// conv_args.h -- kernel argument block shared by the convolution translation units (conv.hip, conv_patch.hip)
#pragma once
#include "mh_common.h"

struct ConvArgs {
    const float* in; const float* w; const float* bias; float* out; const float* mask_ref;
    const void* wb; unsigned wb_bytes;     // fragment bank of w (mh_pack_weights), or null
    unsigned short* shadow; int shadow_ld; // != null: the epilogue also writes bf16(out) to shadow[pixel][shadow_ld] (operand of mh_wgrad_stream)
    unsigned short* shadow_lo;             // != null (with shadow): the epilogue also writes bf16(out - bf16(out)): shadow / shadow_lo = the hi / lo planes mh_conv2d_planes reads
    int shadow_lo_done;                    // set where a kernel family's epilogue wrote it (else conv_entry splits afterwards)
    const unsigned short* in_shadow; unsigned in_shadow_bytes;   // != null: bf16 shadow of `in` (pixel stride = K rounded up to 32, zero padded): the patch-staged
                                                                 // input-gradient kernel stages it as it is instead of converting the fp32 tensor
    const unsigned short* mask_shadow; unsigned mask_shadow_bytes; int mask_shadow_ld;   // != null: the leaky mask reads the bf16 shadow of mask_ref (sign test only)
    int no_f32_out;                      // 1: only the bf16 shadow of the result is stored (the consumers take the shadow; needs shadow, no accumulate)
    float* out2; float* out3; int out2_ld, out3_ld;   // single-output-channel forward conv (mh_conv2d_head): copies of the result (a concat slot, the next stage's accumulator)
    int shadow_done;                       // set by the launcher of a kernel family whose epilogue wrote the shadow (else conv_entry casts afterwards)
#ifdef MH_PHASE_TIMING
    unsigned long long* dbg;                // experiment build only (scripts/exp/phase_timing.sh): per-workgroup phase time stamps
#endif
    int in_ld, out_ld, mask_ld;
    int B, Hi, Wi, Ho, Wo;
    int K, N, G, taps;
    int kh, kw, stride, dil, pad_t, pad_l;
    int mode, w_trans, accumulate, sshift;
    unsigned in_bytes, w_bytes, out_bytes, mask_bytes;
    int bf16;        // throughput mode: bf16 MFMA inputs, fp32 accumulate
    int x3;          // split-bf16 request (precision 2): kernels with an x3 instance run hi/lo bf16 operands, 3 MFMAs per product; the others exact fp32
    int vecC;        // 16-byte epilogue legal (N, out_ld, mask_ld multiples of 4, aligned pointers)
    int vecCpad;     // input gradient whose N is NOT a multiple of 4 but whose rows are padded to one (out_ld, mask_ld >= round_up(N, 4)): the patch-staged
                     // kernel's 16-byte epilogue is legal -- the channels N .. round_up(N, 4) - 1 (row padding nobody reads) receive zeros
    int M;           // B*Ho*Wo
    int vecA, vecB;  // 16-byte vector loads legal for A / B
    int mtiles, ntiles;
    float alpha, mask_alpha;
    int mask_c0, mask_c1;   // channel range the leaky-grad mask applies to
    // Stride-2 input gradient / transposed conv as 4 stride-1 sub-problems, one per output parity class: class c covers
    // the output pixels (2*qy + py, 2*qx + px) and ONLY the taps that land on the input lattice for that parity
    // (iy = qy + dy[t]); walking every tap with a lattice mask instead wastes 3/4 of the loads and MFMAs.
    int ncls;               // 0 = off
    struct Cls { int py, px, Hq, Wq, M, tile0, ntaps, pad; signed char dy[16], dx[16]; unsigned char id[16]; } cls[4];
};

// conv_patch.hip: patch-staged bf16 kernel for the stride-1 3x3 (dilated) layers
bool mh_conv_patch_ok(const ConvArgs& a);
int mh_conv_patch_launch(ConvArgs& a, hipStream_t s);     // a.M < 0: attribute set-up only
extern "C" int mh_tune_conv_patch(int mode);
// conv_patch.hip: fragment-bank kernel of the small layers (a.wb = the bank mh_pack_weights wrote for this mode / precision)
bool mh_conv_bank_small_ok(const ConvArgs& a);
int mh_conv_bank_small_launch(ConvArgs& a, hipStream_t s);     // a.M < 0: attribute set-up only
// conv_rows.hip: row-streaming kernel of the thin 3x3 layers at 1/2 resolution (<= 16 input channels, stride 1)
bool mh_conv_rows_ok(const ConvArgs& a);
int mh_conv_rows_launch(ConvArgs& a, hipStream_t s);
