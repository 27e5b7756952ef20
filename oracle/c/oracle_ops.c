/* CPU ORACLE (test infrastructure, NOT product code) -- plain-C restatement.
 *
 * Third, independent restatement of the arithmetic kernels of the hot path, used by
 * tests/ to cross-check oracle/tf_ops.py (torch) and the HIP kernels.  PARITY UNPINNED:
 * TensorFlow 1.12 cannot run here and the reference has no tests; formulas follow
 * SURVEY.md App. A (A.1 SAME padding, A.14 correlation gradients).
 *
 *  oc_conv2d      : Nets/sharedLayers.py:54-77  (tf.nn.conv2d / atrous_conv2d 'SAME' + bias + leaky)
 *  oc_corr_fwd    : Nets/sharedLayers.py:41-51  (correlation_tf)
 *  oc_corr_bwd    : gradient of the above (the TF formulation; NOT Nets/Native/shift_corr.cu.cc:73-191,
 *                   whose backward is defective)
 *  oc_resize      : tf.image.resize_images bilinear, legacy
 * All tensors NHWC float32, accumulation in double.
 */
#include <math.h>
#include <stdint.h>

static void same_pad(int in, int k, int s, int d, int *out, int *before) {
    int keff = (k - 1) * d + 1;
    int o = (in + s - 1) / s;
    int total = (o - 1) * s + keff - in;
    if (total < 0) total = 0;
    *out = o; *before = total / 2;
}

void oc_conv2d(const float *x, const float *w, const float *b, float *y,
               int B, int H, int W, int Ci, int Co, int kh, int kw, int stride, int dil, float alpha) {
    int Ho, Wo, pt, pl;
    same_pad(H, kh, stride, dil, &Ho, &pt);
    same_pad(W, kw, stride, dil, &Wo, &pl);
    for (int n = 0; n < B; ++n)
        for (int oy = 0; oy < Ho; ++oy)
            for (int ox = 0; ox < Wo; ++ox)
                for (int co = 0; co < Co; ++co) {
                    double acc = 0.0;
                    for (int ky = 0; ky < kh; ++ky) {
                        int iy = oy * stride + ky * dil - pt;
                        if (iy < 0 || iy >= H) continue;
                        for (int kx = 0; kx < kw; ++kx) {
                            int ix = ox * stride + kx * dil - pl;
                            if (ix < 0 || ix >= W) continue;
                            const float *xp = x + (((int64_t)n * H + iy) * W + ix) * Ci;
                            const float *wp = w + ((int64_t)(ky * kw + kx) * Ci) * Co + co;
                            for (int ci = 0; ci < Ci; ++ci) acc += (double)xp[ci] * (double)wp[(int64_t)ci * Co];
                        }
                    }
                    acc += b ? (double)b[co] : 0.0;
                    if (alpha != 1.0f && !(acc > 0.0)) acc *= (double)alpha;
                    y[(((int64_t)n * Ho + oy) * Wo + ox) * Co + co] = (float)acc;
                }
}

void oc_corr_fwd(const float *L, const float *R, float *out, int B, int H, int W, int C, int md, int stride) {
    int D = 0;
    for (int i = -md; i <= md; i += stride) ++D;
    for (int64_t row = 0; row < (int64_t)B * H; ++row)
        for (int x = 0; x < W; ++x) {
            int j = 0;
            for (int i = -md; i <= md; i += stride, ++j) {
                int xs = x + i;
                double acc = 0.0;
                if (xs >= 0 && xs < W)
                    for (int c = 0; c < C; ++c)
                        acc += (double)L[(row * W + x) * C + c] * (double)R[(row * W + xs) * C + c];
                out[(row * W + x) * D + j] = (float)(acc / C);
            }
        }
}

/* dL[y,x,c] = (1/C) sum_i g[y,x,j(i)] R[y,x+i,c] ; dR[y,x',c] = (1/C) sum_i g[y,x'-i,j(i)] L[y,x'-i,c] */
void oc_corr_bwd(const float *L, const float *R, const float *g, float *dL, float *dR,
                 int B, int H, int W, int C, int md, int stride) {
    int D = 0;
    for (int i = -md; i <= md; i += stride) ++D;
    for (int64_t row = 0; row < (int64_t)B * H; ++row)
        for (int x = 0; x < W; ++x)
            for (int c = 0; c < C; ++c) {
                double a = 0.0, r = 0.0;
                int j = 0;
                for (int i = -md; i <= md; i += stride, ++j) {
                    int xs = x + i;
                    if (xs >= 0 && xs < W) a += (double)g[(row * W + x) * D + j] * (double)R[(row * W + xs) * C + c];
                    int xl = x - i;
                    if (xl >= 0 && xl < W) r += (double)g[(row * W + xl) * D + j] * (double)L[(row * W + xl) * C + c];
                }
                dL[(row * W + x) * C + c] = (float)(a / C);
                dR[(row * W + x) * C + c] = (float)(r / C);
            }
}

void oc_resize(const float *x, float *y, int B, int H, int W, int C, int oh, int ow) {
    float sy = (float)H / (float)oh, sx = (float)W / (float)ow;
    for (int n = 0; n < B; ++n)
        for (int oy = 0; oy < oh; ++oy) {
            float fy = (float)oy * sy; int y0 = (int)fy; int y1 = y0 + 1 < H ? y0 + 1 : H - 1; float ty = fy - (float)y0;
            for (int ox = 0; ox < ow; ++ox) {
                float fx = (float)ox * sx; int x0 = (int)fx; int x1 = x0 + 1 < W ? x0 + 1 : W - 1; float tx = fx - (float)x0;
                for (int c = 0; c < C; ++c) {
                    double tl = x[(((int64_t)n * H + y0) * W + x0) * C + c], tr = x[(((int64_t)n * H + y0) * W + x1) * C + c];
                    double bl = x[(((int64_t)n * H + y1) * W + x0) * C + c], br = x[(((int64_t)n * H + y1) * W + x1) * C + c];
                    double top = tl + (tr - tl) * tx, bot = bl + (br - bl) * tx;
                    y[(((int64_t)n * oh + oy) * ow + ox) * C + c] = (float)(top + (bot - top) * ty);
                }
            }
        }
}
