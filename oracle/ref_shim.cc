// TEST INFRASTRUCTURE (oracle/_ref build): extern "C" doors onto the reference's own launchers so ctypes can call them.
// The declarations are the ones the reference's op wrapper uses (/root/reference/Nets/Native/shift_corr.cc:22-23,58-60);
// the definitions come from /root/reference/Nets/Native/shift_corr.cu.cc:193,235, compiled where it lies (oracle/Makefile).
// Layouts (shift_corr.cc:36-56, sharedLayers.py:31-39): in0 / in1 = NHWC, W already zero-padded by max_disp on both sides
// (in_w = W + 2*max_disp); out = NCHW [batch, 2*max_disp+1, in_h, W].  The launchers use the NULL stream and never check
// errors; the shim synchronises the device and returns hipGetLastError().
#include <hip/hip_runtime.h>

void ShiftCorrKernelLauncher(const float* values0, const float* values1, const int max_disp, const int batch_size,
                             const int in_h, const int in_w, const int in_channels, float* out);
void ShiftCorrGradKernelLauncher(const float* input0, const float* input1, const float* grad, const int max_disp,
                                 const int batch_size, const int height, const int paddedwidth, const int channels,
                                 float* output0, float* output1);

extern "C" int ref_shift_corr(const float* in0, const float* in1, int max_disp, int batch, int in_h, int in_w_padded,
                              int channels, float* out) {
    ShiftCorrKernelLauncher(in0, in1, max_disp, batch, in_h, in_w_padded, channels, out);
    hipError_t e = hipDeviceSynchronize();
    return e != hipSuccess ? (int)e : (int)hipGetLastError();
}

extern "C" int ref_shift_corr_grad(const float* in0, const float* in1, const float* grad, int max_disp, int batch,
                                   int height, int padded_width, int channels, float* out0, float* out1) {
    ShiftCorrGradKernelLauncher(in0, in1, grad, max_disp, batch, height, padded_width, channels, out0, out1);
    hipError_t e = hipDeviceSynchronize();
    return e != hipSuccess ? (int)e : (int)hipGetLastError();
}
