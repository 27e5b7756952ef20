// TEST INFRASTRUCTURE (oracle/_ref build): stand-in for TensorFlow's cuda_kernel_helper.h, which
// /root/reference/Nets/Native/shift_corr.cu.cc:5 includes for exactly one macro, the grid-stride loop
// CUDA_1D_KERNEL_LOOP (used by the two backward kernels, :75 and :137).  Semantics as documented by
// TF 1.12 (tensorflow/core/util/cuda_kernel_helper.h): i walks [0, n) with stride gridDim.x*blockDim.x.
#pragma once
#include <hip/hip_runtime.h>
#define CUDA_1D_KERNEL_LOOP(i, n) \
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += blockDim.x * gridDim.x)
