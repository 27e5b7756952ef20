// mh_bf16_intrin.h -- the two bf16 primitives of the throughput-mode GEMM kernels (gfx950):
//   mh_pack_bf16  : two f32 -> packed bf16x2, round-to-nearest-even (v_cvt_pk_bf16_f32)
//   mh_mfma_bf16  : v_mfma_f32_16x16x32_bf16, fp32 accumulate.  Operand layout (wave64):
//                   lane l holds A[i=l&15][k=8*(l>>4)..+7] / B[k=8*(l>>4)..+7][j=l&15] as 8 bf16 (4 dwords);
//                   C/D: col = l&15, row = (l>>4)*4 + r   (same as the f32 16x16x4 form)
// Included as <mh_bf16_intrin.h> so that the CPU functional emulator (tests/emul) can shadow it.
#pragma once
typedef __bf16 mh_bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 mh_bf16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned mh_pack_bf16(float lo, float hi) {
    const mh_bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}

__device__ __forceinline__ f32x4 mh_mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mh_bf16x8_t, a), __builtin_bit_cast(mh_bf16x8_t, b), c, 0, 0, 0);
}

// split-bf16 ("bf16x3") operand: x = hi + lo up to 2^-16 |x|, hi = bf16(x), lo = bf16(x - hi) (the subtraction is exact in fp32).
// Packs the hi halves and the lo halves of two values.
__device__ __forceinline__ void mh_split_bf16x2(float a, float b, unsigned& hi, unsigned& lo) {
    const __bf16 ha = (__bf16)a, hb = (__bf16)b;
    const mh_bf16x2_t h = {ha, hb};
    hi = __builtin_bit_cast(unsigned, h);
    lo = mh_pack_bf16(a - (float)ha, b - (float)hb);
}
