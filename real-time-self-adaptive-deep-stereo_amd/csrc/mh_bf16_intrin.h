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

// ds_read_b64_tr_b16: the LDS transposing read of gfx950.  Every lane passes the address of 4 consecutive bf16 (8-byte aligned); within each
// group of 16 lanes, lane i receives, as element j, element (i & 3) of what lane 4*j + (i >> 2) of the group addressed (measured on the MI355X:
// scripts/exp/tr_probe.hip, profiles/r02_tr_probe.txt).  With lane s addressing row (s >> 2), elements 4*(s & 3)..+3 of a row-major
// [4 rows][16 columns] bf16 block, lane i gets column i of the 4 rows: the MFMA operand order (8 consecutive k per lane = two such reads) out
// of an image whose rows are the REDUCTION index -- NHWC pixels with channels contiguous, exactly as they sit in memory.
typedef short mh_v4s_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 mh_lds_read_tr16(const unsigned short* p) {
    const mh_v4s_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) mh_v4s_t*)p);
    return __builtin_bit_cast(uint2, v);
}

// ---- gfx950 primitives of the streaming filter-gradient kernel (csrc/wgrad_stream.hip) ------------------------------------------------
// v_mfma_f32_32x32x16_bf16: lane l holds A[i = l&31][k = 8*(l>>5) .. +7] / B[k = 8*(l>>5) .. +7][j = l&31] as 8 bf16;
// C/D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5) for register r of 16.
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f32x16 mh_mfma_bf16_32(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mh_bf16x8_t, a), __builtin_bit_cast(mh_bf16x8_t, b), c, 0, 0, 0);
}
// buffer_load_dwordx4 ... lds ("LDS DMA"): every lane copies the 16 bytes at byte offset `voff` of the buffer to LDS address
// lds_wave_base + 16 * lane (the LDS base is wave-uniform: it travels in M0); an out-of-range offset stores zeros.  Asynchronous: counted by
// vmcnt, ordered for the issuing wave's own ds_read only by MH_WAIT_VMCNT.
// Inline asm on purpose: hipcc (ROCm 7.2) answers the builtin form (__builtin_amdgcn_raw_ptr_buffer_load_lds) with an s_waitcnt vmcnt(0) in front
// of the next ds_read -- it cannot tell which LDS bytes a pending DMA writes -- which drains the prefetch every step.  Hidden in asm the DMA is
// invisible to its bookkeeping and the kernel counts it itself (cdna_hip_programming.md 5.7: M0 saved and restored inside the statement).
struct mh_dma_src { u32x4 w; };
__device__ __forceinline__ mh_dma_src mh_make_dma_src(const void* p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    mh_dma_src r;
    r.w = (u32x4){(unsigned)a, (unsigned)(a >> 32) & 0xffffu, bytes, 0x00020000u};      // raw buffer, stride 0 (as mh_make_rsrc)
    return r;
}
__device__ __forceinline__ void mh_glds16(const mh_dma_src& r, void* lds_wave_base, int voff) {
    const unsigned dst = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)lds_wave_base;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(dst), "s"(r.w) : "memory");
}
// keeps an integer in a VGPR as computed (an optimisation barrier on one value): a select between it and a constant then stays a v_cndmask instead of a
// branch around the arithmetic that produced it (a branch near outstanding loads makes hipcc's waitcnt pass drain them at the join)
#define MH_KEEP_VGPR(x) asm volatile("" : "+v"(x))
// a read-only global pointer as a CONSTANT-address-space pointer: loads at wave-uniform addresses through it are scalar loads (s_load_dword*) whose results
// feed v_fmac as SGPR operands -- a filter bank read by every lane alike costs no vector-memory instruction at all
typedef const __attribute__((address_space(4))) float MH_CONST_F32;
#define MH_CONST_F32_PTR(p) ((MH_CONST_F32*)(p))
#define MH_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define MH_WAIT_LGKMCNT0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// wave-local LDS exchange (a wave writes an LDS block and its lanes read each other's words back): the LDS executes one wave's operations in order, so all that is needed is
// that the compiler keeps the order and the data has arrived -- a wavefront-scope fence, no s_barrier
__device__ __forceinline__ void mh_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// s_setprio: the wave's issue priority (0 .. 3) among the waves of its SIMD
template <int P> __device__ __forceinline__ void mh_setprio() { __builtin_amdgcn_s_setprio(P); }
