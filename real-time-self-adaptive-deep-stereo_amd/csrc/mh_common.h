// mh_common.h -- shared helpers for the gfx950 kernels of libmadnet_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/madnet_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

void mh_set_error(const char* fmt, ...);
int mh_check_launch(const char* what);
// records (thread-local) which kernel instance an entry point just dispatched; read back through mh_last_kernel()
void mh_note_kernel(const char* fmt, ...);

#define MH_REQUIRE(cond, code, ...)                \
    do {                                           \
        if (!(cond)) {                             \
            mh_set_error(__VA_ARGS__);             \
            return (code);                         \
        }                                          \
    } while (0)

static inline int mh_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
// hipFuncSetAttribute (the > 64 KiB dynamic-LDS opt-in) applies to the CURRENT device only: the per-instantiation "done" flags are one bit per device
// id (a function-static bool made the first launch on a second device of the same process fail)
static inline uint64_t mh_device_bit() { int d = 0; if (hipGetDevice(&d) != hipSuccess) d = 0; return 1ull << (d & 63); }
static inline bool mh_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// ---- buffer (SRD) loads with hardware bounds checking --------------------------------------
// Every tile load of the GEMM kernels goes through a raw buffer descriptor built from kernel
// arguments (wave-uniform => no waterfall loop): an element that must read as zero (outside the
// image, padded tap, past the tensor) simply gets an out-of-range byte offset and the hardware
// returns 0 -- the load itself stays UNCONDITIONAL, so hipcc keeps it in flight across the MFMA
// block instead of draining vmcnt(0) at a branch join (cdna_hip_programming.md trap (c), T8).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define MH_OOB ((int)0x7fffffff)          /* > any num_records we accept (tensors < 2 GiB) */

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mh_make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 mh_buf_load4(__amdgpu_buffer_rsrc_t r, int byte_off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
    return __builtin_bit_cast(float4, v);
}
__device__ __forceinline__ float mh_buf_load1(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}

// stores through a descriptor: a lane whose offset is out of range (MH_OOB) stores nothing -- unconditional, straight-line epilogues whose
// outstanding-store count the compiler can see (a store under a divergent branch makes every later s_waitcnt vmcnt conservative)
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void mh_buf_store4(__amdgpu_buffer_rsrc_t r, int byte_off, float4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, byte_off, 0, 0);
}
__device__ __forceinline__ void mh_buf_store2(__amdgpu_buffer_rsrc_t r, int byte_off, unsigned a, unsigned b) {
    __builtin_amdgcn_raw_buffer_store_b64((u32x2){a, b}, r, byte_off, 0, 0);
}

#include <mh_bf16_intrin.h>     // bf16 pack + bf16 MFMA (angle brackets: the CPU emulator shadows this header)

// XCD-aware, bijective remap of a linear workgroup id (8 XCDs; block b runs on XCD b%8):
// every XCD gets a contiguous chunk of the logical tile space so neighbouring tiles share
// one L2 (cdna_hip_programming.md T1, bijective variant).
__device__ __forceinline__ int mh_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + idx;
}

// ---- deterministic accumulation (mh_deterministic_add, MH_DETERMINISTIC=1 in the engines) ------------------------------------------------------------
// The step's only order-dependent arithmetic is its float atomics: bias gradients (one atomicAdd per workgroup and channel), the warp-gradient
// scatter of mh_corr_warp_bwd / mh_warp_bwd, the sampler's image gradient, un-split filter gradients.  In deterministic mode a destination inside a
// registered range accumulates into a 64-bit FIXED-POINT twin instead (value * 2^48 -- the per-pixel gradients of a mean-reduced loss are ~1e-8 --, integer
// atomics: associative, so any arrival order gives the same
// bits); mh_det_flush adds the twin into the float buffer and clears it.  One table per translation unit (no relocatable device code): lib.hip keeps
// them in step.  Off (n = 0): one scalar load and a uniform branch per atomic site.
struct mh_det_table { int n; int pad; const float* lo[8]; const float* hi[8]; long long* acc[8]; };
static __device__ mh_det_table g_mh_det;
// Supported addend range of the fixed-point twin: |v| < 2^15 (value * 2^48 must fit 63 bits).  A larger addend -- a loss scale (grad_scale) far above the
// mean-reduced losses' -- is SATURATED, and says so: a sticky per-translation-unit flag that mh_deterministic_overflow() collects.
static __device__ int g_mh_det_ovf;
__device__ __forceinline__ void mh_atomic_add(float* dst, float v) {
    const int n = g_mh_det.n;
    for (int i = 0; i < n; ++i)
        if (dst >= g_mh_det.lo[i] && dst < g_mh_det.hi[i]) {
            // |v| is clamped below 2^15 (llrintf of a larger product is undefined; the running sum of a mean-reduced loss's gradients stays far inside)
            if (!(fabsf(v) < 32767.0f)) g_mh_det_ovf = 1;           // (also NaN) reported by mh_deterministic_overflow()
            atomicAdd(reinterpret_cast<unsigned long long*>(g_mh_det.acc[i] + (dst - g_mh_det.lo[i])),
                      (unsigned long long)(long long)llrintf(fminf(fmaxf(v, -32767.0f), 32767.0f) * 281474976710656.0f));
            return;
        }
    atomicAdd(dst, v);
}
static inline int mh_det_upload(const mh_det_table& t) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_mh_det), &t, sizeof(t)); }
// this translation unit's saturation flag: read (device-synchronising copy) and cleared; < 0 = -hipError
static inline int mh_det_overflow_take() {
    int v = 0;
    const int zero = 0;
    if (hipError_t e = hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_mh_det_ovf), sizeof(v))) return -(int)e;
    if (v) if (hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_mh_det_ovf), &zero, sizeof(zero))) return -(int)e;
    return v ? 1 : 0;
}

// Workgroup -> segment of a batched launch's device table (segs[].blk0 = exclusive prefix of the segments' block counts).  The table's blk0 column goes
// to LDS first (one coalesced round trip) and the binary search runs there: searched in global memory it is log2(nseg) DEPENDENT loads in front of a
// workgroup's first useful load (7 for the 73-layer bank table of mh_pack_weights: ~3 us per workgroup, round 4).  Called by every thread of a
// 256-thread workgroup (it contains a barrier).
template <class SEG>
__device__ __forceinline__ int mh_find_seg(const SEG* __restrict__ segs, int nseg, int blk) {
    __shared__ int s_blk0[256];
    int lo = 0, hi = nseg - 1;
    if (nseg <= 256) {
        if ((int)threadIdx.x < nseg) s_blk0[threadIdx.x] = segs[threadIdx.x].blk0;
        __syncthreads();
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (s_blk0[mid] <= blk) lo = mid; else hi = mid - 1;
        }
        return lo;
    }
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (segs[mid].blk0 <= blk) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// Division of a small non-negative index by a LAUNCH-CONSTANT divisor.  gfx950 has no integer divider: for `lin / tiles_x` hipcc emits a ~30-instruction
// dependent sequence (v_rcp_iflag_f32, Newton step, two corrections) -- five of them in a row decode a workgroup's tile before its first load address is
// known: ~0.6 us at the head of EVERY conv launch (round 4: scripts/exp/node_floor.py, the ISA of conv_bank_small_kernel).  The host computes the
// round-up magic multiplier instead: n / d = umulhi(n, 2^32 / d + 1), exact while n * d < 2^32 (tile counts: n < 2^20, d < 2^12).
struct mh_fastdiv { unsigned d, m; };
static inline mh_fastdiv mh_make_fastdiv(int d) {
    mh_fastdiv f;
    f.d = (unsigned)(d > 0 ? d : 1);
    f.m = f.d > 1 ? (unsigned)((1ull << 32) / f.d + 1ull) : 0u;
    return f;
}
__device__ __forceinline__ int mh_fdiv(int n, const mh_fastdiv& f) { return f.d > 1 ? (int)__umulhi((unsigned)n, f.m) : n; }
// the round-up multiplier is exact while n * d < 2^32: every launcher checks its largest dividend against its largest divisor
static inline bool mh_fastdiv_ok(int64_t nmax, int64_t dmax) { return nmax >= 0 && dmax >= 1 && nmax * dmax < (1ll << 32); }
// lin -> (column tile, x tile, y tile, lattice phase x, lattice phase y, batch) of the patch / bank / planes kernels
struct mh_tile_decode { mh_fastdiv ntn, tx, ty, dd; };
static inline mh_tile_decode mh_make_tile_decode(int ntiles_n, int tiles_x, int tiles_y, int d) {
    mh_tile_decode t;
    t.ntn = mh_make_fastdiv(ntiles_n); t.tx = mh_make_fastdiv(tiles_x); t.ty = mh_make_fastdiv(tiles_y); t.dd = mh_make_fastdiv(d);
    return t;
}
__device__ __forceinline__ void mh_decode_tile(int lin, const mh_tile_decode& t, int& tile_n, int& ttx, int& tty, int& cx, int& cy, int& b) {
    int q = mh_fdiv(lin, t.ntn); tile_n = lin - q * (int)t.ntn.d; lin = q;
    q = mh_fdiv(lin, t.tx); ttx = lin - q * (int)t.tx.d; lin = q;
    q = mh_fdiv(lin, t.ty); tty = lin - q * (int)t.ty.d; lin = q;
    q = mh_fdiv(lin, t.dd); cx = lin - q * (int)t.dd.d; lin = q;
    q = mh_fdiv(lin, t.dd); cy = lin - q * (int)t.dd.d;
    b = q;
}

__device__ __forceinline__ float mh_wave_sum(float v) {
    v += __shfl_xor(v, 32);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 1);
    return v;
}
