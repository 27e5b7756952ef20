// TEST INFRASTRUCTURE: CPU emulation of csrc/mh_bf16_intrin.h (shadows it through the include path).
#pragma once
static inline unsigned emul_bf16_bits(float f) {
    unsigned u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;       // NaN
    u += 0x7fffu + ((u >> 16) & 1u);                                      // round to nearest even
    return u >> 16;
}
static inline float emul_bf16_val(unsigned h) { unsigned u = h << 16; float f; memcpy(&f, &u, 4); return f; }
static inline unsigned mh_pack_bf16(float lo, float hi) { return emul_bf16_bits(lo) | (emul_bf16_bits(hi) << 16); }

static inline void mh_split_bf16x2(float a, float b, unsigned& hi, unsigned& lo) {
    const unsigned ha = emul_bf16_bits(a), hb = emul_bf16_bits(b);
    hi = ha | (hb << 16);
    lo = mh_pack_bf16(a - emul_bf16_val(ha), b - emul_bf16_val(hb));
}

static inline f32x4 mh_mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
    emul::Wave& w = emul::my_wave();
    const int lane = emul::tls().cur_index & 63;
    const int par = w.gen & 1;
    for (int e = 0; e < 4; ++e) { w.ua[par][lane][e] = a[e]; w.ub[par][lane][e] = b[e]; }
    emul::wave_rendezvous(w);
    const int col = lane & 15;
    f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = (lane >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 32; ++k) {
            const int kq = k >> 3, e = k & 7;
            const unsigned aw = w.ua[par][row + 16 * kq][e >> 1], bw = w.ub[par][col + 16 * kq][e >> 1];
            const float av = emul_bf16_val((e & 1) ? (aw >> 16) : (aw & 0xffffu));
            const float bv = emul_bf16_val((e & 1) ? (bw >> 16) : (bw & 0xffffu));
            acc += av * bv;
        }
        d[r] = acc;
    }
    return d;
}

// ds_read_b64_tr_b16 (semantics measured on the MI355X, scripts/exp/tr_probe.hip): lane i of a 16-lane group receives, as element j, element
// (i & 3) of the 4 halfs lane 4*j + (i >> 2) of the group addressed
static inline uint2 mh_lds_read_tr16(const unsigned short* p) {
    emul::Wave& w = emul::my_wave();
    const int lane = emul::tls().cur_index & 63;
    const int par = w.gen & 1;
    const unsigned long long a = (unsigned long long)(uintptr_t)p;
    w.ua[par][lane][0] = (unsigned)(a & 0xffffffffull); w.ua[par][lane][1] = (unsigned)(a >> 32);
    emul::wave_rendezvous(w);
    const int g0 = lane & ~15, i = lane & 15;
    unsigned short h[4];
    for (int j = 0; j < 4; ++j) {
        const int s = g0 + 4 * j + (i >> 2);
        const unsigned long long sa = (unsigned long long)w.ua[par][s][0] | ((unsigned long long)w.ua[par][s][1] << 32);
        h[j] = ((const unsigned short*)(uintptr_t)sa)[i & 3];
    }
    uint2 r;
    r.x = (unsigned)h[0] | ((unsigned)h[1] << 16);
    r.y = (unsigned)h[2] | ((unsigned)h[3] << 16);
    return r;
}

// ---- primitives of csrc/wgrad_stream.hip (see csrc/mh_bf16_intrin.h) ---------------------------------------------------------------
typedef float f32x16 __attribute__((vector_size(64)));
static inline f32x16 mh_mfma_bf16_32(u32x4 a, u32x4 b, f32x16 c) {
    emul::Wave& w = emul::my_wave();
    const int lane = emul::tls().cur_index & 63;
    const int par = w.gen & 1;
    for (int e = 0; e < 4; ++e) { w.ua[par][lane][e] = a[e]; w.ub[par][lane][e] = b[e]; }
    emul::wave_rendezvous(w);
    const int col = lane & 31;
    f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float acc = c[r];
        for (int k = 0; k < 16; ++k) {
            const int kq = k >> 3, e = k & 7;                      // lane (row/col + 32 * kq) holds k = 8 kq .. 8 kq + 7
            const unsigned aw = w.ua[par][row + 32 * kq][e >> 1], bw = w.ub[par][col + 32 * kq][e >> 1];
            const float av = emul_bf16_val((e & 1) ? (aw >> 16) : (aw & 0xffffu));
            const float bv = emul_bf16_val((e & 1) ? (bw >> 16) : (bw & 0xffffu));
            acc += av * bv;
        }
        d[r] = acc;
    }
    return d;
}
// LDS DMA: synchronous on the emulator (a missing / wrong vmcnt wait cannot be detected here; the GPU tests cover that)
struct mh_dma_src { const char* base; unsigned bytes; };
static inline mh_dma_src mh_make_dma_src(const void* p, unsigned bytes) { mh_dma_src r; r.base = (const char*)p; r.bytes = bytes; return r; }
static inline void mh_glds16(const mh_dma_src& r, void* lds_wave_base, int voff) {
    const int lane = emul::tls().cur_index & 63;
    const unsigned off = (unsigned)voff;
    unsigned char* dst = (unsigned char*)lds_wave_base + 16 * lane;
    if ((unsigned long long)off + 16ull <= r.bytes) memcpy(dst, r.base + off, 16); else memset(dst, 0, 16);
}
#define MH_KEEP_VGPR(x) do { } while (0)
typedef const float MH_CONST_F32;
#define MH_CONST_F32_PTR(p) ((MH_CONST_F32*)(p))
#define MH_WAIT_VMCNT(n) do { } while (0)
#define MH_WAIT_LGKMCNT0() do { } while (0)
// wave-local LDS exchange: the lanes of a wave are fibers here, so the fence is a rendezvous of the wave
static inline void mh_wave_sync() { emul::wave_rendezvous(emul::my_wave()); }
template <int P> static inline void mh_setprio() {}
