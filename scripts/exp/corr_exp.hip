// Standalone experiment (not part of the product): what bounds the D=5 correlation at B=64 x 96x320x32 ?
// build: hipcc -O3 --offload-arch=gfx950 -o scripts/exp/corr_exp scripts/exp/corr_exp.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int C = 32, C4 = 8, MD = 2, D = 5;

// V0: pure streaming read of both inputs (ceiling for "two input streams, tiny output")
__global__ __launch_bounds__(256) void v0_stream(const float4* L, const float4* R, float* out, long n4) {
    float acc = 0.f;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 l = L[i], r = R[i];
        acc += l.x * r.x + l.y * r.y + l.z * r.z + l.w * r.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

// V1: direct, 8 lanes per pixel, 6 loads per lane (as the product kernel), one pixel group per iteration
template <int UNROLL>
__global__ __launch_bounds__(256) void v1_direct(const float* L, const float* R, float* out, int npix, int W) {
    const int tid = threadIdx.x, sub = tid & 7;
    const int nit = (npix + 32 * UNROLL - 1) / (32 * UNROLL);
    for (int it = blockIdx.x; it < nit; it += gridDim.x) {
        float4 l[UNROLL], r[UNROLL][D];
        int pixs[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int pix = (it * UNROLL + u) * 32 + (tid >> 3);
            pixs[u] = pix;
            const bool live = pix < npix;
            const int pp = live ? pix : 0;
            const int x = pp % W;
            l[u] = *reinterpret_cast<const float4*>(L + (long)pp * C + sub * 4);
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const int xs = x + j - MD;
                const bool ok = (unsigned)xs < (unsigned)W;
                r[u][j] = ok ? *reinterpret_cast<const float4*>(R + (long)(pp + j - MD) * C + sub * 4) : make_float4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            float a[D];
#pragma unroll
            for (int j = 0; j < D; ++j) {
                a[j] = l[u].x * r[u][j].x + l[u].y * r[u][j].y + l[u].z * r[u][j].z + l[u].w * r[u][j].w;
                a[j] += __shfl_xor(a[j], 4); a[j] += __shfl_xor(a[j], 2); a[j] += __shfl_xor(a[j], 1);
            }
            if (pixs[u] < npix && sub < D) {
                float v = a[0];
#pragma unroll
                for (int j = 1; j < D; ++j) v = (sub == j) ? a[j] : v;
                out[(long)pixs[u] * D + sub] = v * (1.0f / C);      // 5 lanes write 20 contiguous bytes
            }
        }
    }
}

// V2: LDS window.  workgroup = 64-pixel row segment; R window (68 px) staged in LDS, L in registers.
__global__ __launch_bounds__(256) void v2_lds(const float* L, const float* R, float* out, int rows, int W) {
    __shared__ float4 Rs[(64 + 2 * MD) * C4];
    const int tid = threadIdx.x, sub = tid & 7;
    const int segs = W / 64;
    for (int blk = blockIdx.x; blk < rows * segs; blk += gridDim.x) {
        const int row = blk / segs, x0 = (blk % segs) * 64;
        const float* Rrow = R + (long)row * W * C;
        const float* Lrow = L + (long)row * W * C;
        float4 l[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) l[u] = *reinterpret_cast<const float4*>(Lrow + (long)(x0 + u * 32 + (tid >> 3)) * C + sub * 4);
        float4 rv[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int q = tid + 256 * k;
            const int xs = x0 - MD + (q >> 3);
            rv[k] = (q < (64 + 2 * MD) * C4 && xs >= 0 && xs < W) ? *reinterpret_cast<const float4*>(Rrow + (long)xs * C + (q & 7) * 4) : make_float4(0, 0, 0, 0);
        }
        __syncthreads();          // previous iteration's readers are done
#pragma unroll
        for (int k = 0; k < 3; ++k) { const int q = tid + 256 * k; if (q < (64 + 2 * MD) * C4) Rs[q] = rv[k]; }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int px = u * 32 + (tid >> 3);
            float a[D];
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const float4 r = Rs[(px + j) * C4 + sub];
                a[j] = l[u].x * r.x + l[u].y * r.y + l[u].z * r.z + l[u].w * r.w;
                a[j] += __shfl_xor(a[j], 4); a[j] += __shfl_xor(a[j], 2); a[j] += __shfl_xor(a[j], 1);
            }
            if (sub < D) {
                float v = a[0];
#pragma unroll
                for (int j = 1; j < D; ++j) v = (sub == j) ? a[j] : v;
                out[((long)row * W + x0 + px) * D + sub] = v * (1.0f / C);
            }
        }
    }
}

int main() {
    const int B = 64, H = 96, W = 320;
    const long npix = (long)B * H * W;
    float *L, *R, *out;
    CK(hipMalloc(&L, npix * C * 4)); CK(hipMalloc(&R, npix * C * 4)); CK(hipMalloc(&out, npix * D * 4));
    CK(hipMemset(L, 0, npix * C * 4)); CK(hipMemset(R, 0, npix * C * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes_in = 2.0 * npix * C * 4, bytes_all = bytes_in + npix * D * 4.0;
    auto run = [&](const char* name, double bytes, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 10; ++i) launch();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
        printf("%-34s %8.1f us  %7.0f GB/s (%.1f%% of 8 TB/s)\n", name, ms * 1e3, bytes / ms * 1e-6, bytes / ms * 1e-6 / 80.0);
    };
    for (int g : {2048, 4096, 8192, 16384}) {
        char nm[64];
        snprintf(nm, 64, "v0 stream grid %d", g);
        run(nm, bytes_in, [&] { hipLaunchKernelGGL(v0_stream, dim3(g), dim3(256), 0, 0, (const float4*)L, (const float4*)R, out, npix * C4); });
    }
    for (int g : {4096, 8192, 16384, 61440}) {
        char nm[64];
        snprintf(nm, 64, "v1 direct u1 grid %d", g);
        run(nm, bytes_all, [&] { hipLaunchKernelGGL(v1_direct<1>, dim3(g), dim3(256), 0, 0, L, R, out, (int)npix, W); });
        snprintf(nm, 64, "v1 direct u2 grid %d", g);
        run(nm, bytes_all, [&] { hipLaunchKernelGGL(v1_direct<2>, dim3(g), dim3(256), 0, 0, L, R, out, (int)npix, W); });
    }
    for (int g : {2048, 4096, 8192, 30720}) {
        char nm[64];
        snprintf(nm, 64, "v2 lds window grid %d", g);
        run(nm, bytes_all, [&] { hipLaunchKernelGGL(v2_lds, dim3(g), dim3(256), 0, 0, L, R, out, B * H, W); });
    }
    return 0;
}
