// probe: exact lane/element mapping of ds_read_b64_tr_b16 on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void probe(const int* addr_bytes, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int a = addr_bytes[threadIdx.x];
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)((__attribute__((address_space(3))) char*)lds + a));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)r[j];
}
int main() {
    int* d_a; unsigned short* d_o;
    hipMalloc(&d_a, 64 * 4); hipMalloc(&d_o, 64 * 4 * 2);
    for (int pat = 0; pat < 3; ++pat) {
        std::vector<int> a(64);
        for (int l = 0; l < 64; ++l) a[l] = pat == 0 ? l * 8 : pat == 1 ? l * 80 : ((l & 15) * 264 + (l >> 4) * 8);
        hipMemcpy(d_a, a.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_a, d_o);
        std::vector<unsigned short> o(256);
        hipMemcpy(o.data(), d_o, 512, hipMemcpyDeviceToHost);
        printf("pattern %d (lane l address bytes: %s)\n", pat, pat == 0 ? "8*l" : pat == 1 ? "80*l" : "(l&15)*264 + (l>>4)*8");
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d:", l);
            for (int j = 0; j < 4; ++j) {
                // decode: which lane's address range does the half index fall into, and which element
                int h = o[l * 4 + j], src = -1, el = -1;
                for (int s = 0; s < 64; ++s) if (h * 2 >= a[s] && h * 2 < a[s] + 8) { src = s; el = (h * 2 - a[s]) / 2; }
                printf("  %4d(l%02d.e%d)", h, src, el);
            }
            printf("\n");
        }
    }
    return 0;
}
