// grid_barrier_probe.hip -- what does a device-scope barrier between the workgroups of ONE launch cost on the MI355X, fences included?
// (round 5: the price of running a chain of small layers in one launch instead of one launch per layer.)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gbp scripts/exp/grid_barrier_probe.hip && /tmp/gbp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void grid_barrier(unsigned long long* ctr, unsigned nwg) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long target = (t / nwg + 1) * nwg;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        __atomic_thread_fence(__ATOMIC_ACQUIRE);      // agent scope (HIP's default for the C11 builtin is system; see the variant below)
    }
    __syncthreads();
}
__device__ __forceinline__ void grid_barrier_agent(unsigned long long* ctr, unsigned nwg) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long target = (t / nwg + 1) * nwg;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

template <int VAR>
__global__ __launch_bounds__(1024) void k(unsigned long long* ctr, float* buf, int* bad, int rounds, int per) {
    const unsigned nwg = gridDim.x;
    const int tid = threadIdx.x;
    int nb = 0;
    for (int r = 0; r < rounds; ++r) {
        float* mine = buf + ((size_t)(r & 1) * nwg + blockIdx.x) * per;
        for (int i = tid; i < per; i += 1024) mine[i] = (float)(r * 7 + blockIdx.x);
        if (VAR == 0) grid_barrier(ctr, nwg); else grid_barrier_agent(ctr, nwg);
        const unsigned o = (blockIdx.x + 1 + (r % (nwg > 1 ? nwg - 1 : 1))) % nwg;
        const float* theirs = buf + ((size_t)(r & 1) * nwg + o) * per;
        for (int i = tid; i < per; i += 1024) nb += theirs[i] != (float)(r * 7 + o);
    }
    if (nb) atomicAdd(bad, nb);
}
__global__ __launch_bounds__(1024) void knull(float* buf, int per) {
    float* mine = buf + (size_t)blockIdx.x * per;
    for (int i = threadIdx.x; i < per; i += 1024) mine[i] = 1.f;
}

int main() {
    unsigned long long* ctr; float* buf; int* bad;
    hipMalloc(&ctr, 64); hipMemset(ctr, 0, 64);
    hipMalloc(&buf, 2 * 512 * 16384 * 4); hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int R = 400;
    setvbuf(stdout, nullptr, _IONBF, 0);
    for (int var = 0; var < 2; ++var)
        for (int per : {1024, 8192})
            for (int nwg : {1, 6, 24, 72, 144, 240}) {
                float best = 1e9f;
                for (int rep = 0; rep < 5; ++rep) {
                    hipMemset(ctr, 0, 64);      // (the generation arithmetic needs a count that is a multiple of THIS grid)
                    hipEventRecord(e0);
                    if (var == 0) hipLaunchKernelGGL(k<0>, dim3(nwg), dim3(1024), 0, 0, ctr, buf, bad, R, per);
                    else hipLaunchKernelGGL(k<1>, dim3(nwg), dim3(1024), 0, 0, ctr, buf, bad, R, per);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                int hb; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost); hipMemset(bad, 0, 4);
                printf("fence %-6s  %3d workgroups x 1024 thr, %5d floats/wg/round: %.2f us per round (write + barrier + read), stale reads %d\n",
                       var ? "agent" : "c11", nwg, per, best * 1000.f / R, hb);
            }
    // the alternative: one launch per round (back to back on one stream)
    for (int nwg : {24, 72, 240}) {
        hipEventRecord(e0);
        for (int r = 0; r < R; ++r) hipLaunchKernelGGL(knull, dim3(nwg), dim3(1024), 0, 0, buf, 1024);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("launch per round, %3d workgroups: %.2f us per launch (stream, not a graph)\n", nwg, ms * 1000.f / R);
    }
    return 0;
}
