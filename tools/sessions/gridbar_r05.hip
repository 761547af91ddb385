// Round 5, verdict item 4 (feasibility): what does a grid-wide barrier cost on an MI355X for a grid shaped like the level-1 passes of the cycle
// (2900-3072 workgroups of one wavefront, all co-resident at three waves per SIMD)?  cooperative_groups grid.sync() against a hand-written
// sense-reversal barrier (one atomic per workgroup, spin on a generation word).  hipcc --offload-arch=gfx950 -O3 -o ab/gridbar gridbar_r05.hip
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <cstdlib>
namespace cg = cooperative_groups;

__global__ __launch_bounds__(64) void k_cg(int n, double *v) {
    cg::grid_group g = cg::this_grid();
    double a = v[blockIdx.x];
    for (int i = 0; i < n; i++) {
        a = a * 1.0000001 + 1.0;
        g.sync();
    }
    v[blockIdx.x] = a;
}

// sense-reversal barrier: count[0] = arrivals of the current generation, count[1] = generation
__device__ __forceinline__ void grid_barrier(unsigned *count, unsigned nblocks, unsigned &gen) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned target = gen + 1;
        if (atomicAdd(&count[0], 1u) == nblocks - 1) {
            count[0] = 0;
            __threadfence();
            atomicExch(&count[1], target);
        } else {
            while (__atomic_load_n(&count[1], __ATOMIC_ACQUIRE) != target) __builtin_amdgcn_s_sleep(1);
        }
        __threadfence();
    }
    gen++;
    __syncthreads();
}
__global__ __launch_bounds__(64) void k_own(int n, double *v, unsigned *count) {
    unsigned gen = 0;
    double a = v[blockIdx.x];
    for (int i = 0; i < n; i++) {
        a = a * 1.0000001 + 1.0;
        grid_barrier(count, gridDim.x, gen);
    }
    v[blockIdx.x] = a;
}

int main(int argc, char **argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 2900, N = 200;
    double *v; unsigned *cnt;
    hipMalloc(&v, sizeof(double) * G); hipMemset(v, 0, sizeof(double) * G);
    hipMalloc(&cnt, 8); hipMemset(cnt, 0, 8);
    int nb = 0, dev = 0, ncu = 0;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_own, 64, 0);
    printf("CUs %d, co-resident 64-thread workgroups per CU for k_own %d -> %d in all; grid %d\n", ncu, nb, nb * ncu, G);
    if (G > nb * ncu) { printf("grid does not fit\n"); return 1; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int rep = 0; rep < 3; rep++) {
        int n = N; void *args[] = {&n, &v};
        hipEventRecord(e0);
        hipError_t e = hipLaunchCooperativeKernel((const void *)k_cg, dim3(G), dim3(64), args, 0, 0);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("cooperative_groups grid.sync: %s, %.2f us per barrier\n", hipGetErrorString(e), ms * 1e3 / N);
        hipMemset(cnt, 0, 8);
        void *args2[] = {&n, &v, &cnt};
        hipEventRecord(e0);
        e = hipLaunchCooperativeKernel((const void *)k_own, dim3(G), dim3(64), args2, 0, 0);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("own barrier (cooperative launch): %s, %.2f us per barrier\n", hipGetErrorString(e), ms * 1e3 / N);
        // empty-kernel launch cost for comparison: N dependent launches of a trivial kernel
        hipEventRecord(e0);
        for (int i = 0; i < N; i++) { int one = 0; hipLaunchKernelGGL(k_own, dim3(G), dim3(64), 0, 0, one, v, cnt); }
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("%d dependent launches of an empty kernel of that grid: %.2f us each\n", N, ms * 1e3 / N);
    }
    return 0;
}
