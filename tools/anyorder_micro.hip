// Does hipExtAnyOrderLaunch let two independent kernels of one stream overlap on gfx950?  (tools/anyorder_micro.hip)
// Level l = kernel A (ordinary launch: waits for everything before it) + kernel B (any-order: may start while A runs).
// Each kernel spins ~20 us on the constant-rate clock with a handful of workgroups: if the flag is honoured the 100-level chain
// takes ~half the time.  A dependency check follows: B(l) increments what A(l+1) reads -- A(l+1) must still see every B(l).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_spin(long long ticks, int* cnt, int* seen, int expect) {
    if (seen && threadIdx.x == 0) {
        const int v = __atomic_load_n(cnt, __ATOMIC_RELAXED);
        if (v < expect) atomicAdd(seen, 1);                 // a predecessor had not finished
    }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(cnt, 1);
}

int main() {
    int *cnt, *bad;
    CHK(hipMalloc(&cnt, 8));
    CHK(hipMalloc(&bad, 4));
    hipStream_t st;
    CHK(hipStreamCreate(&st));
    const int WG = 16, LEVELS = 200;
    const long long ticks = 2000;                           // 100 MHz clock: 20 us
    for (int mode = 0; mode < 3; ++mode) {                  // 0: both ordinary, 1: B any-order, 2: warm repeat of 1
        CHK(hipMemsetAsync(cnt, 0, 8, st));
        CHK(hipMemsetAsync(bad, 0, 4, st));
        CHK(hipStreamSynchronize(st));
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0));
        CHK(hipEventCreate(&e1));
        CHK(hipEventRecord(e0, st));
        for (int l = 0; l < LEVELS; ++l) {
            // A(l): must see all 2 * WG * l increments of the levels before
            hipExtLaunchKernelGGL(k_spin, dim3(WG), dim3(256), 0, st, nullptr, nullptr, 0, ticks, cnt, bad, 2 * WG * l);
            hipExtLaunchKernelGGL(k_spin, dim3(WG), dim3(256), 0, st, nullptr, nullptr, mode ? hipExtAnyOrderLaunch : 0, ticks, cnt,
                                  bad, 2 * WG * l);
        }
        CHK(hipEventRecord(e1, st));
        CHK(hipStreamSynchronize(st));
        float ms = 0.f;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        int hb = 0, hc = 0;
        CHK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(&hc, cnt, 4, hipMemcpyDeviceToHost));
        printf("mode %d: %d levels of 2 kernels: %.3f ms (%.1f us per level), ordering violations %d, count %d\n", mode, LEVELS, ms,
               1e3 * ms / LEVELS, hb, hc);
    }
    return 0;
}
