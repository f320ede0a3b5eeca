// Go / no-go probe for a dataflow factor sweep (DESIGN.md 5a "upper levels"): a chain of L dependent levels of W workgroups, each workgroup
// (256 threads) reads the 23 KB tile its predecessor of the previous level wrote, "works" for about T us and writes its own 23 KB.
//   (a) one launch per level (what k_level does today);
//   (b) ONE launch, workgroups ordered by level, a workgroup waits for ITS predecessor's ready flag (agent-scope release / acquire,
//       bounded spin with s_sleep) -- the XCD L2s are not coherent for plain stores, so the fences carry cache write-back / invalidate.
// Prints the wall time of both and the checksum (must agree).     hipcc --offload-arch=gfx950 -O3 dataflow_micro.hip -o dataflow_micro
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
constexpr int TILE = 23 * 128;      // doubles (23.5 KB)

__device__ __forceinline__ void body(const double* in, double* out, int work) {
    double acc[3];
    for (int j = 0; j < 3; ++j) acc[j] = 0.0;
    for (int i = threadIdx.x; i < TILE; i += 256) acc[(i / 256) % 3] += in ? in[i] : 1.0;
    double x = acc[0] + acc[1] + acc[2];
    for (int k = 0; k < work; ++k) x = fma(x, 1.0000001, 1e-9);      // dependent chain: ~work * 8 cycles
    for (int i = threadIdx.x; i < TILE; i += 256) out[i] = x * 1e-3 + (double)i * 1e-9;
}

__global__ void k_level_launch(const double* in, double* out, int W, int work) {
    const int w = blockIdx.x;
    body(in ? in + (size_t)w * TILE : nullptr, out + (size_t)w * TILE, work);
}

__global__ void k_dataflow(double* buf, int* flags, int W, int L, int work, int epoch, int* timeout) {
    const int l = blockIdx.x / W, w = blockIdx.x % W;
    __shared__ int ok;
    if (l > 0) {
        if (threadIdx.x == 0) {
            int spins = 0, got = 0;
            while (spins < 4000000) {
                if (__hip_atomic_load(flags + (size_t)(l - 1) * W + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch) { got = 1; break; }
                __builtin_amdgcn_s_sleep(2);
                ++spins;
            }
            if (!got) atomicAdd(timeout, 1);
            ok = got;
        }
        __syncthreads();
        __atomic_thread_fence(__ATOMIC_ACQUIRE);        // agent scope by default for HIP device code: invalidates non-coherent cache lines
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    body(l > 0 ? buf + ((size_t)(l - 1) * W + w) * TILE : nullptr, buf + ((size_t)l * W + w) * TILE, work);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // every thread: its stores are visible at agent scope before the flag
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flags + (size_t)l * W + w, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

int main(int argc, char** argv) {
    const int W = argc > 1 ? atoi(argv[1]) : 48, L = argc > 2 ? atoi(argv[2]) : 10, work = argc > 3 ? atoi(argv[3]) : 4000, reps = 20;
    double* buf; int *flags, *timeout;
    hipMalloc(&buf, sizeof(double) * (size_t)L * W * TILE);
    hipMalloc(&flags, sizeof(int) * (size_t)L * W);
    hipMalloc(&timeout, sizeof(int));
    hipMemset(flags, 0, sizeof(int) * (size_t)L * W);
    hipMemset(timeout, 0, sizeof(int));
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<double> last(TILE);
    float ms_a = 0, ms_b = 0;
    for (int pass = 0; pass < 2; ++pass) {               // pass 0 warms up
        hipEventRecord(e0, st);
        for (int r = 0; r < reps; ++r)
            for (int l = 0; l < L; ++l)
                hipLaunchKernelGGL(k_level_launch, dim3(W), dim3(256), 0, st, l ? buf + (size_t)(l - 1) * W * TILE : nullptr, buf + (size_t)l * W * TILE, W, work);
        hipEventRecord(e1, st); hipEventSynchronize(e1); hipEventElapsedTime(&ms_a, e0, e1);
    }
    hipMemcpy(last.data(), buf + ((size_t)(L - 1) * W + (W - 1)) * TILE, sizeof(double) * TILE, hipMemcpyDeviceToHost);
    double ca = 0; for (double v : last) ca += v;
    hipMemset(buf, 0, sizeof(double) * (size_t)L * W * TILE);
    int epoch = 0;
    for (int pass = 0; pass < 2; ++pass) {
        hipEventRecord(e0, st);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_dataflow, dim3(W * L), dim3(256), 0, st, buf, flags, W, L, work, ++epoch, timeout);
        hipEventRecord(e1, st); hipEventSynchronize(e1); hipEventElapsedTime(&ms_b, e0, e1);
    }
    hipMemcpy(last.data(), buf + ((size_t)(L - 1) * W + (W - 1)) * TILE, sizeof(double) * TILE, hipMemcpyDeviceToHost);
    double cb = 0; for (double v : last) cb += v;
    int to = 0; hipMemcpy(&to, timeout, sizeof(int), hipMemcpyDeviceToHost);
    printf("W=%d L=%d work=%d: per sweep  launches %.1f us   dataflow %.1f us   (checksums %.6f %.6f %s, timeouts %d)\n", W, L, work,
           1e3 * ms_a / reps, 1e3 * ms_b / reps, ca, cb, ca == cb ? "equal" : "DIFFER", to);
    return 0;
}
