// Dependent-load latency under a full chip as a function of the working set's size and page locality (address translation reach):
// every workgroup (one wavefront) walks a chain of HOPS dependent 16-byte loads; the chain's addresses are random inside a window of
// W bytes (one load per 4 KiB-aligned slot, so cache lines are never reused).  hipcc --offload-arch=gfx950 -O3 tlb_latency_micro.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(64) void k_chase(const unsigned long long* __restrict__ buf, const unsigned* __restrict__ start, int hops,
                                              unsigned long long* __restrict__ out, long long* __restrict__ cyc) {
    unsigned long long idx = start[blockIdx.x];
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int h = 0; h < hops; ++h) idx = buf[idx * 512 + (threadIdx.x & 1)];     // slot = 4 KiB = 512 x 8 B; lanes read 16 B of the slot
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        out[blockIdx.x] = idx;
        cyc[blockIdx.x] = t1 - t0;
    }
}

int main(int argc, char** argv) {
    const int hops = 32, nwg = argc > 1 ? atoi(argv[1]) : 4096;
    const size_t maxW = (size_t)24 << 30;
    unsigned long long* buf;
    CHK(hipMalloc(&buf, maxW));
    unsigned *d_start;
    unsigned long long* d_out;
    long long* d_cyc;
    CHK(hipMalloc(&d_start, nwg * 4));
    CHK(hipMalloc(&d_out, nwg * 8));
    CHK(hipMalloc(&d_cyc, nwg * 8));
    std::mt19937_64 rng(1);
    for (size_t W : {(size_t)1 << 30, (size_t)4 << 30, (size_t)12 << 30, (size_t)24 << 30}) {
        const size_t slots = W / 4096;
        // chains: nwg disjoint random chains of `hops` slots each (slots >= nwg * hops for every W used here)
        std::vector<unsigned> perm(slots);
        for (size_t i = 0; i < slots; ++i) perm[i] = (unsigned)i;
        for (size_t i = 0; i < (size_t)nwg * (hops + 1) && i < slots; ++i) std::swap(perm[i], perm[i + rng() % (slots - i)]);
        std::vector<unsigned> start(nwg);
        std::vector<unsigned long long> nxt(2);
        for (int w = 0; w < nwg; ++w) {
            start[w] = perm[(size_t)w * (hops + 1)];
            for (int h = 0; h < hops; ++h) {
                const unsigned long long from = perm[(size_t)w * (hops + 1) + h], to = perm[(size_t)w * (hops + 1) + h + 1];
                nxt[0] = nxt[1] = to;
                CHK(hipMemcpy(buf + from * 512, nxt.data(), 16, hipMemcpyHostToDevice));
            }
        }
        CHK(hipMemcpy(d_start, start.data(), nwg * 4, hipMemcpyHostToDevice));
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0));
        CHK(hipEventCreate(&e1));
        float ms = 0.f;
        for (int rep = 0; rep < 3; ++rep) {
            CHK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_chase, dim3(nwg), dim3(64), 0, 0, buf, d_start, hops, d_out, d_cyc);
            CHK(hipEventRecord(e1, 0));
            CHK(hipDeviceSynchronize());
            CHK(hipEventElapsedTime(&ms, e0, e1));
        }
        std::vector<long long> cyc(nwg);
        CHK(hipMemcpy(cyc.data(), d_cyc, nwg * 8, hipMemcpyDeviceToHost));
        double sum = 0;
        for (long long c : cyc) sum += (double)c;
        printf("window %6zu MiB, %d workgroups x %d hops: %.0f s_memtime ticks per dependent load (mean), kernel %.1f us = %.0f ns per hop\n", W >> 20, nwg, hops, sum / nwg / hops, 1e3 * ms, 1e6 * ms / hops);
    }
    return 0;
}
