// Known-bytes calibration of the rocprofv3 HBM counters (FETCH_SIZE / WRITE_SIZE) on the access patterns of the block-tree
// kernels.  MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reads 1/2 of a wide 16 B/lane streaming read, WRITE_SIZE is exact for
// 16 B/lane streaming stores, "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern".
// Every kernel here moves an exactly known number of bytes, far beyond the 256 MiB Infinity Cache, in one pattern:
//   k_load16 / k_store16   16 B per lane, 1 KB contiguous per wave instruction (the paired row groups of a tile image)
//   k_load8  / k_store8     8 B per lane, 512 B contiguous per wave instruction (the unpaired row group, w / x / f images)
//   k_tile_load / k_tile_store   the tile image of one 52 x 52 block exactly as TileIO<52> lays it out (4 waves, 6 x 16 B + 1 x 8 B
//                                per lane, last tile column 8 wide): 23 296 B per block image
// Build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o tools/bin/pmc_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/calib_FETCH_SIZE -- tools/bin/pmc_calib
//   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/calib_WRITE_SIZE -- tools/bin/pmc_calib
// then tools/pmc_calib_report.py.  The program prints the known byte count of every launch.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHK(x)                                                                   \
    do {                                                                         \
        hipError_t e_ = (x);                                                     \
        if (e_ != hipSuccess) {                                                  \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));              \
            exit(1);                                                             \
        }                                                                        \
    } while (0)

__global__ __launch_bounds__(256) void k_load16(const double2* __restrict__ p, size_t n, double* __restrict__ sink) {
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double2 v = p[i];
        acc += v.x + v.y;
    }
    if (acc == 1.2345e300) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_load8(const double* __restrict__ p, size_t n, double* __restrict__ sink) {
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += p[i];
    if (acc == 1.2345e300) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_store16(double2* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = double2{1.0, 2.0};
}
__global__ __launch_bounds__(256) void k_store8(double* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 3.0;
}

// TileIO<52>: NT = 4 waves, NE = 13 row groups = 6 pairs + 1; row group = (NT-1)*64 + 4*8 = 224 doubles
constexpr int TB_NT = 4, TB_LW = 8, TB_RG = (TB_NT - 1) * 64 + 4 * TB_LW, TB_NP = 6;
constexpr size_t TILE_DOUBLES = (size_t)(2 * TB_NP + 1) * TB_RG;     // 2 912 doubles = 23 296 B
__device__ __forceinline__ size_t t_off2(int p, int wv, int lg, int jj) {
    return (size_t)p * 2 * TB_RG + (wv < TB_NT - 1 ? wv * 128 + (lg * 16 + jj) * 2 : (TB_NT - 1) * 128 + (lg * TB_LW + jj) * 2);
}
__device__ __forceinline__ size_t t_off1(int wv, int lg, int jj) {
    return (size_t)TB_NP * 2 * TB_RG + (wv < TB_NT - 1 ? wv * 64 + lg * 16 + jj : (TB_NT - 1) * 64 + lg * TB_LW + jj);
}
__global__ __launch_bounds__(256) void k_tile_load(const double* __restrict__ base, size_t nblk, double* __restrict__ sink) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, lg = lane >> 4, jj = lane & 15;
    const bool in = wv < TB_NT - 1 || jj < TB_LW;
    double acc = 0.0;
    for (size_t b = blockIdx.x; b < nblk; b += gridDim.x) {
        const double* t = base + b * TILE_DOUBLES;
        if (in) {
#pragma unroll
            for (int p = 0; p < TB_NP; ++p) {
                const double2 v = *reinterpret_cast<const double2*>(t + t_off2(p, wv, lg, jj));
                acc += v.x + v.y;
            }
            acc += t[t_off1(wv, lg, jj)];
        }
    }
    if (acc == 1.2345e300) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_tile_store(double* __restrict__ base, size_t nblk) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, lg = lane >> 4, jj = lane & 15;
    const bool in = wv < TB_NT - 1 || jj < TB_LW;
    for (size_t b = blockIdx.x; b < nblk; b += gridDim.x) {
        double* t = base + b * TILE_DOUBLES;
        if (in) {
#pragma unroll
            for (int p = 0; p < TB_NP; ++p) *reinterpret_cast<double2*>(t + t_off2(p, wv, lg, jj)) = double2{1.0, 2.0};
            t[t_off1(wv, lg, jj)] = 3.0;
        }
    }
}

int main() {
    const size_t BYTES = (size_t)2 << 30;                 // 2 GiB per launch: 8x the Infinity Cache
    double *buf = nullptr, *sink = nullptr;
    CHK(hipMalloc((void**)&buf, BYTES));
    CHK(hipMalloc((void**)&sink, 64));
    CHK(hipMemset(buf, 0, BYTES));
    const int grid = 256 * 8;
    const size_t nblk = BYTES / (TILE_DOUBLES * 8);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_load16, dim3(grid), dim3(256), 0, 0, (const double2*)buf, BYTES / 16, sink);
        hipLaunchKernelGGL(k_load8, dim3(grid), dim3(256), 0, 0, (const double*)buf, BYTES / 8, sink);
        hipLaunchKernelGGL(k_tile_load, dim3(grid), dim3(256), 0, 0, (const double*)buf, nblk, sink);
        hipLaunchKernelGGL(k_store16, dim3(grid), dim3(256), 0, 0, (double2*)buf, BYTES / 16);
        hipLaunchKernelGGL(k_store8, dim3(grid), dim3(256), 0, 0, buf, BYTES / 8);
        hipLaunchKernelGGL(k_tile_store, dim3(grid), dim3(256), 0, 0, buf, nblk);
        CHK(hipDeviceSynchronize());
    }
    printf("{\"k_load16\": %zu, \"k_load8\": %zu, \"k_tile_load\": %zu, \"k_store16\": %zu, \"k_store8\": %zu, \"k_tile_store\": %zu, \"launches_each\": 3}\n",
           BYTES, BYTES, nblk * TILE_DOUBLES * 8, BYTES, BYTES, nblk * TILE_DOUBLES * 8);
    CHK(hipFree(buf));
    CHK(hipFree(sink));
    return 0;
}
