// Dependent-issue latencies of the instructions on the Gauss-Jordan sub-step's critical path (csrc/hpf_quad.hpp, section C), in
// shader-clock cycles (s_memtime), for ONE wavefront on an otherwise idle CU and for the same wavefront with `bg` other waves of the
// same kind on its SIMD.      hipcc --offload-arch=gfx950 -O3 -ffp-contract=off lat_micro.hip -o lat_micro ; ./lat_micro
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double d4_t __attribute__((ext_vector_type(4)));
constexpr int N = 256;

__device__ __forceinline__ double rcp_nr(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    e = fma(-x, r, 1.0);
    return fma(r, e, r);
}

__global__ void __launch_bounds__(64) k_lat(double* out, long long* cyc, double seed) {
    __shared__ double lds[1024];
    const int lane = threadIdx.x & 63;
    long long t0, t1;
    double x = seed + lane * 1e-3, y = 1.0000001;
    int slot = 0;
    // 1. v_fma_f64 chain
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < N; ++i) x = fma(x, y, 1e-9);
    asm volatile("" : "+v"(x));
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) cyc[slot] = t1 - t0;
    ++slot;
    // 2. v_mul_f64 chain
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < N; ++i) x = x * y;
    asm volatile("" : "+v"(x));
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) cyc[slot] = t1 - t0;
    ++slot;
    // 3. v_rcp_f64 chain
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < N; ++i) x = __builtin_amdgcn_rcp(x);
    asm volatile("" : "+v"(x));
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) cyc[slot] = t1 - t0;
    ++slot;
    // 4. rcp + two Newton steps
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < N; ++i) x = rcp_nr(x);
    asm volatile("" : "+v"(x));
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) cyc[slot] = t1 - t0;
    ++slot;
    // 5. quad reduction step: two v_mov_dpp + v_add_f64
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        int lo = __double2loint(x), hi = __double2hiint(x);
        x += __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xf, 0xf, false));
    }
    asm volatile("" : "+v"(x));
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) cyc[slot] = t1 - t0;
    ++slot;
    // 6. LDS round trip: ds_write_b64 -> s_waitcnt -> ds_read_b64 (other lane's value) -> s_waitcnt
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        lds[lane] = x;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        x = lds[lane ^ 5];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("" : "+v"(x));
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) cyc[slot] = t1 - t0;
    ++slot;
    // 7. v_mfma_f64_16x16x4 chain through the accumulator
    d4_t acc = {x, x, x, x};
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < N; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, 1e-3, acc, 0, 0, 0);
    asm volatile("" : "+v"(acc));
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) cyc[slot] = t1 - t0;
    ++slot;
    // 8. v_mfma_f64_16x16x4 chain through the B operand (result -> VALU register -> next B)
    double b = x;
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const d4_t z = {0.0, 0.0, 0.0, 0.0};
        const d4_t d = __builtin_amdgcn_mfma_f64_16x16x4f64(y, b, z, 0, 0, 0);
        b = d[0];
    }
    asm volatile("" : "+v"(b));
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) cyc[slot] = t1 - t0;
    ++slot;
    // 9. ds_bpermute_b32 pair (one f64) chain
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        int lo = __double2loint(b), hi = __double2hiint(b);
        lo = __builtin_amdgcn_ds_bpermute((lane ^ 5) * 4, lo);
        hi = __builtin_amdgcn_ds_bpermute((lane ^ 5) * 4, hi);
        b = __hiloint2double(hi, lo);
    }
    asm volatile("" : "+v"(b));
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) cyc[slot] = t1 - t0;
    ++slot;
    // 10. v_readlane pair + v_mov (broadcast of one f64 through SGPRs)
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        int lo = __double2loint(b), hi = __double2hiint(b);
        lo = __builtin_amdgcn_readlane(lo, 5);
        hi = __builtin_amdgcn_readlane(hi, 5);
        b = __hiloint2double(hi, lo) + (double)lane;
    }
    asm volatile("" : "+v"(b));
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) cyc[slot] = t1 - t0;
    ++slot;
    // 11. s_memtime back to back (cost of a stamp)
    t0 = __builtin_amdgcn_s_memtime();
    long long tt = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) tt += __builtin_amdgcn_s_memtime();
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) cyc[slot] = (t1 - t0) * (N / 16);
    ++slot;
    out[blockIdx.x * 64 + lane] = x + acc[0] + acc[3] + b + (double)(tt & 1);
}

int main(int argc, char** argv) {
    double* out; long long* cyc;
    hipMalloc(&out, sizeof(double) * 64 * 8192);
    hipMalloc(&cyc, sizeof(long long) * 16);
    const char* names[] = {"v_fma_f64", "v_mul_f64", "v_rcp_f64", "rcp + 2 Newton steps (5 ops)", "2 x v_mov_dpp + v_add_f64", "LDS write->wait->read->wait",
                           "mfma_f64_16x16x4 (acc chain)", "mfma_f64_16x16x4 -> B operand", "2 x ds_bpermute_b32", "2 x v_readlane + v_add_f64", "s_memtime"};
    // grid sizes: 1 wave on the chip; 256 CUs x 4 SIMDs x w waves (blocks of one wave are dealt round-robin over the CUs / SIMDs)
    const int grids[] = {1, 1024, 2048, 4096};
    for (int g : grids) {
        long long h[16];
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k_lat, dim3(g), dim3(64), 0, 0, out, cyc, 1.5);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        printf("grid %5d single-wave workgroups (~%d per SIMD): cycles per dependent step\n", g, g <= 1024 ? 1 : g / 1024);
        for (int i = 0; i < 11; ++i) printf("    %-32s %7.1f\n", names[i], (double)h[i] / N);
    }
    return 0;
}
