// Micro-benchmark: what do scalar and branch instructions cost next to vector instructions on one SIMD?
// Every wave runs 64 dependent v_mul/v_add per loop trip plus S scalar ALU instructions and B taken-or-not branches,
// at 8 waves per SIMD.  If scalar work were issued for free from other waves, time would not depend on S or B.
// usage: issue_mix
#include <hip/hip_runtime.h>

#include <cstdio>

template <int S, int B>
__global__ __launch_bounds__(256) void k_mix(float* out, int iters) {
    float x = (float)threadIdx.x * 1e-3f, y = 1.0001f, z = 0.5f;
    unsigned s = blockIdx.x;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            x = x * y;
            x = x + z;
            z = z * 0.9999f;
            y = y + 1e-7f;
#pragma unroll
            for (int j = 0; j < S / 16; j++) asm volatile("s_add_u32 %0, %0, 0x12345" : "+s"(s) : : "scc");
#pragma unroll
            for (int j = 0; j < B / 16; j++) asm volatile("s_cmp_eq_u32 %0, 0xffffffff\n s_cbranch_scc1 Lskip%=\n s_nop 0\nLskip%=:" : : "s"(s) : "scc");
        }
    }
    if (x == 123.456f) out[threadIdx.x] = x + y + z + (float)s;
}

template <int S, int B>
static void run(float* d, const char* what) {
    const int iters = 2000, blocks = 256 * 8;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL((k_mix<S, B>), dim3(blocks), dim3(256), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((k_mix<S, B>), dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double trips = 8.0 * iters;  // loop trips per SIMD
    std::printf("%-40s %8.3f ms  %.1f ns per trip of 64 VALU per SIMD (%.3f ns per VALU)\n", what, ms, ms * 1e6 / trips, ms * 1e6 / trips / 64.0);
}

int main() {
    float* d = nullptr;
    hipMalloc(&d, 4096);
    run<0, 0>(d, "64 VALU");
    run<16, 0>(d, "64 VALU + 16 SALU");
    run<32, 0>(d, "64 VALU + 32 SALU");
    run<64, 0>(d, "64 VALU + 64 SALU");
    run<128, 0>(d, "64 VALU + 128 SALU");
    run<0, 16>(d, "64 VALU + 16 x (cmp, branch, nop)");
    run<0, 32>(d, "64 VALU + 32 x (cmp, branch, nop)");
    run<32, 16>(d, "64 VALU + 32 SALU + 16 x (cmp, branch, nop)");
    hipFree(d);
    return 0;
}
