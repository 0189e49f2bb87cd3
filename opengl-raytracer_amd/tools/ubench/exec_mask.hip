// Micro-benchmark: does a wave64 VALU instruction get cheaper when whole 16-lane groups of the wave are masked off?
// Every wave runs the same dependent chain of v_fma-free multiply/adds under one of several EXEC patterns.
// Also: issue rate against resident waves per SIMD, and the shader clock under that load.
// usage: exec_mask
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void k_chain(float* out, unsigned long long mask, int iters) {
    const int lane = threadIdx.x & 63;
    float x = (float)threadIdx.x * 1e-3f, y = 1.0001f, z = 0.5f;
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                x = x * y;
                x = x + z;
                z = z * 0.9999f;
                y = y + 1e-7f;
            }
        }
    }
    if (x == 123.456f) out[threadIdx.x] = x + y + z;  // never true: keeps the chain alive
}

// One wave per SIMD-slot reports shader-clock cycles (s_memtime) and 100 MHz wall ticks around the same chain.
__global__ __launch_bounds__(256) void k_clock(unsigned long long* t, int iters) {
    float x = (float)threadIdx.x * 1e-3f, y = 1.0001f, z = 0.5f;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            x = x * y;
            x = x + z;
            z = z * 0.9999f;
            y = y + 1e-7f;
        }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        t[0] = c1 - c0;
        t[1] = w1 - w0;
    }
    if (x == 123.456f) t[2] = (unsigned long long)(x + y + z);
}

int main() {
    float* d = nullptr;
    hipMalloc(&d, 4096);
    const int iters = 2000, blocks = 256 * 8;  // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    struct P { const char* name; unsigned long long m; } pats[] = {
        {"all 64 lanes", ~0ull},
        {"lower 32 lanes", 0xffffffffull},
        {"lower 16 lanes", 0xffffull},
        {"lanes 16..31", 0xffff0000ull},
        {"every 4th lane (16 lanes, all quarters)", 0x1111111111111111ull},
        {"every 2nd lane (32 lanes, all quarters)", 0x5555555555555555ull},
        {"one lane", 1ull},
        {"lanes 0..15 and 32..47", 0x0000ffff0000ffffull},
    };
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (auto& p : pats) {
        hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(256), 0, 0, d, p.m, 10);
        hipDeviceSynchronize();
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(256), 0, 0, d, p.m, iters);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        // VALU instructions per SIMD: 8 waves x iters x 64
        const double insts_per_simd = 8.0 * iters * 64.0;
        std::printf("%-44s %8.3f ms  %.3f ns per wave instruction per SIMD\n", p.name, ms, ms * 1e6 / insts_per_simd);
    }
    // issue rate against resident waves per SIMD (all lanes active)
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int nb = 256 * wps;
        hipLaunchKernelGGL(k_chain, dim3(nb), dim3(256), 0, 0, d, ~0ull, 10);
        hipDeviceSynchronize();
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(k_chain, dim3(nb), dim3(256), 0, 0, d, ~0ull, iters);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        std::printf("%d waves per SIMD: %8.3f ms  %.3f ns per wave instruction per SIMD\n", wps, ms, ms * 1e6 / (wps * iters * 64.0));
    }
    // shader clock under this load: s_memtime ticks against the constant 100 MHz counter
    unsigned long long* dt = nullptr;
    hipMalloc(&dt, 64);
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int it = 20000;
        hipLaunchKernelGGL(k_clock, dim3(256 * wps), dim3(256), 0, 0, dt, it);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(k_clock, dim3(256 * wps), dim3(256), 0, 0, dt, it);
        hipDeviceSynchronize();
        unsigned long long h[2];
        hipMemcpy(h, dt, 16, hipMemcpyDeviceToHost);
        // block 0 holds the oldest waves of its SIMDs, which the arbiter favours: this is the latency between two
        // dependent instructions of ONE wave, not a throughput
        const double ns = h[1] * 10.0, own = it * 64.0;
        std::printf("%d waves per SIMD, oldest wave: s_memtime %.3f GHz; %.3f ns = %.2f ticks between its dependent instructions\n",
                    wps, h[0] / ns, ns / own, h[0] / own);
    }
    hipFree(dt);
    hipFree(d);
    return 0;
}
