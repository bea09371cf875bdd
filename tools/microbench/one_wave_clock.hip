// one_wave_clock.hip — what clock does a ONE-wave kernel run at?  (the LDS-resident ALS sweep is one wave on one CU)
// Core-clock ticks (s_memtime) against the 100 MHz constant clock (s_memrealtime) over a few milliseconds of a dependent
// fp64 chain, for 1 wave and for a full grid.   hipcc --offload-arch=gfx950 -O3 one_wave_clock.hip -o /tmp/one_wave_clock
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void spin(double *out, uint64_t *ticks, int iters) {
    double x = 1.0 + threadIdx.x * 1e-9;
    const uint64_t t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) x = x * 1.0000001 + 1e-12;      // dependent fp64 fma chain
    const uint64_t t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { ticks[0] = t1 - t0; ticks[1] = r1 - r0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

int main() {
    double *out;
    uint64_t *ticks, h[2];
    hipMalloc(&out, 1024 * 1024 * sizeof(double));
    hipMalloc(&ticks, 2 * sizeof(uint64_t));
    const int iters = 400000;
    for (int rep = 0; rep < 3; ++rep)
        for (int blocks : {1, 1024}) {
            hipLaunchKernelGGL(spin, dim3(blocks), dim3(blocks == 1 ? 64 : 256), 0, 0, out, ticks, iters);
            hipDeviceSynchronize();
            hipMemcpy(h, ticks, sizeof h, hipMemcpyDeviceToHost);
            const double us = h[1] / 100.0;
            printf("blocks %4d: %.0f us, %.0f core ticks -> %.0f MHz, %.2f ns and %.2f ticks per dependent fp64 fma\n", blocks, us, (double)h[0],
                   h[0] / us, us * 1e3 / iters, (double)h[0] / iters);
        }
    return 0;
}
