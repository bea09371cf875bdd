// ta_shapes.hip — how many cycles does the vector-memory path spend per wave-instruction, by shape?
// One 256-thread workgroup per CU x WG_PER_CU, every wave issues N loads of one shape back to back
// (addresses precomputed in registers / cheap LCG), tables L2-resident.  Prints ns and cycles per
// wave-instruction per CU.  Build: hipcc -O3 --offload-arch=gfx950 ta_shapes.hip -o ta_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITERS = 2048;

__device__ __forceinline__ uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s >> 8; }

// shape 0: 8 random 128-B rows per instr (dwordx4/lane)   [the V / P row gather at k=32]
// shape 1: 16 random 64-B rows per instr (dwordx4/lane)    [k=16]
// shape 2: 64 random dwords per instr                      [w / e gather]
// shape 3: 8 random dwords, each read by 8 lanes            [scalar per slot]
// shape 4: contiguous 256 B (dword/lane)                   [coalesced dword stream]
// shape 5: 8 x 32-B contiguous pieces 256 B apart (dword)  [the per-slot stream loads]
// shape 6: contiguous 1 KB (dwordx4/lane)
// shape 7: 8 random 128-B rows, only lane 0 of each 8 active (dwordx4) [one 16-B piece per row]
// shape 8: 4 random 256-B rows per instr (dwordx4/lane)    [k=64 half]
template <int SHAPE>
__global__ __launch_bounds__(256) void k(const float *tab, uint32_t rows_mask, float *out) {
    const int lane = threadIdx.x & 63;
    uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.f;
    for (int it = 0; it < ITERS; it += 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            uint32_t r = lcg(s);
            if (SHAPE == 0) {
                r = __shfl(r, lane & ~7, 64) & rows_mask;
                v[u] = reinterpret_cast<const float4 *>(tab + (size_t)r * 32)[lane & 7];
            } else if (SHAPE == 1) {
                r = __shfl(r, lane & ~3, 64) & rows_mask;
                v[u] = reinterpret_cast<const float4 *>(tab + (size_t)r * 16)[lane & 3];
            } else if (SHAPE == 2) {
                r &= rows_mask;
                v[u] = make_float4(tab[(size_t)r * 32], 0, 0, 0);
            } else if (SHAPE == 3) {
                r = __shfl(r, lane & ~7, 64) & rows_mask;
                v[u] = make_float4(tab[(size_t)r * 32], 0, 0, 0);
            } else if (SHAPE == 4) {
                r = __shfl(r, 0, 64) & rows_mask;
                v[u] = make_float4(tab[(size_t)r * 32 + lane], 0, 0, 0);
            } else if (SHAPE == 5) {
                r = __shfl(r, 0, 64) & (rows_mask >> 4);
                v[u] = make_float4(tab[(size_t)r * 512 + (lane >> 3) * 64 + (lane & 7)], 0, 0, 0);
            } else if (SHAPE == 6) {
                r = __shfl(r, 0, 64) & (rows_mask >> 3);
                v[u] = reinterpret_cast<const float4 *>(tab + (size_t)r * 256)[lane];
            } else if (SHAPE == 7) {
                r = __shfl(r, lane & ~7, 64) & rows_mask;
                v[u] = make_float4(0, 0, 0, 0);
                if ((lane & 7) == 0) v[u] = reinterpret_cast<const float4 *>(tab + (size_t)r * 32)[0];
            } else {
                r = __shfl(r, lane & ~15, 64) & (rows_mask >> 1);
                v[u] = reinterpret_cast<const float4 *>(tab + (size_t)r * 64)[lane & 15];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int SHAPE>
void run(const char *name, const float *tab, uint32_t mask, float *out, int wg_per_cu) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    dim3 g(256 * wg_per_cu), blk(256);
    hipLaunchKernelGGL(k<SHAPE>, g, blk, 0, 0, tab, mask, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<SHAPE>, g, blk, 0, 0, tab, mask, out);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double instr_per_cu = 5.0 * wg_per_cu * 4 * ITERS;     // wave-instructions issued on one CU
    const double ns = ms * 1e6 / instr_per_cu;
    printf("%-52s wg/cu=%d  %7.2f ns/wave-instr/CU  (~%5.1f cycles @2.1GHz)\n", name, wg_per_cu, ns, ns * 2.1);
}

int main(int argc, char **argv) {
    const size_t table_mb = argc > 1 ? atoi(argv[1]) : 2;       // table size: 2 MB (L2) by default
    const size_t floats = table_mb * 1024 * 1024 / 4;
    float *tab, *out;
    CHECK(hipMalloc(&tab, floats * 4 + 4096));
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(tab, 0, floats * 4 + 4096));
    const uint32_t mask = (uint32_t)(floats / 32) - 1;           // 128-B rows
    printf("table %zu MB, %u rows of 128 B\n", table_mb, mask + 1);
    for (int w : {2, 4, 8}) {
        run<0>("0: 8 random 128-B rows (dwordx4)          [k=32 row]", tab, mask, out, w);
        run<1>("1: 16 random 64-B rows (dwordx4)           [k=16 row]", tab, mask, out, w);
        run<8>("8: 4 random 256-B rows (dwordx4)           [k=64 row]", tab, mask, out, w);
        run<2>("2: 64 random dwords                        [w/e gather]", tab, mask, out, w);
        run<3>("3: 8 random dwords x 8 lanes each", tab, mask, out, w);
        run<7>("7: 8 random rows, 1 active lane each (dwordx4)", tab, mask, out, w);
        run<4>("4: contiguous 256 B (dword)", tab, mask, out, w);
        run<5>("5: 8 x 32 B pieces, 256 B apart (dword)   [slot streams]", tab, mask, out, w);
        run<6>("6: contiguous 1 KB (dwordx4)", tab, mask, out, w);
    }
    return 0;
}
