// Micro-benchmark: sustained rate of v_mfma_f32_32x32x2_f32 over ~100 ms with operands that change
// every instruction (pseudo-random bit patterns of normal floats), 2 waves per SIMD, 8 independent
// accumulators, 16 distinct random operand registers per lane -- the power-limited ceiling a real kernel sees, as opposed to the 1 ms burst rate.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_sustained.hip -o mfma_sustained
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int RANDOM>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    float av[16], bv[16];
    for (int i = 0; i < 16; ++i) {
        s = s * 1664525u + 1013904223u;
        av[i] = RANDOM ? __uint_as_float(0x3f000000u | (s >> 9)) * ((s & 256) ? -1.f : 1.f) : 1.f;
        bv[i] = RANDOM ? __uint_as_float(0x3f000000u | ((s * 747796405u) >> 9)) : 0.5f;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u * 8 + i], bv[(u * 8 + i * 5) & 15], acc[i], 0, 0, 0);
        // keep the accumulators bounded without touching the matrix pipe's schedule much
        if ((it & 1023) == 1023)
            for (int i = 0; i < 8; ++i)
                for (int r = 0; r < 16; ++r) acc[i][r] *= 1e-3f;
    }
    float t = 0;
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) t += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}

template <int RANDOM>
void run(int iters) {
    float* out;
    hipMalloc(&out, 512 * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 512;      // 2 blocks of 4 waves per CU
    k<RANDOM><<<grid, 256>>>(out, 100);
    hipEventRecord(e0);
    k<RANDOM><<<grid, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * 4 * iters * 16 * 4096.0;
    printf("random=%d iters=%d  %.2f ms  %.1f TFLOP/s\n", RANDOM, iters, ms, flop / ms / 1e9);
    hipFree(out);
}
int main() {
    run<0>(2000); run<0>(200000); run<1>(2000); run<1>(200000); run<1>(200000);
    return 0;
}
