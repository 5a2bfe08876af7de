// Micro-benchmark: issue rate of v_mfma_f32_32x32x2_f32 for different accumulator rotation
// depths and waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 mfma_f32_rate.hip -o mfma_f32_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(int blocks_per_cu) {
    int iters = 2000;
    float* out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = 256 * blocks_per_cu;
    k<NACC><<<grid, 256>>>(out, 10, 1.f, 2.f);
    hipEventRecord(e0);
    k<NACC><<<grid, 256>>>(out, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)grid * 4 * iters * 16 * 4096.0;
    printf("nacc=%d waves/SIMD=%d  %.3f ms  %.1f TFLOP/s\n", NACC, blocks_per_cu, ms, flop / ms / 1e9);
    hipFree(out);
}
int main() {
    for (int bpc = 1; bpc <= 3; ++bpc) { run<1>(bpc); run<2>(bpc); run<4>(bpc); run<8>(bpc); }
    return 0;
}
