// Checks (1) the operand layout of v_mfma_f32_32x32x16_bf16 assumed by the split-product kernels and
// (2) the accuracy of an fp32 product formed from three bf16 pieces per operand (six piece products,
// two per instruction: lane half 0 and lane half 1 carry different piece pairs).
// Build: hipcc --offload-arch=gfx950 -O3 mfma_bf16_split.hip -o mfma_bf16_split
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    l = (__bf16)r2;
}

// D[32][32] = A[32][8] * B[8][32] in fp32-equivalent arithmetic, one wave
__global__ void k(const float* A, const float* B, float* D) {
    const int lane = threadIdx.x, i = lane & 31, lh = lane >> 5;
    bf16x8 ah, am, al, bh, bm, bl;
    for (int j = 0; j < 8; ++j) {
        __bf16 h, m, l;
        split3(A[i * 8 + j], h, m, l);
        ah[j] = h; am[j] = m; al[j] = l;
        split3(B[j * 32 + i], h, m, l);
        bh[j] = h; bm[j] = m; bl[j] = l;
    }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // lane half 0 | lane half 1
    const bf16x8 a1 = ah, a2 = lh ? ah : am, a3 = lh ? am : al;
    const bf16x8 bx = lh ? bm : bh, by = lh ? bl : bh;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, bx, acc, 0, 0, 0);   // l*h | m*m
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, by, acc, 0, 0, 0);   // m*h | h*l
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bx, acc, 0, 0, 0);   // h*h | h*m
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + i] = acc[r];
}

int main() {
    float hA[256], hB[256], hD[1024];
    double worst = 0, worst32 = 0;
    float *A, *B, *D;
    hipMalloc(&A, 1024); hipMalloc(&B, 1024); hipMalloc(&D, 4096);
    srand(1);
    for (int trial = 0; trial < 20; ++trial) {
        for (int i = 0; i < 256; ++i) {
            hA[i] = (float)((rand() / (double)RAND_MAX - 0.5) * exp2((rand() % 16) - 8));
            hB[i] = (float)((rand() / (double)RAND_MAX - 0.5) * exp2((rand() % 16) - 8));
        }
        hipMemcpy(A, hA, 1024, hipMemcpyHostToDevice);
        hipMemcpy(B, hB, 1024, hipMemcpyHostToDevice);
        k<<<1, 64>>>(A, B, D);
        hipMemcpy(hD, D, 4096, hipMemcpyDeviceToHost);
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double ref = 0, mag = 0;
                float f32 = 0.f;
                for (int kk = 0; kk < 8; ++kk) {
                    ref += (double)hA[i * 8 + kk] * hB[kk * 32 + j];
                    mag += fabs((double)hA[i * 8 + kk] * hB[kk * 32 + j]);
                    f32 = fmaf(hA[i * 8 + kk], hB[kk * 32 + j], f32);
                }
                worst = fmax(worst, fabs(hD[i * 32 + j] - ref) / mag);
                worst32 = fmax(worst32, fabs(f32 - ref) / mag);
            }
    }
    printf("max |split - exact| / sum|terms| = %.3e   (fp32 fmaf chain: %.3e)\n", worst, worst32);
    return worst < 1e-6 ? 0 : 1;
}
