// In-kernel clock stamps of the box-form input gradient of an upsampled convolution (64 -> 64, output
// 80x96x80, batch 8): where the time of a step goes -- the waits at its four barriers, the box-sum phase, the
// issue of the next halo loads, the multiply loop.
// Build:  hipcc --offload-arch=gfx950 -O3 -DSIV_UDSTAMP -I include -I soft-intro-vae-for-3d-mri_amd/csrc tools/ubench/up_dgrad_stamps.hip -o tools/ubench/up_dgrad_stamps
#include "conv_up.hip"
#include <vector>
#include <algorithm>
namespace siv {
void set_error(const char*, ...) {}
int conv_path() { return 0; }
void* pack_slot(const void*, int, int, int, int, int, size_t, void* ws, bool* need) { *need = true; return ws; }
}
__global__ void fill(float* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (float)(h & 0xffff) / 65536.f - 0.5f;
    }
}
int main() {
    const int N = 8, D = 80, H = 96, W = 80, C = 64;
    const size_t n = (size_t)N * D * H * W * C;
    float *dy, *dx, *w, *ws;
    hipMalloc(&dy, n * 4); hipMalloc(&dx, n / 8 * 4); hipMalloc(&w, 64 * 27 * 64 * 4); hipMalloc(&ws, 64 << 20);
    fill<<<4096, 256>>>(dy, n, 1); fill<<<64, 256>>>(w, 64 * 27 * 64, 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) siv::k2_up_dgrad_launch(dy, w, dx, N, D, H, W, C, C, ws, 64 << 20, 0, false, false);
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) siv::k2_up_dgrad_launch(dy, w, dx, N, D, H, W, C, C, ws, 64 << 20, 0, false, false);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    printf("%.3f ms per launch (with stamps)\n", ms);
    std::vector<unsigned long long> st(4096 * 4 * 32);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(siv::ud_stamps), st.size() * 8);
    const char* names[7] = {"wait B0", "store raw + wait B1", "box reads + sums", "wait B2", "store box + wait B3", "issue halo loads", "multiply loop"};
    double seg[7] = {0}, stepc = 0, cnt = 0;
    for (int b = 512; b < 2048; ++b)
        for (int wv = 0; wv < 4; ++wv)
            for (int s = 0; s < 3; ++s) {
                const unsigned long long* p = &st[((size_t)b * 4 + wv) * 32 + s * 8];
                if (!p[0] || !p[8]) continue;
                for (int k = 0; k < 7; ++k) seg[k] += (double)(p[k + 1] - p[k]);
                stepc += (double)(p[8] - p[0]);
                cnt += 1;
            }
    printf("cycles per step %.0f (floor: 144 x 64 = 9216 of matrix instructions per wave, 2 waves per SIMD)\n", stepc / cnt);
    for (int k = 0; k < 7; ++k) printf("  %-22s %7.0f\n", names[k], seg[k] / cnt);
    return 0;
}
