// In-kernel clock stamps of k3_fwd_wino2d (64 -> 64 @ 80x96x80, batch 8, random data): where a block's
// time goes -- prologue, each chunk (multiply + staging), the wait at each barrier, epilogue.
// Build:  hipcc --offload-arch=gfx950 -O3 -DSIV_W2STAMP -I include -I soft-intro-vae-for-3d-mri_amd/csrc tools/ubench/wino2d_stamps.hip -o tools/ubench/wino2d_stamps
#include "conv_wino2d.hip"
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
    float *x, *y, *w, *ws;
    hipMalloc(&x, n * 4); hipMalloc(&y, n * 4); hipMalloc(&w, 64 * 27 * 64 * 4); hipMalloc(&ws, 64 << 20);
    fill<<<4096, 256>>>(x, n, 1); fill<<<64, 256>>>(w, 64 * 27 * 64, 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) siv::k3_fwd_wino2d_launch(x, w, nullptr, y, N, D, H, W, C, C, 0, ws, 64 << 20, 0, nullptr, nullptr, 0);
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) siv::k3_fwd_wino2d_launch(x, w, nullptr, y, N, D, H, W, C, C, 0, ws, 64 << 20, 0, nullptr, nullptr, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    printf("%.3f ms per launch (with stamps)\n", ms);
    std::vector<unsigned long long> st(2048 * 8 * 32);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(siv::w2_stamps), st.size() * 8);
    // per block: the stamps of the LAST brick of the last launch (a block runs its bricks as one chunk stream;
    // stamp 2 = start of that brick's chunk loop, stamps 0/1 = the block's prologue)
    auto at = [&](int b, int wv, int i) { return (double)st[((size_t)b * 8 + wv) * 32 + i]; };
    double pro = 0, epi = 0, tot = 0, chunk[8] = {0}, wait[9] = {0}, waitw[8] = {0}, cnt = 0;
    for (int b = 0; b < 256; ++b) {
        double t0 = 1e30, tend = 0;
        for (int wv = 0; wv < 8; ++wv) { t0 = std::min(t0, at(b, wv, 0)); tend = std::max(tend, at(b, wv, 30)); }
        tot += tend - t0;
        double after_pro = 0; for (int wv = 0; wv < 8; ++wv) after_pro = std::max(after_pro, at(b, wv, 2));
        pro += after_pro - t0;
        double prev = after_pro;
        for (int c = 0; c < 8; ++c) {
            double aft = 0; for (int wv = 0; wv < 8; ++wv) aft = std::max(aft, at(b, wv, 4 + 2 * c));
            chunk[c] += aft - prev; prev = aft;
            for (int wv = 0; wv < 8; ++wv) { wait[c + 1] += (at(b, wv, 4 + 2 * c) - at(b, wv, 3 + 2 * c)) / 8; waitw[wv] += (at(b, wv, 4 + 2 * c) - at(b, wv, 3 + 2 * c)) / 8; }
        }
        for (int wv = 0; wv < 8; ++wv) wait[0] += (at(b, wv, 2) - at(b, wv, 1)) / 8;
        epi += tend - prev;
        cnt += 1;
    }
    printf("cycles per block %.0f: prologue %.0f, chunks", tot / cnt, pro / cnt);
    for (int c = 0; c < 8; ++c) printf(" %.0f", chunk[c] / cnt);
    printf(", epilogue %.0f\n", epi / cnt);
    printf("average wait at the barrier (per wave): prologue %.0f, chunks", wait[0] / cnt);
    for (int c = 1; c < 9; ++c) printf(" %.0f", wait[c] / cnt);
    printf("\nper-wave barrier wait per chunk:");
    for (int wv = 0; wv < 8; ++wv) printf(" w%d %.0f", wv, waitw[wv] / cnt);
    printf("\n(MFMA floor per chunk: 2 waves x 96 x 64 = 12288 cycles per SIMD)\n");
    return 0;
}
