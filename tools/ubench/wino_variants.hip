// Timing-only ablations of the Winograd forward conv kernel (see SIV_WEXP in conv_wino.hip).
// Build one binary per mask:  hipcc --offload-arch=gfx950 -O3 -DSIV_WEXP=<mask> -I include -I csrc wino_variants.hip
#include "conv_wino.hip"
namespace siv {
void set_error(const char*, ...) {}
int conv_path() { return 0; }
void* pack_slot(const void*, int, int, int, int, int, size_t, void* ws, bool* need) { *need = true; return ws; }
}
int main() {
    const int N = 8, D = 80, H = 96, W = 80, C = 64;
    const size_t n = (size_t)N * D * H * W * C;
    float *x, *y, *w, *ws;
    hipMalloc(&x, n * 4); hipMalloc(&y, n * 4); hipMalloc(&w, 64 * 27 * 64 * 4); hipMalloc(&ws, 64 << 20);
    hipMemset(x, 0, n * 4); hipMemset(w, 0, 64 * 27 * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    siv::k3_fwd_wino_launch(x, w, nullptr, y, N, D, H, W, C, C, 1, 0, ws, 64 << 20, 0, nullptr);
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) siv::k3_fwd_wino_launch(x, w, nullptr, y, N, D, H, W, C, C, 1, 0, ws, 64 << 20, 0, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    printf("SIV_WEXP=%d  %.3f ms  %.1f TFLOP/s algorithmic, %.1f executed\n", SIV_WEXP, ms,
           2.0 * 27 * 64 * 64 * N * D * H * W / ms / 1e9, 2.0 * 18 * 64 * 64 * N * D * H * W / ms / 1e9);
    return 0;
}
