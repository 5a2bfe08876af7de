// Stand-alone timing of the weight-gradient kernel (no torch).
#include "conv_mfma.hip"
namespace siv {
void set_error(const char*, ...) {}
int conv_path() { return 0; }
void* pack_slot(const void*, int, int, int, int, int, size_t, void* ws, bool* need) { *need = true; return ws; }
}
int main() {
    const int N = 8, D = 80, H = 96, W = 80, C = 64;
    const size_t n = (size_t)N * D * H * W * C;
    float *x, *dy, *dw, *ws;
    hipMalloc(&x, n * 4); hipMalloc(&dy, n * 4); hipMalloc(&dw, 64 * 27 * 64 * 4); hipMalloc(&ws, 512 << 20);
    hipMemset(x, 0, n * 4); hipMemset(dy, 0, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    siv::k3_wgrad_mfma_launch(x, dy, dw, N, D, H, W, C, C, 1, 0, ws, (size_t)512 << 20, 0);
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) siv::k3_wgrad_mfma_launch(x, dy, dw, N, D, H, W, C, C, 1, 0, ws, (size_t)512 << 20, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    printf("wgrad 64->64@80x96x80 B=8  %.3f ms  %.1f TFLOP/s\n", ms, 2.0 * 27 * 64 * 64 * N * D * H * W / ms / 1e9);
    return 0;
}
