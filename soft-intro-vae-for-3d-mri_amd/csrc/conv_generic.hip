// Any-shape direct convolution kernels (VALU).  They serve channel counts the
// specialised kernels do not cover (narrow test/plumbing models, 1x1x1 heads) and are
// the on-device cross-check for the specialised paths.  NDHWC fp32 throughout.
#include "siv_common.h"

namespace siv {

// ---------------------------------------------------------------------------------
// 3x3x3 forward: one thread per output element (co fastest -> coalesced stores).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k3_fwd_generic(const float* __restrict__ x,
                                                      const float* __restrict__ w,
                                                      const float* __restrict__ bias,
                                                      float* __restrict__ y, int N, int D, int H,
                                                      int W, int Cin, int Cout, int ups) {
    const long total = (long)N * D * H * W * Cout;
    const int Di = D / ups, Hi = H / ups, Wi = W / ups;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const int co = (int)(i % Cout);
        long v = i / Cout;
        const int ww = (int)(v % W); v /= W;
        const int hh = (int)(v % H); v /= H;
        const int dd = (int)(v % D);
        const int n = (int)(v / D);
        float acc = bias ? bias[co] : 0.f;
        for (int kd = 0; kd < 3; ++kd) {
            const int zd = dd + kd - 1;
            if (zd < 0 || zd >= D) continue;
            for (int kh = 0; kh < 3; ++kh) {
                const int zh = hh + kh - 1;
                if (zh < 0 || zh >= H) continue;
                for (int kw = 0; kw < 3; ++kw) {
                    const int zw = ww + kw - 1;
                    if (zw < 0 || zw >= W) continue;
                    const float* xp =
                        x + ((((long)n * Di + zd / ups) * Hi + zh / ups) * Wi + zw / ups) * Cin;
                    const float* wp = w + ((long)co * 27 + (kd * 9 + kh * 3 + kw)) * Cin;
                    for (int ci = 0; ci < Cin; ++ci) acc = fmaf(xp[ci], wp[ci], acc);
                }
            }
        }
        y[i] = acc;
    }
}

// 3x3x3 dgrad: dx[v][ci] = sum_{tap,co} dy[v - (tap-1)][co] * w[co][tap][ci]
__global__ __launch_bounds__(256) void k3_dgrad_generic(const float* __restrict__ dy,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ bias,
                                                        float* __restrict__ dx, int N, int D,
                                                        int H, int W, int Cin, int Cout) {
    const long total = (long)N * D * H * W * Cin;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Cin);
        long v = i / Cin;
        const int ww = (int)(v % W); v /= W;
        const int hh = (int)(v % H); v /= H;
        const int dd = (int)(v % D);
        const int n = (int)(v / D);
        float acc = bias ? bias[ci] : 0.f;      // bias: the transposed convolution's (per dx channel)
        for (int kd = 0; kd < 3; ++kd) {
            const int zd = dd - kd + 1;
            if (zd < 0 || zd >= D) continue;
            for (int kh = 0; kh < 3; ++kh) {
                const int zh = hh - kh + 1;
                if (zh < 0 || zh >= H) continue;
                for (int kw = 0; kw < 3; ++kw) {
                    const int zw = ww - kw + 1;
                    if (zw < 0 || zw >= W) continue;
                    const float* gp = dy + ((((long)n * D + zd) * H + zh) * W + zw) * Cout;
                    const float* wp = w + (long)(kd * 9 + kh * 3 + kw) * Cin + ci;
                    for (int co = 0; co < Cout; ++co)
                        acc = fmaf(gp[co], wp[(long)co * 27 * Cin], acc);
                }
            }
        }
        dx[i] = acc;
    }
}

// 3x3x3 wgrad: one block per dw element, fixed-order tree reduction over all voxels.
__global__ __launch_bounds__(256) void k3_wgrad_generic(const float* __restrict__ x,
                                                        const float* __restrict__ dy,
                                                        float* __restrict__ dw, int N, int D,
                                                        int H, int W, int Cin, int Cout, int ups,
                                                        int accumulate) {
    __shared__ float sh[4];
    const long e = blockIdx.x;  // (co, tap, ci)
    const int ci = (int)(e % Cin);
    const int tap = (int)((e / Cin) % 27);
    const int co = (int)(e / ((long)Cin * 27));
    const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
    const int Di = D / ups, Hi = H / ups, Wi = W / ups;
    const long V = (long)N * D * H * W;
    float acc = 0.f;
    for (long v = threadIdx.x; v < V; v += blockDim.x) {
        long t = v;
        const int ww = (int)(t % W); t /= W;
        const int hh = (int)(t % H); t /= H;
        const int dd = (int)(t % D);
        const int n = (int)(t / D);
        const int zd = dd + kd - 1, zh = hh + kh - 1, zw = ww + kw - 1;
        if (zd < 0 || zd >= D || zh < 0 || zh >= H || zw < 0 || zw >= W) continue;
        const float xv =
            x[((((long)n * Di + zd / ups) * Hi + zh / ups) * Wi + zw / ups) * Cin + ci];
        acc = fmaf(xv, dy[v * Cout + co], acc);
    }
    acc = block_sum_256(acc, sh);
    if (threadIdx.x == 0) dw[e] = accumulate ? dw[e] + acc : acc;
}

// Column sums of a [M][C] matrix (bias gradients), two deterministic stages: per-block partial
// rows (threads = (row lane, channel), channel fastest so that a wave reads contiguous memory),
// then one wave per channel in fp64.
__global__ __launch_bounds__(256) void colsum_partial(const float* __restrict__ a, long M, int C,
                                                      float* __restrict__ part) {
    __shared__ float sh[256];
    const int lanes = C <= 256 ? 256 / C : 1;
    const long rows_per_block = (M + gridDim.x - 1) / gridDim.x;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = (r0 + rows_per_block < M) ? r0 + rows_per_block : M;
    for (int cb = 0; cb < C; cb += 256) {
        const int c = cb + (C <= 256 ? (int)(threadIdx.x % C) : (int)threadIdx.x);
        const int lane = C <= 256 ? (int)(threadIdx.x / C) : 0;
        float acc = 0.f;
        if (lane < lanes && c < C)
            for (long r = r0 + lane; r < r1; r += lanes) acc += a[r * C + c];
        __syncthreads();
        sh[threadIdx.x] = acc;
        __syncthreads();
        if (lane == 0 && c < C) {
            float t = 0.f;
            for (int l = 0; l < lanes; ++l) t += sh[C <= 256 ? l * C + (int)(threadIdx.x % C) : (int)threadIdx.x];
            part[(long)blockIdx.x * C + c] = t;
        }
        if (C <= 256) break;
    }
}
__global__ __launch_bounds__(256) void colsum_final(const float* __restrict__ part, int nblk, int C,
                                                    float* __restrict__ out, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    double s = 0.0;
    for (int b = lane; b < nblk; b += 64) s += (double)part[(long)b * C + c];
    s = wave_sum_d(s);
    if (lane == 0) out[c] = accumulate ? out[c] + (float)s : (float)s;
}

__global__ __launch_bounds__(256) void upsample2_bwd_k(const float* __restrict__ dy,
                                                       float* __restrict__ dx, int N, int D, int H,
                                                       int W, int C) {
    const int Do = D / 2, Ho = H / 2, Wo = W / 2;
    const long total = (long)N * Do * Ho * Wo * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long v = i / C;
        const int ww = (int)(v % Wo); v /= Wo;
        const int hh = (int)(v % Ho); v /= Ho;
        const int dd = (int)(v % Do);
        const int n = (int)(v / Do);
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 2; ++e)
                    acc += dy[((((long)n * D + 2 * dd + a) * H + 2 * hh + b) * W + 2 * ww + e) * C + c];
        dx[i] = acc;
    }
}

// ---------------------------------------------------------------------------------
// 1x1x1
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void c1_fwd_generic(const float* __restrict__ x,
                                                      const float* __restrict__ w,
                                                      const float* __restrict__ bias,
                                                      float* __restrict__ y, long M, int Cin,
                                                      int Cout) {
    const long total = M * Cout;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const int co = (int)(i % Cout);
        const long r = i / Cout;
        float acc = bias ? bias[co] : 0.f;
        const float* xp = x + r * Cin;
        const float* wp = w + (long)co * Cin;
        for (int ci = 0; ci < Cin; ++ci) acc = fmaf(xp[ci], wp[ci], acc);
        y[i] = acc;
    }
}

__global__ __launch_bounds__(256) void c1_dgrad_generic(const float* __restrict__ dy,
                                                        const float* __restrict__ w,
                                                        float* __restrict__ dx, long M, int Cin,
                                                        int Cout) {
    const long total = M * Cin;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Cin);
        const long r = i / Cin;
        float acc = 0.f;
        const float* gp = dy + r * Cout;
        for (int co = 0; co < Cout; ++co) acc = fmaf(gp[co], w[(long)co * Cin + ci], acc);
        dx[i] = acc;
    }
}

__global__ __launch_bounds__(256) void c1_wgrad_generic(const float* __restrict__ x,
                                                        const float* __restrict__ dy,
                                                        float* __restrict__ dw, long M, int Cin,
                                                        int Cout, int accumulate) {
    __shared__ float sh[4];
    const int ci = blockIdx.x % Cin;
    const int co = blockIdx.x / Cin;
    float acc = 0.f;
    for (long r = threadIdx.x; r < M; r += blockDim.x)
        acc = fmaf(x[r * Cin + ci], dy[r * Cout + co], acc);
    acc = block_sum_256(acc, sh);
    if (threadIdx.x == 0) dw[blockIdx.x] = accumulate ? dw[blockIdx.x] + acc : acc;
}

static inline int grid_for(long total) {
    long g = cdiv(total, 256);
    if (g > 256L * 32) g = 256L * 32;
    if (g < 1) g = 1;
    return (int)g;
}

int k3_fwd_generic_launch(const float* x, const float* w, const float* bias, float* y, int N, int D,
                          int H, int W, int Cin, int Cout, int ups, hipStream_t s) {
    k3_fwd_generic<<<grid_for((long)N * D * H * W * Cout), 256, 0, s>>>(x, w, bias, y, N, D, H, W,
                                                                       Cin, Cout, ups);
    SIV_CHECK_LAUNCH("conv3d_k3_fwd(generic)");
    return SIV_OK;
}
int k3_dgrad_generic_launch(const float* dy, const float* w, const float* bias, float* dx, int N, int D,
                            int H, int W, int Cin, int Cout, hipStream_t s) {
    k3_dgrad_generic<<<grid_for((long)N * D * H * W * Cin), 256, 0, s>>>(dy, w, bias, dx, N, D, H, W, Cin,
                                                                        Cout);
    SIV_CHECK_LAUNCH("conv3d_k3_dgrad(generic)");
    return SIV_OK;
}
int colsum_launch(const float* a, float* out, long M, int C, int accumulate, void* ws, size_t ws_bytes,
                  hipStream_t s) {
    long nblk = cdiv(M, 64);
    if (nblk > 1024) nblk = 1024;
    SIV_CHECK_WS((size_t)nblk * C * sizeof(float), ws_bytes, "bias gradient");
    colsum_partial<<<(unsigned)nblk, 256, 0, s>>>(a, M, C, (float*)ws);
    SIV_CHECK_LAUNCH("bias gradient(partial)");
    colsum_final<<<(C + 3) / 4, 256, 0, s>>>((const float*)ws, (int)nblk, C, out, accumulate);
    SIV_CHECK_LAUNCH("bias gradient(final)");
    return SIV_OK;
}

int k3_wgrad_generic_launch(const float* x, const float* dy, float* dw, float* dbias, int N, int D,
                            int H, int W, int Cin, int Cout, int ups, int accumulate, void* ws,
                            size_t ws_bytes, hipStream_t s) {
    k3_wgrad_generic<<<(unsigned)((long)Cout * 27 * Cin), 256, 0, s>>>(x, dy, dw, N, D, H, W, Cin,
                                                                      Cout, ups, accumulate);
    SIV_CHECK_LAUNCH("conv3d_k3_wgrad(generic)");
    if (dbias) return colsum_launch(dy, dbias, (long)N * D * H * W, Cout, accumulate, ws, ws_bytes, s);
    return SIV_OK;
}

}  // namespace siv

using namespace siv;

extern "C" int siv_upsample2_bwd(const float* dy, float* dx, int N, int D, int H, int W, int C,
                                 siv_stream_t stream) {
    SIV_CHECK_ARG(dy && dx, "upsample2_bwd: null pointer");
    SIV_CHECK_ARG(N > 0 && C > 0 && D > 0 && H > 0 && W > 0 && D % 2 == 0 && H % 2 == 0 && W % 2 == 0,
                  "upsample2_bwd: extents must be positive and even (got %d,%d,%d)", D, H, W);
    upsample2_bwd_k<<<grid_for((long)N * (D / 2) * (H / 2) * (W / 2) * C), 256, 0,
                      (hipStream_t)stream>>>(dy, dx, N, D, H, W, C);
    SIV_CHECK_LAUNCH("upsample2_bwd");
    return SIV_OK;
}

extern "C" int siv_conv1x1_fwd(const float* x, const float* w, const float* bias, float* y, long M,
                               int Cin, int Cout, siv_stream_t stream) {
    SIV_CHECK_ARG(x && w && y, "conv1x1_fwd: null pointer");
    SIV_CHECK_ARG(M > 0 && Cin > 0 && Cout > 0, "conv1x1_fwd: bad shape");
    c1_fwd_generic<<<grid_for(M * Cout), 256, 0, (hipStream_t)stream>>>(x, w, bias, y, M, Cin, Cout);
    SIV_CHECK_LAUNCH("conv1x1_fwd");
    return SIV_OK;
}

extern "C" int siv_conv1x1_dgrad(const float* dy, const float* w, float* dx, long M, int Cin,
                                 int Cout, siv_stream_t stream) {
    SIV_CHECK_ARG(dy && w && dx, "conv1x1_dgrad: null pointer");
    SIV_CHECK_ARG(M > 0 && Cin > 0 && Cout > 0, "conv1x1_dgrad: bad shape");
    c1_dgrad_generic<<<grid_for(M * Cin), 256, 0, (hipStream_t)stream>>>(dy, w, dx, M, Cin, Cout);
    SIV_CHECK_LAUNCH("conv1x1_dgrad");
    return SIV_OK;
}

extern "C" int siv_conv1x1_wgrad(const float* x, const float* dy, float* dw, float* dbias, long M,
                                 int Cin, int Cout, int accumulate, void* ws, size_t ws_bytes,
                                 siv_stream_t stream) {
    SIV_CHECK_ARG(x && dy && dw, "conv1x1_wgrad: null pointer");
    SIV_CHECK_ARG(M > 0 && Cin > 0 && Cout > 0, "conv1x1_wgrad: bad shape");
    c1_wgrad_generic<<<Cin * Cout, 256, 0, (hipStream_t)stream>>>(x, dy, dw, M, Cin, Cout, accumulate);
    SIV_CHECK_LAUNCH("conv1x1_wgrad");
    if (dbias) return colsum_launch(dy, dbias, M, Cout, accumulate, ws, ws_bytes, (hipStream_t)stream);
    return SIV_OK;
}
