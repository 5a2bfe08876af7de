// Fully connected layers of the FC-latent model (batch of a few samples against a 19200-wide
// feature vector): weight-bandwidth bound, so each kernel is one coalesced pass over the weight
// matrix.  Also the elementwise add + activation of that model's skip connections.
#include "siv_common.h"

namespace siv {

constexpr int LB = 8;  // batch rows processed together

// y[b][o] = bias[o] + sum_k x[b][k] * w[o][k]; one wave per output row o.
__global__ __launch_bounds__(256) void linear_fwd_k(const float* __restrict__ x,
                                                    const float* __restrict__ w,
                                                    const float* __restrict__ bias,
                                                    float* __restrict__ y, int B, int K, int O) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= O) return;
    const float* wr = w + (long)o * K;
    for (int b0 = 0; b0 < B; b0 += LB) {
        float acc[LB];
#pragma unroll
        for (int j = 0; j < LB; ++j) acc[j] = 0.f;
        for (int k = lane; k < K; k += 64) {
            const float wv = wr[k];
#pragma unroll
            for (int j = 0; j < LB; ++j)
                if (b0 + j < B) acc[j] = fmaf(wv, x[(long)(b0 + j) * K + k], acc[j]);
        }
#pragma unroll
        for (int j = 0; j < LB; ++j) {
            const float s = wave_sum(acc[j]);
            if (lane == 0 && b0 + j < B) y[(long)(b0 + j) * O + o] = s + (bias ? bias[o] : 0.f);
        }
    }
}

// partial[s][b][k] = sum over the s-th slice of output rows o of dy[b][o] * w[o][k]
__global__ __launch_bounds__(256) void linear_dgrad_partial(const float* __restrict__ dy,
                                                            const float* __restrict__ w,
                                                            float* __restrict__ part, int B, int K,
                                                            int O, int rows_per_slice) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const int o0 = blockIdx.y * rows_per_slice;
    const int o1 = (o0 + rows_per_slice < O) ? o0 + rows_per_slice : O;
    for (int b0 = 0; b0 < B; b0 += LB) {
        float acc[LB];
#pragma unroll
        for (int j = 0; j < LB; ++j) acc[j] = 0.f;
        for (int o = o0; o < o1; ++o) {
            const float wv = w[(long)o * K + k];
#pragma unroll
            for (int j = 0; j < LB; ++j)
                if (b0 + j < B) acc[j] = fmaf(wv, dy[(long)(b0 + j) * O + o], acc[j]);
        }
#pragma unroll
        for (int j = 0; j < LB; ++j)
            if (b0 + j < B) part[((long)blockIdx.y * B + b0 + j) * K + k] = acc[j];
    }
}
__global__ __launch_bounds__(256) void linear_dgrad_reduce(const float* __restrict__ part, int nslice,
                                                           long n, float* __restrict__ dx) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int t = 0; t < nslice; ++t) s += part[(long)t * n + i];
        dx[i] = s;
    }
}

// dw[o][k] = sum_b dy[b][o] * x[b][k]
__global__ __launch_bounds__(256) void linear_wgrad_k(const float* __restrict__ x,
                                                      const float* __restrict__ dy,
                                                      float* __restrict__ dw, int B, int K, int O,
                                                      int accumulate) {
    const long total = (long)O * K;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % K);
        const int o = (int)(i / K);
        float s = 0.f;
        for (int b = 0; b < B; ++b) s = fmaf(dy[(long)b * O + o], x[(long)b * K + k], s);
        dw[i] = accumulate ? dw[i] + s : s;
    }
}
__global__ __launch_bounds__(256) void linear_bgrad_k(const float* __restrict__ dy,
                                                      float* __restrict__ db, int B, int O,
                                                      int accumulate) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= O) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dy[(long)b * O + o];
    db[o] = accumulate ? db[o] + s : s;
}

__global__ __launch_bounds__(256) void add_act_fwd_k(const float* __restrict__ a,
                                                     const float* __restrict__ b, float slope,
                                                     float* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long)gridDim.x * blockDim.x)
        out[i] = leaky(a[i] + b[i], slope);
}
__global__ __launch_bounds__(256) void add_act_bwd_k(const float* __restrict__ dout,
                                                     const float* __restrict__ a,
                                                     const float* __restrict__ b, float slope,
                                                     float* __restrict__ d, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long)gridDim.x * blockDim.x)
        d[i] = dout[i] * leaky_grad(a[i] + b[i], slope);
}

static inline int lin_grid(long n) {
    long g = cdiv(n, 256);
    if (g > 256L * 16) g = 256L * 16;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace siv

using namespace siv;

extern "C" int siv_linear_fwd(const float* x, const float* w, const float* bias, float* y, int B,
                              int K, int O, siv_stream_t stream) {
    SIV_CHECK_ARG(x && w && y, "linear_fwd: null pointer");
    SIV_CHECK_ARG(B > 0 && K > 0 && O > 0, "linear_fwd: bad shape B=%d K=%d O=%d", B, K, O);
    linear_fwd_k<<<(O + 3) / 4, 256, 0, (hipStream_t)stream>>>(x, w, bias, y, B, K, O);
    SIV_CHECK_LAUNCH("linear_fwd");
    return SIV_OK;
}

extern "C" int siv_linear_dgrad(const float* dy, const float* w, float* dx, int B, int K, int O,
                                void* ws, size_t ws_bytes, siv_stream_t stream) {
    SIV_CHECK_ARG(dy && w && dx, "linear_dgrad: null pointer");
    SIV_CHECK_ARG(B > 0 && K > 0 && O > 0, "linear_dgrad: bad shape B=%d K=%d O=%d", B, K, O);
    const int kblocks = (K + 255) / 256;
    int nslice = 1024 / kblocks;
    if (nslice < 1) nslice = 1;
    if (nslice > O) nslice = O;
    const int rps = (O + nslice - 1) / nslice;
    nslice = (O + rps - 1) / rps;
    SIV_CHECK_WS((size_t)nslice * B * K * sizeof(float), ws_bytes, "linear_dgrad");
    hipStream_t s = (hipStream_t)stream;
    linear_dgrad_partial<<<dim3(kblocks, nslice), 256, 0, s>>>(dy, w, (float*)ws, B, K, O, rps);
    SIV_CHECK_LAUNCH("linear_dgrad(partial)");
    linear_dgrad_reduce<<<lin_grid((long)B * K), 256, 0, s>>>((const float*)ws, nslice, (long)B * K, dx);
    SIV_CHECK_LAUNCH("linear_dgrad(reduce)");
    return SIV_OK;
}

extern "C" int siv_linear_wgrad(const float* x, const float* dy, float* dw, float* dbias, int B,
                                int K, int O, int accumulate, siv_stream_t stream) {
    SIV_CHECK_ARG(x && dy && dw, "linear_wgrad: null pointer");
    SIV_CHECK_ARG(B > 0 && K > 0 && O > 0, "linear_wgrad: bad shape B=%d K=%d O=%d", B, K, O);
    hipStream_t s = (hipStream_t)stream;
    linear_wgrad_k<<<lin_grid((long)O * K), 256, 0, s>>>(x, dy, dw, B, K, O, accumulate);
    SIV_CHECK_LAUNCH("linear_wgrad");
    if (dbias) {
        linear_bgrad_k<<<(O + 255) / 256, 256, 0, s>>>(dy, dbias, B, O, accumulate);
        SIV_CHECK_LAUNCH("linear_wgrad(bias)");
    }
    return SIV_OK;
}

extern "C" int siv_add_act_fwd(const float* a, const float* b, float slope, float* out, long n,
                               siv_stream_t stream) {
    SIV_CHECK_ARG(a && b && out && n > 0, "add_act_fwd: bad argument");
    add_act_fwd_k<<<lin_grid(n), 256, 0, (hipStream_t)stream>>>(a, b, slope, out, n);
    SIV_CHECK_LAUNCH("add_act_fwd");
    return SIV_OK;
}

extern "C" int siv_add_act_bwd(const float* dout, const float* a, const float* b, float slope,
                               float* d, long n, siv_stream_t stream) {
    SIV_CHECK_ARG(dout && a && b && d && n > 0, "add_act_bwd: bad argument");
    add_act_bwd_k<<<lin_grid(n), 256, 0, (hipStream_t)stream>>>(dout, a, b, slope, d, n);
    SIV_CHECK_LAUNCH("add_act_bwd");
    return SIV_OK;
}
