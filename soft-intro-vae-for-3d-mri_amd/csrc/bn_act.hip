// BatchNorm3d statistics and the fused normalise + residual + activation + pool +
// dropout tail (forward and backward).  All HBM-bound streaming kernels over NDHWC fp32.
#include "siv_common.h"

namespace siv {

constexpr int kMaxStatBlocks = 2048;

// Thread <-> (row lane, channel) mapping shared by the per-channel reductions:
// channel index fastest so that a wave reads contiguous memory.
struct ChanMap {
    int lanes;   // rows processed concurrently by one block
    int lane;    // this thread's row lane (or -1 when idle)
    int c0;      // first channel of this thread
    int cstep;   // channel stride (>= C means a single channel)
};
__device__ __forceinline__ ChanMap chan_map(int C) {
    ChanMap m;
    if (C <= 256) {
        m.lanes = 256 / C;
        m.lane = threadIdx.x / C;
        m.c0 = threadIdx.x % C;
        m.cstep = C;
        if (m.lane >= m.lanes) m.lane = -1;
    } else {
        m.lanes = 1;
        m.lane = 0;
        m.c0 = threadIdx.x;
        m.cstep = 256;
    }
    return m;
}

// Per-block partial sums of (y - shift) and (y - shift)^2, shift = y[0][c].
// part layout: [block][2][C]
__global__ __launch_bounds__(256) void bn_stats_partial(const float* __restrict__ y, long M, int C,
                                                        float* __restrict__ part) {
    extern __shared__ float sh[];  // [2][256]
    const ChanMap m = chan_map(C);
    const long rows_per_block = (M + gridDim.x - 1) / gridDim.x;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = (r0 + rows_per_block < M) ? r0 + rows_per_block : M;
    for (int cb = 0; cb < C; cb += (C <= 256 ? C : 256)) {
        const int c = cb + m.c0;
        float s1 = 0.f, s2 = 0.f;
        if (m.lane >= 0 && c < C) {
            const float shift = y[c];
            for (long r = r0 + m.lane; r < r1; r += m.lanes) {
                const float d = y[r * C + c] - shift;
                s1 += d;
                s2 = fmaf(d, d, s2);
            }
        }
        __syncthreads();
        sh[threadIdx.x] = s1;
        sh[256 + threadIdx.x] = s2;
        __syncthreads();
        if (m.lane == 0 && c < C) {
            float a1 = 0.f, a2 = 0.f;
            for (int l = 0; l < m.lanes; ++l) {
                const int t = (C <= 256) ? l * C + m.c0 : threadIdx.x;
                a1 += sh[t];
                a2 += sh[256 + t];
            }
            part[((long)blockIdx.x * 2 + 0) * C + c] = a1;
            part[((long)blockIdx.x * 2 + 1) * C + c] = a2;
        }
    }
}

// one wave per channel: lanes stride over the per-block partials, fixed-order wave reduction
__global__ __launch_bounds__(256) void bn_stats_finalize(const float* __restrict__ y,
                                                         const float* __restrict__ part, int nblk,
                                                         long M, int C, float* __restrict__ mean,
                                                         float* __restrict__ var,
                                                         float* __restrict__ rmean,
                                                         float* __restrict__ rvar,
                                                         int64_t* __restrict__ nbt, float momentum) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int b = lane; b < nblk; b += 64) {
        s1 += (double)part[((long)b * 2 + 0) * C + c];
        s2 += (double)part[((long)b * 2 + 1) * C + c];
    }
    s1 = wave_sum_d(s1);
    s2 = wave_sum_d(s2);
    if (lane != 0) return;
    const double ms = s1 / (double)M;
    double v = s2 / (double)M - ms * ms;
    if (v < 0.0) v = 0.0;
    const double mu = (double)y[c] + ms;
    mean[c] = (float)mu;
    var[c] = (float)v;
    if (rmean) {
        const double unb = (M > 1) ? v * ((double)M / (double)(M - 1)) : v;
        rmean[c] = (float)((1.0 - (double)momentum) * (double)rmean[c] + (double)momentum * mu);
        rvar[c] = (float)((1.0 - (double)momentum) * (double)rvar[c] + (double)momentum * unb);
    }
}

// out = dropout(avgpool(act(gamma*xhat + beta [+ res])))
template <int POOL>
__global__ __launch_bounds__(256) void bn_act_fwd_k(const float* __restrict__ y,
                                                    const float* __restrict__ mean,
                                                    const float* __restrict__ var,
                                                    const float* __restrict__ gamma,
                                                    const float* __restrict__ beta,
                                                    const float* __restrict__ res,
                                                    const uint8_t* __restrict__ keep, float p,
                                                    float slope, float eps, float* __restrict__ out,
                                                    int N, int D, int H, int W, int C) {
    const int Do = D / POOL, Ho = H / POOL, Wo = W / POOL;
    const long total = (long)N * Do * Ho * Wo * C;
    const float dscale = 1.0f / (1.0f - p);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const float a = gamma[c] / sqrtf(var[c] + eps);
        const float b = beta[c] - mean[c] * a;
        float o;
        if (POOL == 1) {
            float z = fmaf(y[i], a, b);
            if (res) z += res[i];
            o = leaky(z, slope);
        } else {
            long v = i / C;
            const int ww = (int)(v % Wo); v /= Wo;
            const int hh = (int)(v % Ho); v /= Ho;
            const int dd = (int)(v % Do);
            const int n = (int)(v / Do);
            float acc = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const long j = ((((long)n * D + 2 * dd + (e >> 2)) * H + 2 * hh + ((e >> 1) & 1)) * W +
                                2 * ww + (e & 1)) * C + c;
                float z = fmaf(y[j], a, b);
                if (res) z += res[j];
                acc += leaky(z, slope);
            }
            o = acc * 0.125f;
        }
        if (keep) o = keep[i] ? o * dscale : 0.f;
        out[i] = o;
    }
}

// dz for input element j (full resolution), given the output-space gradient
template <int POOL>
__device__ __forceinline__ float bn_dz(long j, int c, const float* __restrict__ dout,
                                       const float* __restrict__ y, const float* __restrict__ res,
                                       const uint8_t* __restrict__ keep, float dscale, float slope,
                                       float a, float b, int D, int H, int W, int C, float& xhat_out,
                                       float mean_c, float invstd_c) {
    const float yv = y[j];
    float z = fmaf(yv, a, b);
    if (res) z += res[j];
    long o;
    float sc = 1.f;
    if (POOL == 1) {
        o = j;
    } else {
        long v = j / C;
        const int ww = (int)(v % W); v /= W;
        const int hh = (int)(v % H); v /= H;
        const int dd = (int)(v % D);
        const long n = v / D;
        o = (((n * (D / 2) + dd / 2) * (H / 2) + hh / 2) * (W / 2) + ww / 2) * C + c;
        sc = 0.125f;
    }
    float g = dout[o] * sc;
    if (keep) g = keep[o] ? g * dscale : 0.f;
    xhat_out = (yv - mean_c) * invstd_c;
    return g * leaky_grad(z, slope);
}

// partial sums of dz and dz*xhat; part layout [block][2][C]
template <int POOL>
__global__ __launch_bounds__(256) void bn_bwd_partial(const float* __restrict__ dout,
                                                      const float* __restrict__ y,
                                                      const float* __restrict__ mean,
                                                      const float* __restrict__ var,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ beta,
                                                      const float* __restrict__ res,
                                                      const uint8_t* __restrict__ keep, float p,
                                                      float slope, float eps, long M, int D, int H,
                                                      int W, int C, float* __restrict__ part) {
    extern __shared__ float sh[];
    const ChanMap m = chan_map(C);
    const long rows_per_block = (M + gridDim.x - 1) / gridDim.x;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = (r0 + rows_per_block < M) ? r0 + rows_per_block : M;
    const float dscale = 1.0f / (1.0f - p);
    for (int cb = 0; cb < C; cb += (C <= 256 ? C : 256)) {
        const int c = cb + m.c0;
        float s1 = 0.f, s2 = 0.f;
        if (m.lane >= 0 && c < C) {
            const float invstd = 1.0f / sqrtf(var[c] + eps);
            const float a = gamma[c] * invstd;
            const float b = beta[c] - mean[c] * a;
            const float mc = mean[c];
            for (long r = r0 + m.lane; r < r1; r += m.lanes) {
                float xh;
                const float dz = bn_dz<POOL>(r * C + c, c, dout, y, res, keep, dscale, slope, a, b, D,
                                             H, W, C, xh, mc, invstd);
                s1 += dz;
                s2 = fmaf(dz, xh, s2);
            }
        }
        __syncthreads();
        sh[threadIdx.x] = s1;
        sh[256 + threadIdx.x] = s2;
        __syncthreads();
        if (m.lane == 0 && c < C) {
            float a1 = 0.f, a2 = 0.f;
            for (int l = 0; l < m.lanes; ++l) {
                const int t = (C <= 256) ? l * C + m.c0 : threadIdx.x;
                a1 += sh[t];
                a2 += sh[256 + t];
            }
            part[((long)blockIdx.x * 2 + 0) * C + c] = a1;
            part[((long)blockIdx.x * 2 + 1) * C + c] = a2;
        }
    }
}

// sums[0][c] = sum dz, sums[1][c] = sum dz*xhat; also dgamma/dbeta.  One wave per channel.
__global__ __launch_bounds__(256) void bn_bwd_finalize(const float* __restrict__ part, int nblk, int C,
                                                       float* __restrict__ sums,
                                                       float* __restrict__ dgamma,
                                                       float* __restrict__ dbeta, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int b = lane; b < nblk; b += 64) {
        s1 += (double)part[((long)b * 2 + 0) * C + c];
        s2 += (double)part[((long)b * 2 + 1) * C + c];
    }
    s1 = wave_sum_d(s1);
    s2 = wave_sum_d(s2);
    if (lane != 0) return;
    sums[c] = (float)s1;
    sums[C + c] = (float)s2;
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)s1 : (float)s1;
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)s2 : (float)s2;
}

template <int POOL>
__global__ __launch_bounds__(256) void bn_bwd_apply(const float* __restrict__ dout,
                                                    const float* __restrict__ y,
                                                    const float* __restrict__ mean,
                                                    const float* __restrict__ var,
                                                    const float* __restrict__ gamma,
                                                    const float* __restrict__ beta,
                                                    const float* __restrict__ res,
                                                    const uint8_t* __restrict__ keep, float p,
                                                    float slope, float eps,
                                                    const float* __restrict__ sums, long M, int D,
                                                    int H, int W, int C, float* __restrict__ dy,
                                                    float* __restrict__ dres) {
    const long total = M * C;
    const float dscale = 1.0f / (1.0f - p);
    const float invM = 1.0f / (float)M;
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < total;
         j += (long)gridDim.x * blockDim.x) {
        const int c = (int)(j % C);
        const float invstd = 1.0f / sqrtf(var[c] + eps);
        const float a = gamma[c] * invstd;
        const float b = beta[c] - mean[c] * a;
        float xh;
        const float dz = bn_dz<POOL>(j, c, dout, y, res, keep, dscale, slope, a, b, D, H, W, C, xh,
                                     mean[c], invstd);
        dy[j] = a * (dz - sums[c] * invM - xh * (sums[C + c] * invM));
        if (dres) dres[j] = dz;
    }
}

__global__ __launch_bounds__(256) void act_dropout_fwd_k(const float* __restrict__ x,
                                                         const uint8_t* __restrict__ keep, float p,
                                                         float slope, float* __restrict__ out, long n) {
    const float dscale = 1.0f / (1.0f - p);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long)gridDim.x * blockDim.x) {
        float o = leaky(x[i], slope);
        if (keep) o = keep[i] ? o * dscale : 0.f;
        out[i] = o;
    }
}
__global__ __launch_bounds__(256) void act_dropout_bwd_k(const float* __restrict__ dout,
                                                         const float* __restrict__ x,
                                                         const uint8_t* __restrict__ keep, float p,
                                                         float slope, float* __restrict__ dx, long n) {
    const float dscale = 1.0f / (1.0f - p);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long)gridDim.x * blockDim.x) {
        float g = dout[i];
        if (keep) g = keep[i] ? g * dscale : 0.f;
        dx[i] = g * leaky_grad(x[i], slope);
    }
}

static inline int stream_grid(long total) {
    long g = cdiv(total, 256);
    if (g > 256L * 16) g = 256L * 16;
    if (g < 1) g = 1;
    return (int)g;
}
static inline int stat_blocks(long M) {
    long g = cdiv(M, 64);
    if (g > kMaxStatBlocks) g = kMaxStatBlocks;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace siv

using namespace siv;

extern "C" int siv_bn_stats(const float* y, long M, int C, float* mean, float* var,
                            float* running_mean, float* running_var, int64_t* num_batches_tracked,
                            float momentum, void* ws, size_t ws_bytes, siv_stream_t stream) {
    SIV_CHECK_ARG(y && mean && var, "bn_stats: null pointer");
    SIV_CHECK_ARG(M > 0 && C > 0, "bn_stats: bad shape M=%ld C=%d", M, C);
    SIV_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr),
                  "bn_stats: running_mean and running_var must both be given or both be NULL");
    const int nblk = stat_blocks(M);
    SIV_CHECK_WS((size_t)nblk * 2 * C * sizeof(float), ws_bytes, "bn_stats");
    hipStream_t s = (hipStream_t)stream;
    bn_stats_partial<<<nblk, 256, 512 * sizeof(float), s>>>(y, M, C, (float*)ws);
    SIV_CHECK_LAUNCH("bn_stats(partial)");
    bn_stats_finalize<<<(C + 3) / 4, 256, 0, s>>>(y, (const float*)ws, nblk, M, C, mean, var,
                                                      running_mean, running_var,
                                                      num_batches_tracked, momentum);
    SIV_CHECK_LAUNCH("bn_stats(finalize)");
    return SIV_OK;
}

extern "C" int siv_bn_act_fwd(const float* y, const float* mean, const float* var,
                              const float* gamma, const float* beta, const float* residual,
                              const uint8_t* keep, float p, float slope, int pool, float eps,
                              float* out, int N, int D, int H, int W, int C, siv_stream_t stream) {
    SIV_CHECK_ARG(y && mean && var && gamma && beta && out, "bn_act_fwd: null pointer");
    SIV_CHECK_ARG(N > 0 && D > 0 && H > 0 && W > 0 && C > 0, "bn_act_fwd: bad shape");
    SIV_CHECK_ARG(pool == 1 || pool == 2, "bn_act_fwd: pool must be 1 or 2 (got %d)", pool);
    SIV_CHECK_ARG(pool == 1 || (D % 2 == 0 && H % 2 == 0 && W % 2 == 0),
                  "bn_act_fwd: pool=2 needs even extents (got %d,%d,%d)", D, H, W);
    SIV_CHECK_ARG(p >= 0.f && p < 1.f, "bn_act_fwd: dropout p out of range");
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)N * (D / pool) * (H / pool) * (W / pool) * C;
    if (pool == 1)
        bn_act_fwd_k<1><<<stream_grid(total), 256, 0, s>>>(y, mean, var, gamma, beta, residual, keep, p,
                                                           slope, eps, out, N, D, H, W, C);
    else
        bn_act_fwd_k<2><<<stream_grid(total), 256, 0, s>>>(y, mean, var, gamma, beta, residual, keep, p,
                                                           slope, eps, out, N, D, H, W, C);
    SIV_CHECK_LAUNCH("bn_act_fwd");
    return SIV_OK;
}

extern "C" int siv_bn_act_bwd(const float* dout, const float* y, const float* mean, const float* var,
                              const float* gamma, const float* beta, const float* residual,
                              const uint8_t* keep, float p, float slope, int pool, float eps,
                              float* dy, float* dgamma, float* dbeta, float* dresidual,
                              int accumulate, int N, int D, int H, int W, int C, void* ws,
                              size_t ws_bytes, siv_stream_t stream) {
    SIV_CHECK_ARG(dout && y && mean && var && gamma && beta && dy, "bn_act_bwd: null pointer");
    SIV_CHECK_ARG(N > 0 && D > 0 && H > 0 && W > 0 && C > 0, "bn_act_bwd: bad shape");
    SIV_CHECK_ARG(pool == 1 || pool == 2, "bn_act_bwd: pool must be 1 or 2 (got %d)", pool);
    SIV_CHECK_ARG(pool == 1 || (D % 2 == 0 && H % 2 == 0 && W % 2 == 0),
                  "bn_act_bwd: pool=2 needs even extents");
    SIV_CHECK_ARG(dresidual == nullptr || residual != nullptr,
                  "bn_act_bwd: dresidual requested without residual");
    const long M = (long)N * D * H * W;
    const int nblk = stat_blocks(M);
    const size_t part_bytes = align256((size_t)nblk * 2 * C * sizeof(float));
    SIV_CHECK_WS(part_bytes + (size_t)2 * C * sizeof(float), ws_bytes, "bn_act_bwd");
    float* part = (float*)ws;
    float* sums = (float*)((char*)ws + part_bytes);
    hipStream_t s = (hipStream_t)stream;
    if (pool == 1)
        bn_bwd_partial<1><<<nblk, 256, 512 * sizeof(float), s>>>(dout, y, mean, var, gamma, beta,
                                                                 residual, keep, p, slope, eps, M, D,
                                                                 H, W, C, part);
    else
        bn_bwd_partial<2><<<nblk, 256, 512 * sizeof(float), s>>>(dout, y, mean, var, gamma, beta,
                                                                 residual, keep, p, slope, eps, M, D,
                                                                 H, W, C, part);
    SIV_CHECK_LAUNCH("bn_act_bwd(partial)");
    bn_bwd_finalize<<<(C + 3) / 4, 256, 0, s>>>(part, nblk, C, sums, dgamma, dbeta, accumulate);
    SIV_CHECK_LAUNCH("bn_act_bwd(finalize)");
    if (pool == 1)
        bn_bwd_apply<1><<<stream_grid(M * C), 256, 0, s>>>(dout, y, mean, var, gamma, beta, residual,
                                                           keep, p, slope, eps, sums, M, D, H, W, C,
                                                           dy, dresidual);
    else
        bn_bwd_apply<2><<<stream_grid(M * C), 256, 0, s>>>(dout, y, mean, var, gamma, beta, residual,
                                                           keep, p, slope, eps, sums, M, D, H, W, C,
                                                           dy, dresidual);
    SIV_CHECK_LAUNCH("bn_act_bwd(apply)");
    return SIV_OK;
}

extern "C" int siv_act_dropout_fwd(const float* x, const uint8_t* keep, float p, float slope,
                                   float* out, long n, siv_stream_t stream) {
    SIV_CHECK_ARG(x && out && n > 0, "act_dropout_fwd: bad argument");
    SIV_CHECK_ARG(p >= 0.f && p < 1.f, "act_dropout_fwd: dropout p out of range");
    act_dropout_fwd_k<<<stream_grid(n), 256, 0, (hipStream_t)stream>>>(x, keep, p, slope, out, n);
    SIV_CHECK_LAUNCH("act_dropout_fwd");
    return SIV_OK;
}

extern "C" int siv_act_dropout_bwd(const float* dout, const float* x, const uint8_t* keep, float p,
                                   float slope, float* dx, long n, siv_stream_t stream) {
    SIV_CHECK_ARG(dout && x && dx && n > 0, "act_dropout_bwd: bad argument");
    act_dropout_bwd_k<<<stream_grid(n), 256, 0, (hipStream_t)stream>>>(dout, x, keep, p, slope, dx, n);
    SIV_CHECK_LAUNCH("act_dropout_bwd");
    return SIV_OK;
}
