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
    y += (long)blockIdx.y * M * C;                       // group
    part += (long)blockIdx.y * gridDim.x * 2 * C;
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

// Groups: a batch may consist of `groups` equal parts that are normalised separately (each part is what the
// reference passes through the layer in a call of its own).  Group g owns rows [g*M, (g+1)*M) of the tensor,
// partial rows [g*prow, g*prow + nblk) and statistics [g][C]; the running statistics are updated once per group,
// in group order, as by consecutive calls.
// one wave per channel: lanes stride over the per-block partials, fixed-order wave reduction
__global__ __launch_bounds__(256) void bn_stats_finalize(const float* __restrict__ y,
                                                         const float* __restrict__ part, int nblk,
                                                         long M, int C, float* __restrict__ mean,
                                                         float* __restrict__ var,
                                                         float* __restrict__ rmean,
                                                         float* __restrict__ rvar,
                                                         int64_t* __restrict__ nbt, float momentum,
                                                         int groups, long prow, long ystride) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += groups;
    if (c >= C) return;
    for (int g = 0; g < groups; ++g) {
        const float* pg = part + (long)g * prow * 2 * C;
        double s1 = 0.0, s2 = 0.0;
        for (int b = lane; b < nblk; b += 64) {
            s1 += (double)pg[((long)b * 2 + 0) * C + c];
            s2 += (double)pg[((long)b * 2 + 1) * C + c];
        }
        s1 = wave_sum_d(s1);
        s2 = wave_sum_d(s2);
        if (lane != 0) continue;
        const double ms = s1 / (double)M;
        double v = s2 / (double)M - ms * ms;
        if (v < 0.0) v = 0.0;
        const double mu = (y ? (double)y[(long)g * ystride + c] : 0.0) + ms;     // y: the shift the partials were taken against
        mean[g * C + c] = (float)mu;
        var[g * C + c] = (float)v;
        if (rmean) {
            const double unb = (M > 1) ? v * ((double)M / (double)(M - 1)) : v;
            rmean[c] = (float)((1.0 - (double)momentum) * (double)rmean[c] + (double)momentum * mu);
            rvar[c] = (float)((1.0 - (double)momentum) * (double)rvar[c] + (double)momentum * unb);
        }
    }
}

// Same result for C % 4 == 0 with many partial rows: one 256-thread block per channel quad, each
// thread sums the float4 of every 256th row in double, then a fixed-order block reduction.
__global__ __launch_bounds__(256) void bn_stats_finalize_q(const float* __restrict__ y,
                                                           const float* __restrict__ part, int nblk,
                                                           long M, int C, float* __restrict__ mean,
                                                           float* __restrict__ var,
                                                           float* __restrict__ rmean,
                                                           float* __restrict__ rvar,
                                                           int64_t* __restrict__ nbt, float momentum) {
    __shared__ double sh[4][8];
    const int q = blockIdx.x;
    if (q == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
    double a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) {
        const float4 p1 = *reinterpret_cast<const float4*>(part + ((long)b * 2 + 0) * C + 4 * q);
        const float4 p2 = *reinterpret_cast<const float4*>(part + ((long)b * 2 + 1) * C + 4 * q);
        a[0] += (double)p1.x; a[1] += (double)p1.y; a[2] += (double)p1.z; a[3] += (double)p1.w;
        a[4] += (double)p2.x; a[5] += (double)p2.y; a[6] += (double)p2.z; a[7] += (double)p2.w;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = wave_sum_d(a[j]);
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int j = 0; j < 8; ++j) sh[threadIdx.x >> 6][j] = a[j];
    __syncthreads();
    if (threadIdx.x < 4) {
        const int j = threadIdx.x, c = 4 * q + j;
        const double s1 = (sh[0][j] + sh[1][j]) + (sh[2][j] + sh[3][j]);
        const double s2 = (sh[0][4 + j] + sh[1][4 + j]) + (sh[2][4 + j] + sh[3][4 + j]);
        const double ms = s1 / (double)M;
        double v = s2 / (double)M - ms * ms;
        if (v < 0.0) v = 0.0;
        const double mu = (y ? (double)y[c] : 0.0) + ms;
        mean[c] = (float)mu;
        var[c] = (float)v;
        if (rmean) {
            const double unb = (M > 1) ? v * ((double)M / (double)(M - 1)) : v;
            rmean[c] = (float)((1.0 - (double)momentum) * (double)rmean[c] + (double)momentum * mu);
            rvar[c] = (float)((1.0 - (double)momentum) * (double)rvar[c] + (double)momentum * unb);
        }
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
// Groups as above: sums[g][2][C]; dgamma / dbeta receive the groups' contributions one after the other.
__global__ __launch_bounds__(256) void bn_bwd_finalize(const float* __restrict__ part, int nblk, int C,
                                                       float* __restrict__ sums,
                                                       float* __restrict__ dgamma,
                                                       float* __restrict__ dbeta, int accumulate,
                                                       int groups, long prow) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    for (int g = 0; g < groups; ++g) {
        const float* pg = part + (long)g * prow * 2 * C;
        double s1 = 0.0, s2 = 0.0;
        for (int b = lane; b < nblk; b += 64) {
            s1 += (double)pg[((long)b * 2 + 0) * C + c];
            s2 += (double)pg[((long)b * 2 + 1) * C + c];
        }
        s1 = wave_sum_d(s1);
        s2 = wave_sum_d(s2);
        if (lane != 0) continue;
        sums[(long)g * 2 * C + c] = (float)s1;
        sums[(long)g * 2 * C + C + c] = (float)s2;
        const bool acc = accumulate || g > 0;
        if (dbeta) dbeta[c] = acc ? dbeta[c] + (float)s1 : (float)s1;
        if (dgamma) dgamma[c] = acc ? dgamma[c] + (float)s2 : (float)s2;
    }
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

// ---------------------------------------------------------------------------------------
// float4 path.  Requires C % 4 == 0 and (C/4) a divisor of 256 (C = 4 .. 1024, power of two):
// a thread of a 256-thread block then owns the same four channels in every iteration of a
// grid-stride loop over channel quads, so per-channel coefficients live in registers and
// no index arithmetic beyond an add is needed per 16-byte access.
// ---------------------------------------------------------------------------------------
typedef float v4f __attribute__((ext_vector_type(4)));
#ifndef SIV_NT
#define SIV_NT 3   // streaming tensors: non-temporal 16-byte loads (1) and stores (2), measured +4-10 % on these kernels
#endif
__device__ __forceinline__ float4 ld4(const float* __restrict__ p, long q) {
#if SIV_NT & 1
    const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p) + q);
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return reinterpret_cast<const float4*>(p)[q];
#endif
}
__device__ __forceinline__ void st4(float* __restrict__ p, long q, float4 v) {
#if SIV_NT & 2
    v4f t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(p) + q);
#else
    reinterpret_cast<float4*>(p)[q] = v;
#endif
}
__device__ __forceinline__ float4 f4(float v) { return make_float4(v, v, v, v); }
__device__ __forceinline__ float4 operator+(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 operator-(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 operator*(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) {
    return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
__device__ __forceinline__ float4 leaky4(float4 z, float s) {
    return make_float4(leaky(z.x, s), leaky(z.y, s), leaky(z.z, s), leaky(z.w, s));
}
__device__ __forceinline__ float4 leaky_grad4(float4 z, float s) {
    return make_float4(leaky_grad(z.x, s), leaky_grad(z.y, s), leaky_grad(z.z, s), leaky_grad(z.w, s));
}
__device__ __forceinline__ float4 keep4(const uint8_t* __restrict__ keep, long q, float4 v, float dscale) {
    const uchar4 k = reinterpret_cast<const uchar4*>(keep)[q];
    return make_float4(k.x ? v.x * dscale : 0.f, k.y ? v.y * dscale : 0.f, k.z ? v.z * dscale : 0.f,
                       k.w ? v.w * dscale : 0.f);
}

// four channels held as bf16 (8 bytes): the storage format of activations kept for the backward pass
__device__ __forceinline__ void st_bf4(uint16_t* __restrict__ p, long q, float4 v) {
    reinterpret_cast<uint2*>(p)[q] = make_uint2(bf16_rne(v.x) | (bf16_rne(v.y) << 16), bf16_rne(v.z) | (bf16_rne(v.w) << 16));
}
__device__ __forceinline__ float4 ld_bf4(const uint16_t* __restrict__ p, long q) {
    const uint2 u = reinterpret_cast<const uint2*>(p)[q];
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
}
// the saved convolution output: fp32, or (YBF) its bf16 copy
template <bool YBF>
__device__ __forceinline__ float4 ld_y(const float* __restrict__ y, long q) {
    if constexpr (YBF) return ld_bf4(reinterpret_cast<const uint16_t*>(y), q);
    else return ld4(y, q);
}

// per-thread coefficients of its four channels
struct Coef {
    float4 a, b, mean, invstd;
};
__device__ __forceinline__ Coef load_coef(const float* __restrict__ mean, const float* __restrict__ var,
                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                          float eps, int c4) {
    Coef k;
    const float4 v = ld4(var, c4), g = ld4(gamma, c4), be = ld4(beta, c4);
    k.mean = ld4(mean, c4);
    k.invstd = make_float4(1.0f / sqrtf(v.x + eps), 1.0f / sqrtf(v.y + eps), 1.0f / sqrtf(v.z + eps),
                           1.0f / sqrtf(v.w + eps));
    // same association as the scalar kernels: a = gamma / sqrt(var + eps), b = beta - mean * a
    k.a = make_float4(g.x / sqrtf(v.x + eps), g.y / sqrtf(v.y + eps), g.z / sqrtf(v.z + eps),
                      g.w / sqrtf(v.w + eps));
    k.b = make_float4(be.x - k.mean.x * k.a.x, be.y - k.mean.y * k.a.y, be.z - k.mean.z * k.a.z,
                      be.w - k.mean.w * k.a.w);
    return k;
}

// Sum the per-thread float4 pairs of the threads that own the same channel quad and write
// the block's partial [2][C] row.  sh: 2 * 256 float4.
__device__ __forceinline__ void block_reduce_quads(float4 s1, float4 s2, int C4, int C,
                                                   float* __restrict__ part, float4* sh) {
    sh[threadIdx.x] = s1;
    sh[256 + threadIdx.x] = s2;
    __syncthreads();
    if ((int)threadIdx.x < C4) {
        float4 a1 = f4(0.f), a2 = f4(0.f);
        for (int t = threadIdx.x; t < 256; t += C4) {
            a1 = a1 + sh[t];
            a2 = a2 + sh[256 + t];
        }
        st4(part + ((long)blockIdx.x * 2 + 0) * C, threadIdx.x, a1);
        st4(part + ((long)blockIdx.x * 2 + 1) * C, threadIdx.x, a2);
    }
}

__global__ __launch_bounds__(256) void bn_stats_partial_v(const float* __restrict__ y, long Q, int C,
                                                          float* __restrict__ part) {
    __shared__ float4 sh[512];
    y += (long)blockIdx.y * Q * 4;                       // group: Q quads each
    part += (long)blockIdx.y * gridDim.x * 2 * C;
    const int C4 = C >> 2;
    const float4 shift = ld4(y, threadIdx.x % C4);
    float4 s1 = f4(0.f), s2 = f4(0.f);
    const long T = (long)gridDim.x * 256;
    long q = (long)blockIdx.x * 256 + threadIdx.x;
    for (; q + 3 * T < Q; q += 4 * T) {
        const float4 v0 = ld4(y, q), v1 = ld4(y, q + T), v2 = ld4(y, q + 2 * T), v3 = ld4(y, q + 3 * T);
        const float4 d0 = v0 - shift, d1 = v1 - shift, d2 = v2 - shift, d3 = v3 - shift;
        s1 = s1 + ((d0 + d1) + (d2 + d3));
        s2 = fma4(d0, d0, s2);
        s2 = fma4(d1, d1, s2);
        s2 = fma4(d2, d2, s2);
        s2 = fma4(d3, d3, s2);
    }
    for (; q < Q; q += T) {
        const float4 d = ld4(y, q) - shift;
        s1 = s1 + d;
        s2 = fma4(d, d, s2);
    }
    block_reduce_quads(s1, s2, C4, C, part, sh);
}

// decode an output-space quad index of the pooled tensor into the quad index of its first
// child in the full-resolution tensor; C4 is a power of two
struct PoolGeom {
    int Do, Ho, Wo, H, W, C4, log2C4;
};
__device__ __forceinline__ long pool_child0(long oq, const PoolGeom& g, long* strideW, long* strideH,
                                            long* strideD) {
    const unsigned vox = (unsigned)(oq >> g.log2C4);
    const int cq = (int)(oq & (g.C4 - 1));
    const unsigned ww = vox % g.Wo, t1 = vox / g.Wo;
    const unsigned hh = t1 % g.Ho, t2 = t1 / g.Ho;
    const unsigned dd = t2 % g.Do, n = t2 / g.Do;
    *strideW = g.C4;
    *strideH = (long)g.W * g.C4;
    *strideD = (long)g.H * g.W * g.C4;
    return ((((long)n * (2 * g.Do) + 2 * dd) * g.H + 2 * hh) * g.W + 2 * ww) * g.C4 + cq;
}

template <int POOL>
__global__ __launch_bounds__(256) void bn_act_fwd_v(const float* __restrict__ y,
                                                    const float* __restrict__ mean,
                                                    const float* __restrict__ var,
                                                    const float* __restrict__ gamma,
                                                    const float* __restrict__ beta,
                                                    const float* __restrict__ res,
                                                    const uint8_t* __restrict__ keep, float p,
                                                    float slope, float eps, float* __restrict__ out,
                                                    long OQ, PoolGeom g, uint16_t* __restrict__ ybf,
                                                    uint16_t* __restrict__ obf) {
    // ybf / obf (optional): bf16 copies of the input y and of the output, written in the same pass -- what the
    // backward pass keeps when activations are stored as bf16
    {   // group blockIdx.y: OQ output quads (8 OQ input quads when pooled), its own statistics
        const long go = (long)blockIdx.y * OQ * 4, gi = go * (POOL == 2 ? 8 : 1);
        y += gi, out += go, mean += blockIdx.y * g.C4 * 4, var += blockIdx.y * g.C4 * 4;
        if (res) res += gi;
        if (keep) keep += go;
        if (ybf) ybf += gi;
        if (obf) obf += go;
    }
    const Coef k = load_coef(mean, var, gamma, beta, eps, threadIdx.x % g.C4);
    const float dscale = 1.0f / (1.0f - p);
    const long T = (long)gridDim.x * 256;
    long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (POOL == 1) {
        for (; q + T < OQ; q += 2 * T) {
            const float4 y0 = ld4(y, q), y1 = ld4(y, q + T);
            float4 z0 = fma4(y0, k.a, k.b), z1 = fma4(y1, k.a, k.b);
            if (res) {
                z0 = z0 + ld4(res, q);
                z1 = z1 + ld4(res, q + T);
            }
            float4 o0 = leaky4(z0, slope), o1 = leaky4(z1, slope);
            if (keep) {
                o0 = keep4(keep, q, o0, dscale);
                o1 = keep4(keep, q + T, o1, dscale);
            }
            st4(out, q, o0);
            st4(out, q + T, o1);
            if (ybf) st_bf4(ybf, q, y0), st_bf4(ybf, q + T, y1);
            if (obf) st_bf4(obf, q, o0), st_bf4(obf, q + T, o1);
        }
        for (; q < OQ; q += T) {
            const float4 y0 = ld4(y, q);
            float4 z = fma4(y0, k.a, k.b);
            if (res) z = z + ld4(res, q);
            float4 o = leaky4(z, slope);
            if (keep) o = keep4(keep, q, o, dscale);
            st4(out, q, o);
            if (ybf) st_bf4(ybf, q, y0);
            if (obf) st_bf4(obf, q, o);
        }
    } else {
        for (; q < OQ; q += T) {
            long sw, shh, sd;
            const long j0 = pool_child0(q, g, &sw, &shh, &sd);
            float4 v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = ld4(y, j0 + (e >> 2) * sd + ((e >> 1) & 1) * shh + (e & 1) * sw);
            float4 acc = f4(0.f);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float4 z = fma4(v[e], k.a, k.b);
                if (res) z = z + ld4(res, j0 + (e >> 2) * sd + ((e >> 1) & 1) * shh + (e & 1) * sw);
                acc = acc + leaky4(z, slope);
            }
            if (ybf)
#pragma unroll
                for (int e = 0; e < 8; ++e) st_bf4(ybf, j0 + (e >> 2) * sd + ((e >> 1) & 1) * shh + (e & 1) * sw, v[e]);
            float4 o = acc * f4(0.125f);
            if (keep) o = keep4(keep, q, o, dscale);
            st4(out, q, o);
            if (obf) st_bf4(obf, q, o);
        }
    }
}

// One backward element quad: dz and xhat for input quad j with output-space gradient g.
__device__ __forceinline__ float4 bwd_dz(float4 g, float4 yv, const float* __restrict__ res, long j,
                                         const Coef& k, float slope, float4* xh) {
    float4 z = fma4(yv, k.a, k.b);
    if (res) z = z + ld4(res, j);
    *xh = (yv - k.mean) * k.invstd;
    return g * leaky_grad4(z, slope);
}

// APPLY = 0: per-block partial sums of dz and dz*xhat.  APPLY = 1: dy (and dres) from the sums.
template <int POOL, int APPLY, bool YBF = false>
__global__ __launch_bounds__(256) void bn_bwd_v(const float* __restrict__ dout,
                                                const float* __restrict__ y,
                                                const float* __restrict__ mean,
                                                const float* __restrict__ var,
                                                const float* __restrict__ gamma,
                                                const float* __restrict__ beta,
                                                const float* __restrict__ res,
                                                const uint8_t* __restrict__ keep, float p, float slope,
                                                float eps, const float* __restrict__ sums, float invM,
                                                long OQ, PoolGeom g, int C, float* __restrict__ part,
                                                float* __restrict__ dy, float* __restrict__ dres) {
    __shared__ float4 sh[APPLY ? 1 : 512];
    {   // group blockIdx.y (see bn_act_fwd_v): its own statistics, sums and partial rows
        const long go = (long)blockIdx.y * OQ * 4, gi = go * (POOL == 2 ? 8 : 1);
        dout += go, y += YBF ? gi / 2 : gi, mean += blockIdx.y * C, var += blockIdx.y * C;   // y: bf16 = half a float each
        if (keep) keep += go;
        if (res) res += gi;
        if (APPLY) {
            sums += (long)blockIdx.y * 2 * C, dy += gi;
            if (dres) dres += gi;
        } else {
            part += (long)blockIdx.y * gridDim.x * 2 * C;
        }
    }
    const int c4 = threadIdx.x % g.C4;
    const Coef k = load_coef(mean, var, gamma, beta, eps, c4);
    float4 k1 = f4(0.f), k2 = f4(0.f);
    if (APPLY) {
        k1 = ld4(sums, c4) * f4(invM);
        k2 = ld4(sums + C, c4) * f4(invM);
    }
    const float dscale = 1.0f / (1.0f - p);
    float4 s1 = f4(0.f), s2 = f4(0.f);
    const long T = (long)gridDim.x * 256;
    long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (POOL == 1) {
        for (; q + T < OQ; q += 2 * T) {
            float4 g0 = ld4(dout, q), g1 = ld4(dout, q + T);
            const float4 y0 = ld_y<YBF>(y, q), y1 = ld_y<YBF>(y, q + T);
            if (keep) {
                g0 = keep4(keep, q, g0, dscale);
                g1 = keep4(keep, q + T, g1, dscale);
            }
            float4 x0, x1;
            const float4 d0 = bwd_dz(g0, y0, res, q, k, slope, &x0);
            const float4 d1 = bwd_dz(g1, y1, res, q + T, k, slope, &x1);
            if (APPLY) {
                st4(dy, q, k.a * (d0 - k1 - x0 * k2));
                st4(dy, q + T, k.a * (d1 - k1 - x1 * k2));
                if (dres) {
                    st4(dres, q, d0);
                    st4(dres, q + T, d1);
                }
            } else {
                s1 = s1 + (d0 + d1);
                s2 = fma4(d0, x0, s2);
                s2 = fma4(d1, x1, s2);
            }
        }
        for (; q < OQ; q += T) {
            float4 g0 = ld4(dout, q);
            if (keep) g0 = keep4(keep, q, g0, dscale);
            float4 x0;
            const float4 d0 = bwd_dz(g0, ld_y<YBF>(y, q), res, q, k, slope, &x0);
            if (APPLY) {
                st4(dy, q, k.a * (d0 - k1 - x0 * k2));
                if (dres) st4(dres, q, d0);
            } else {
                s1 = s1 + d0;
                s2 = fma4(d0, x0, s2);
            }
        }
    } else {
        for (; q < OQ; q += T) {
            long sw, shh, sd;
            const long j0 = pool_child0(q, g, &sw, &shh, &sd);
            float4 go = ld4(dout, q);
            if (keep) go = keep4(keep, q, go, dscale);
            go = go * f4(0.125f);
            float4 v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = ld_y<YBF>(y, j0 + (e >> 2) * sd + ((e >> 1) & 1) * shh + (e & 1) * sw);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const long j = j0 + (e >> 2) * sd + ((e >> 1) & 1) * shh + (e & 1) * sw;
                float4 xh;
                const float4 d = bwd_dz(go, v[e], res, j, k, slope, &xh);
                if (APPLY) {
                    st4(dy, j, k.a * (d - k1 - xh * k2));
                    if (dres) st4(dres, j, d);
                } else {
                    s1 = s1 + d;
                    s2 = fma4(d, xh, s2);
                }
            }
        }
    }
    if (!APPLY) block_reduce_quads(s1, s2, g.C4, C, part, sh);
}

static inline bool aligned16(const void* p) { return p == nullptr || ((uintptr_t)p & 15) == 0; }
// float4 path applicable?
static inline bool vec_channels(int C) { return C % 4 == 0 && C / 4 <= 256 && 256 % (C / 4) == 0; }
static inline int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}
// Grids of the float4 kernels.  A grid-stride loop whose stride is a power of two (2048 blocks x 4 KB =
// 8 MB) keeps the two to four loads a thread has in flight on the same memory channels: measured
// 4.8 TB/s where an odd block count or one contiguous pass reaches 6.0-6.4 TB/s.  So the kernels that
// write per-block partials (statistics, backward sums) run an odd number of blocks, and the apply
// kernels get one block per 256 x per_thread quads, no cap.
constexpr int kVecReduceBlocks = 1531;             // odd, <= kMaxStatBlocks (partials are sized by that)
static inline int vec_grid(long quads, int per_thread) {
    long g = cdiv(quads, 256L * per_thread);
    if (g > kVecReduceBlocks) g = kVecReduceBlocks;
    if (g < 1) g = 1;
    return (int)g;
}
static inline int vec_grid_apply(long quads, int per_thread) {
    long g = cdiv(quads, 256L * per_thread);
    if (g > 0x7fffffffL) g = 0x7fffffffL;
    if (g < 1) g = 1;
    return (int)g;
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

// Rows b, b + G, b + 2G, ... of part[nrows][cols] summed (in double, fixed order) into row b, in place: a row
// is read by one block only, so nothing else is needed to keep the blocks apart.  The finalize kernels give a
// wave or a block a few channels and walk every row for them: 16 bytes out of each 512-byte row, so with
// thousands of rows they are bound by the cache lines a CU can take in (28 us for 9600 rows x 64 channels on
// 16 CUs).  Folded onto G = 256 rows first, whole rows per block, the two launches take ~10 us.
constexpr int FOLD_ROWS = 256, FOLD_MIN_ROWS = 1024;
__global__ __launch_bounds__(256) void rows_fold_k(float* __restrict__ part, int nrows, int cols) {
    part += (long)blockIdx.y * nrows * cols;             // group: its own nrows rows
    const long G = gridDim.x;
    for (int col = threadIdx.x; col < cols; col += 256) {
        double a[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = 0.0;
        long r = blockIdx.x;
        for (; r + 7 * G < nrows; r += 8 * G) {
            float p[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) p[k] = part[(r + k * G) * cols + col];
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] += (double)p[k];
        }
        for (; r < nrows; r += G) a[0] += (double)part[r * cols + col];
        part[(long)blockIdx.x * cols + col] =
            (float)(((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7])));
    }
}
// part is workspace of the caller's: folding overwrites its first FOLD_ROWS rows
static int fold_rows(const float* part, int nblk, int cols, hipStream_t s, int groups = 1) {
    if (nblk < FOLD_MIN_ROWS) return nblk;
    rows_fold_k<<<dim3(FOLD_ROWS, groups), 256, 0, s>>>(const_cast<float*>(part), nblk, cols);
    return FOLD_ROWS;
}

}  // namespace siv

using namespace siv;

extern "C" int siv_bn_stats_grouped(const float* y, long M, int C, int groups, float* mean, float* var,
                                    float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                    float momentum, void* ws, size_t ws_bytes, siv_stream_t stream) {
    SIV_CHECK_ARG(y && mean && var, "bn_stats: null pointer");
    SIV_CHECK_ARG(M > 0 && C > 0 && groups > 0 && groups <= 65535, "bn_stats: bad shape M=%ld C=%d groups=%d", M, C, groups);
    SIV_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr),
                  "bn_stats: running_mean and running_var must both be given or both be NULL");
    const bool vec = vec_channels(C) && aligned16(y) && aligned16(ws);
    const int nblk = vec ? vec_grid(M * (C / 4), 8) : stat_blocks(M);
    SIV_CHECK_WS((size_t)groups * nblk * 2 * C * sizeof(float), ws_bytes, "bn_stats");
    hipStream_t s = (hipStream_t)stream;
    if (vec)
        bn_stats_partial_v<<<dim3(nblk, groups), 256, 0, s>>>(y, M * (C / 4), C, (float*)ws);
    else
        bn_stats_partial<<<dim3(nblk, groups), 256, 512 * sizeof(float), s>>>(y, M, C, (float*)ws);
    SIV_CHECK_LAUNCH("bn_stats(partial)");
    const int nfold = fold_rows((const float*)ws, nblk, 2 * C, s, groups);
    bn_stats_finalize<<<(C + 3) / 4, 256, 0, s>>>(y, (const float*)ws, nfold, M, C, mean, var, running_mean,
                                                      running_var, num_batches_tracked, momentum, groups, nblk,
                                                      M * C);
    SIV_CHECK_LAUNCH("bn_stats(finalize)");
    return SIV_OK;
}
extern "C" int siv_bn_stats(const float* y, long M, int C, float* mean, float* var,
                            float* running_mean, float* running_var, int64_t* num_batches_tracked,
                            float momentum, void* ws, size_t ws_bytes, siv_stream_t stream) {
    return siv_bn_stats_grouped(y, M, C, 1, mean, var, running_mean, running_var, num_batches_tracked, momentum, ws,
                                ws_bytes, stream);
}

namespace siv {
// Finish batch statistics from per-block partial sums [nblk][2][C] of (value - shift[c]); shift may be
// NULL (zero).  Used by the convolution that emits those sums from its epilogue.
int bn_finalize_launch(const float* shift, const float* part, int nblk, long M, int C, float* mean,
                       float* var, float* running_mean, float* running_var, int64_t* nbt,
                       float momentum, hipStream_t s) {
    nblk = fold_rows(part, nblk, 2 * C, s);
    if (C % 4 == 0 && nblk >= 512 && (((uintptr_t)part) & 15) == 0)
        bn_stats_finalize_q<<<C / 4, 256, 0, s>>>(shift, part, nblk, M, C, mean, var, running_mean,
                                                  running_var, nbt, momentum);
    else
        bn_stats_finalize<<<(C + 3) / 4, 256, 0, s>>>(shift, part, nblk, M, C, mean, var, running_mean,
                                                          running_var, nbt, momentum, 1, 0, 0);
    SIV_CHECK_LAUNCH("bn_stats(finalize)");
    return SIV_OK;
}
// for kernels in other files that produce the [block][2][C] partials of (sum dz, sum dz*xhat) themselves
int bn_bwd_finalize_launch(const float* part, int nblk, int C, float* sums, float* dgamma, float* dbeta,
                           int accumulate, hipStream_t s) {
    nblk = fold_rows(part, nblk, 2 * C, s);
    bn_bwd_finalize<<<(C + 3) / 4, 256, 0, s>>>(part, nblk, C, sums, dgamma, dbeta, accumulate, 1, 0);
    SIV_CHECK_LAUNCH("bn_bwd(finalize)");
    return SIV_OK;
}
}  // namespace siv

// ybf / obf: optional bf16 copies of y and of the output (float4 kernels only)
static int bn_act_fwd_impl(const float* y, const float* mean, const float* var,
                           const float* gamma, const float* beta, const float* residual,
                           const uint8_t* keep, float p, float slope, int pool, float eps,
                           float* out, int N, int D, int H, int W, int C, int groups,
                           uint16_t* ybf, uint16_t* obf, siv_stream_t stream) {
    SIV_CHECK_ARG(y && mean && var && gamma && beta && out, "bn_act_fwd: null pointer");
    SIV_CHECK_ARG(N > 0 && D > 0 && H > 0 && W > 0 && C > 0, "bn_act_fwd: bad shape");
    SIV_CHECK_ARG(groups > 0 && groups <= 65535 && N % groups == 0,
                  "bn_act_fwd: the batch (%d) must be a multiple of the number of groups (%d)", N, groups);
    SIV_CHECK_ARG(pool == 1 || pool == 2, "bn_act_fwd: pool must be 1 or 2 (got %d)", pool);
    SIV_CHECK_ARG(pool == 1 || (D % 2 == 0 && H % 2 == 0 && W % 2 == 0),
                  "bn_act_fwd: pool=2 needs even extents (got %d,%d,%d)", D, H, W);
    SIV_CHECK_ARG(p >= 0.f && p < 1.f, "bn_act_fwd: dropout p out of range");
    hipStream_t s = (hipStream_t)stream;
    const int Ng = N / groups;                                    // samples per group
    const long total = (long)Ng * (D / pool) * (H / pool) * (W / pool) * C;      // output elements per group
    if (vec_channels(C) && (long)Ng * D * H * W < (1L << 31) && aligned16(y) && aligned16(out) &&
        aligned16(residual) && aligned16(mean) && aligned16(var) && aligned16(gamma) && aligned16(beta) &&
        ((uintptr_t)keep & 3) == 0) {
        const PoolGeom g = {D / pool, H / pool, W / pool, H, W, C / 4, ilog2(C / 4)};
        const long OQ = total / 4;
        SIV_CHECK_ARG((((uintptr_t)ybf | (uintptr_t)obf) & 7) == 0, "bn_act_fwd: the bf16 copies must be 8-byte aligned");
        if (pool == 1)
            bn_act_fwd_v<1><<<dim3(vec_grid_apply(OQ, 4), groups), 256, 0, s>>>(y, mean, var, gamma, beta, residual,
                                                                             keep, p, slope, eps, out, OQ, g, ybf, obf);
        else
            bn_act_fwd_v<2><<<dim3(vec_grid_apply(OQ, 1), groups), 256, 0, s>>>(y, mean, var, gamma, beta, residual,
                                                                             keep, p, slope, eps, out, OQ, g, ybf, obf);
        SIV_CHECK_LAUNCH("bn_act_fwd");
        return SIV_OK;
    }
    SIV_CHECK_ARG(!ybf && !obf, "bn_act_fwd: bf16 copies need C = 4..1024 a power of two and 16-byte aligned tensors "
                                "(ask siv_bn_act_bf16_supported)");
    const long in_total = (long)Ng * D * H * W * C;
    for (int gi = 0; gi < groups; ++gi) {                         // scalar kernels: one launch per group
        const float* yg = y + gi * in_total;
        const float* rg = residual ? residual + gi * in_total : nullptr;
        const uint8_t* kg = keep ? keep + gi * total : nullptr;
        if (pool == 1)
            bn_act_fwd_k<1><<<stream_grid(total), 256, 0, s>>>(yg, mean + gi * C, var + gi * C, gamma, beta, rg, kg, p,
                                                               slope, eps, out + gi * total, Ng, D, H, W, C);
        else
            bn_act_fwd_k<2><<<stream_grid(total), 256, 0, s>>>(yg, mean + gi * C, var + gi * C, gamma, beta, rg, kg, p,
                                                               slope, eps, out + gi * total, Ng, D, H, W, C);
        SIV_CHECK_LAUNCH("bn_act_fwd");
    }
    return SIV_OK;
}
extern "C" int siv_bn_act_fwd_grouped(const float* y, const float* mean, const float* var,
                                      const float* gamma, const float* beta, const float* residual,
                                      const uint8_t* keep, float p, float slope, int pool, float eps,
                                      float* out, int N, int D, int H, int W, int C, int groups,
                                      siv_stream_t stream) {
    return bn_act_fwd_impl(y, mean, var, gamma, beta, residual, keep, p, slope, pool, eps, out, N, D, H, W, C, groups,
                           nullptr, nullptr, stream);
}
extern "C" int siv_bn_act_bf16_supported(int C) { return vec_channels(C) ? 1 : 0; }
extern "C" int siv_bn_act_fwd_bf16(const float* y, const float* mean, const float* var,
                                   const float* gamma, const float* beta, const float* residual,
                                   const uint8_t* keep, float p, float slope, int pool, float eps,
                                   float* out, void* y_bf16, void* out_bf16, int N, int D, int H, int W, int C,
                                   int groups, siv_stream_t stream) {
    return bn_act_fwd_impl(y, mean, var, gamma, beta, residual, keep, p, slope, pool, eps, out, N, D, H, W, C, groups,
                           (uint16_t*)y_bf16, (uint16_t*)out_bf16, stream);
}
extern "C" int siv_bn_act_fwd(const float* y, const float* mean, const float* var,
                              const float* gamma, const float* beta, const float* residual,
                              const uint8_t* keep, float p, float slope, int pool, float eps,
                              float* out, int N, int D, int H, int W, int C, siv_stream_t stream) {
    return siv_bn_act_fwd_grouped(y, mean, var, gamma, beta, residual, keep, p, slope, pool, eps, out, N, D, H, W, C,
                                  1, stream);
}

// ybf: y points at the bf16 copy of the convolution output (float4 kernels only)
static int bn_act_bwd_impl(const float* dout, const float* y, bool ybf, const float* mean, const float* var,
                           const float* gamma, const float* beta, const float* residual,
                           const uint8_t* keep, float p, float slope, int pool, float eps,
                           float* dy, float* dgamma, float* dbeta, float* dresidual,
                           int accumulate, int N, int D, int H, int W, int C, int groups, void* ws,
                           size_t ws_bytes, siv_stream_t stream) {
    SIV_CHECK_ARG(dout && y && mean && var && gamma && beta && dy, "bn_act_bwd: null pointer");
    SIV_CHECK_ARG(N > 0 && D > 0 && H > 0 && W > 0 && C > 0, "bn_act_bwd: bad shape");
    SIV_CHECK_ARG(groups > 0 && groups <= 65535 && N % groups == 0,
                  "bn_act_bwd: the batch (%d) must be a multiple of the number of groups (%d)", N, groups);
    SIV_CHECK_ARG(pool == 1 || pool == 2, "bn_act_bwd: pool must be 1 or 2 (got %d)", pool);
    SIV_CHECK_ARG(pool == 1 || (D % 2 == 0 && H % 2 == 0 && W % 2 == 0),
                  "bn_act_bwd: pool=2 needs even extents");
    SIV_CHECK_ARG(dresidual == nullptr || residual != nullptr,
                  "bn_act_bwd: dresidual requested without residual");
    const int Ng = N / groups;
    const long M = (long)Ng * D * H * W;                          // rows per group
    const bool vec = vec_channels(C) && M < (1L << 31) && aligned16(dout) && aligned16(y) && aligned16(dy) &&
                     aligned16(residual) && aligned16(dresidual) && aligned16(mean) && aligned16(var) &&
                     aligned16(gamma) && aligned16(beta) && aligned16(ws) && ((uintptr_t)keep & 3) == 0;
    SIV_CHECK_ARG(vec || !ybf, "bn_act_bwd: a bf16 y needs C = 4..1024 a power of two and 16-byte aligned tensors");
    const long OQ = (long)Ng * (D / pool) * (H / pool) * (W / pool) * (C / 4);   // loop extent of the float4 path
    const int nblk = vec ? vec_grid(OQ, pool == 1 ? 8 : 2) : stat_blocks(M);
    const size_t part_bytes = align256((size_t)groups * nblk * 2 * C * sizeof(float));
    SIV_CHECK_WS(part_bytes + (size_t)groups * 2 * C * sizeof(float), ws_bytes, "bn_act_bwd");
    float* part = (float*)ws;
    float* sums = (float*)((char*)ws + part_bytes);
    hipStream_t s = (hipStream_t)stream;
    if (vec) {
        const PoolGeom g = {D / pool, H / pool, W / pool, H, W, C / 4, ilog2(C / 4)};
        const float invM = 1.0f / (float)M;
#define SIV_BWD_V(POOL, APPLY, GRID, ...)                                                        \
    (ybf ? bn_bwd_v<POOL, APPLY, true><<<GRID, 256, 0, s>>>(__VA_ARGS__)                          \
         : bn_bwd_v<POOL, APPLY, false><<<GRID, 256, 0, s>>>(__VA_ARGS__))
        if (pool == 1)
            SIV_BWD_V(1, 0, dim3(nblk, groups), dout, y, mean, var, gamma, beta, residual, keep, p, slope, eps, nullptr,
                      invM, OQ, g, C, part, nullptr, nullptr);
        else
            SIV_BWD_V(2, 0, dim3(nblk, groups), dout, y, mean, var, gamma, beta, residual, keep, p, slope, eps, nullptr,
                      invM, OQ, g, C, part, nullptr, nullptr);
        SIV_CHECK_LAUNCH("bn_act_bwd(partial)");
        const int nfold = fold_rows(part, nblk, 2 * C, s, groups);
        bn_bwd_finalize<<<(C + 3) / 4, 256, 0, s>>>(part, nfold, C, sums, dgamma, dbeta, accumulate, groups, nblk);
        SIV_CHECK_LAUNCH("bn_act_bwd(finalize)");
        if (pool == 1)
            SIV_BWD_V(1, 1, dim3(vec_grid_apply(OQ, 4), groups), dout, y, mean, var, gamma, beta, residual, keep, p, slope,
                      eps, sums, invM, OQ, g, C, nullptr, dy, dresidual);
        else
            SIV_BWD_V(2, 1, dim3(vec_grid_apply(OQ, 1), groups), dout, y, mean, var, gamma, beta, residual, keep, p, slope,
                      eps, sums, invM, OQ, g, C, nullptr, dy, dresidual);
#undef SIV_BWD_V
        SIV_CHECK_LAUNCH("bn_act_bwd(apply)");
        return SIV_OK;
    }
    const long in_total = M * C, out_total = (long)Ng * (D / pool) * (H / pool) * (W / pool) * C;
    for (int gi = 0; gi < groups; ++gi) {                         // scalar kernels: one launch sequence per group
        const float* dg = dout + gi * out_total;
        const float* yg = y + gi * in_total;
        const float* rg = residual ? residual + gi * in_total : nullptr;
        const uint8_t* kg = keep ? keep + gi * out_total : nullptr;
        float* drg = dresidual ? dresidual + gi * in_total : nullptr;
        const float *mg = mean + gi * C, *vg = var + gi * C;
        if (pool == 1)
            bn_bwd_partial<1><<<nblk, 256, 512 * sizeof(float), s>>>(dg, yg, mg, vg, gamma, beta, rg, kg, p, slope, eps,
                                                                     M, D, H, W, C, part);
        else
            bn_bwd_partial<2><<<nblk, 256, 512 * sizeof(float), s>>>(dg, yg, mg, vg, gamma, beta, rg, kg, p, slope, eps,
                                                                     M, D, H, W, C, part);
        SIV_CHECK_LAUNCH("bn_act_bwd(partial)");
        bn_bwd_finalize<<<(C + 3) / 4, 256, 0, s>>>(part, fold_rows(part, nblk, 2 * C, s), C, sums, dgamma, dbeta,
                                                    accumulate || gi > 0, 1, 0);
        SIV_CHECK_LAUNCH("bn_act_bwd(finalize)");
        if (pool == 1)
            bn_bwd_apply<1><<<stream_grid(M * C), 256, 0, s>>>(dg, yg, mg, vg, gamma, beta, rg, kg, p, slope, eps, sums,
                                                               M, D, H, W, C, dy + gi * in_total, drg);
        else
            bn_bwd_apply<2><<<stream_grid(M * C), 256, 0, s>>>(dg, yg, mg, vg, gamma, beta, rg, kg, p, slope, eps, sums,
                                                               M, D, H, W, C, dy + gi * in_total, drg);
        SIV_CHECK_LAUNCH("bn_act_bwd(apply)");
    }
    return SIV_OK;
}
extern "C" int siv_bn_act_bwd_grouped(const float* dout, const float* y, const float* mean, const float* var,
                                      const float* gamma, const float* beta, const float* residual,
                                      const uint8_t* keep, float p, float slope, int pool, float eps,
                                      float* dy, float* dgamma, float* dbeta, float* dresidual,
                                      int accumulate, int N, int D, int H, int W, int C, int groups, void* ws,
                                      size_t ws_bytes, siv_stream_t stream) {
    return bn_act_bwd_impl(dout, y, false, mean, var, gamma, beta, residual, keep, p, slope, pool, eps, dy, dgamma,
                           dbeta, dresidual, accumulate, N, D, H, W, C, groups, ws, ws_bytes, stream);
}
extern "C" int siv_bn_act_bwd_bf16(const float* dout, const void* y_bf16, const float* mean, const float* var,
                                   const float* gamma, const float* beta, const float* residual,
                                   const uint8_t* keep, float p, float slope, int pool, float eps,
                                   float* dy, float* dgamma, float* dbeta, float* dresidual,
                                   int accumulate, int N, int D, int H, int W, int C, int groups, void* ws,
                                   size_t ws_bytes, siv_stream_t stream) {
    return bn_act_bwd_impl(dout, (const float*)y_bf16, true, mean, var, gamma, beta, residual, keep, p, slope, pool,
                           eps, dy, dgamma, dbeta, dresidual, accumulate, N, D, H, W, C, groups, ws, ws_bytes, stream);
}
extern "C" int siv_bn_act_bwd(const float* dout, const float* y, const float* mean, const float* var,
                              const float* gamma, const float* beta, const float* residual,
                              const uint8_t* keep, float p, float slope, int pool, float eps,
                              float* dy, float* dgamma, float* dbeta, float* dresidual,
                              int accumulate, int N, int D, int H, int W, int C, void* ws,
                              size_t ws_bytes, siv_stream_t stream) {
    return siv_bn_act_bwd_grouped(dout, y, mean, var, gamma, beta, residual, keep, p, slope, pool, eps, dy, dgamma,
                                  dbeta, dresidual, accumulate, N, D, H, W, C, 1, ws, ws_bytes, stream);
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
