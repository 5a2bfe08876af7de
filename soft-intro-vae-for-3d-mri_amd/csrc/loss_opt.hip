// Latent sampling, per-sample loss reductions, Adam, Philox RNG.  fp32, HBM/latency bound.
#include "siv_common.h"
#include <atomic>

namespace siv {

static inline int ew_grid(long n) {
    long g = cdiv(n, 256);
    if (g > 256L * 16) g = 256L * 16;
    if (g < 1) g = 1;
    return (int)g;
}

__global__ __launch_bounds__(256) void reparam_fwd_k(const float* __restrict__ mu,
                                                     const float* __restrict__ logvar,
                                                     const float* __restrict__ eps, float eps_const,
                                                     float* __restrict__ z, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long)gridDim.x * blockDim.x) {
        const float e = eps ? eps[i] : eps_const;
        z[i] = fmaf(e, expf(0.5f * logvar[i]), mu[i]);
    }
}
__global__ __launch_bounds__(256) void reparam_bwd_k(const float* __restrict__ dz,
                                                     const float* __restrict__ logvar,
                                                     const float* __restrict__ eps, float eps_const,
                                                     float* __restrict__ dmu,
                                                     float* __restrict__ dlogvar, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long)gridDim.x * blockDim.x) {
        const float e = eps ? eps[i] : eps_const;
        const float g = dz[i];
        if (dmu) dmu[i] = g;
        if (dlogvar) dlogvar[i] = g * e * 0.5f * expf(0.5f * logvar[i]);
    }
}

// one block per sample
__global__ __launch_bounds__(256) void kl_fwd_k(const float* __restrict__ mu,
                                                const float* __restrict__ logvar,
                                                float* __restrict__ out, long n) {
    __shared__ float sh[4];
    const long base = (long)blockIdx.x * n;
    float acc = 0.f;
    for (long i = threadIdx.x; i < n; i += blockDim.x) {
        const float m = mu[base + i], lv = logvar[base + i];
        acc += 1.f + lv - m * m - expf(lv);
    }
    acc = block_sum_256(acc, sh);
    if (threadIdx.x == 0) out[blockIdx.x] = -0.5f * acc;
}
__global__ __launch_bounds__(256) void kl_bwd_k(const float* __restrict__ g,
                                                const float* __restrict__ mu,
                                                const float* __restrict__ logvar,
                                                float* __restrict__ dmu, float* __restrict__ dlogvar,
                                                int B, long n) {
    const long total = (long)B * n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const float gb = g[i / n];
        dmu[i] = gb * mu[i];
        dlogvar[i] = gb * 0.5f * (expf(logvar[i]) - 1.f);
    }
}

// grid (chunks, B): partial[b][chunk]
__global__ __launch_bounds__(256) void sse_partial_k(const float* __restrict__ x,
                                                     const float* __restrict__ y,
                                                     float* __restrict__ part, long n) {
    __shared__ float sh[4];
    const long base = (long)blockIdx.y * n;
    float acc = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long)gridDim.x * blockDim.x) {
        const float d = x[base + i] - y[base + i];
        acc = fmaf(d, d, acc);
    }
    acc = block_sum_256(acc, sh);
    if (threadIdx.x == 0) part[(long)blockIdx.y * gridDim.x + blockIdx.x] = acc;
}
__global__ void sse_finalize_k(const float* __restrict__ part, int chunks, float* __restrict__ out,
                               int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double s = 0.0;
    for (int i = 0; i < chunks; ++i) s += (double)part[(long)b * chunks + i];
    out[b] = (float)s;
}
__global__ __launch_bounds__(256) void sse_bwd_k(const float* __restrict__ g,
                                                 const float* __restrict__ x,
                                                 const float* __restrict__ y, float* __restrict__ dx,
                                                 float* __restrict__ dy, int B, long n) {
    const long total = (long)B * n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const float d = 2.f * g[i / n] * (x[i] - y[i]);
        if (dx) dx[i] = d;
        if (dy) dy[i] = -d;
    }
}

// ---- the two scalar objectives of one iteration (one wave each; B is the batch, a handful of samples) ----
// E-step, utils/my_trainer.py:260-284:
//   lossE = 10 * ( s*(br*mean(sse_real) + bk*mean(kl_real))
//                  + 0.5*(mean_b exp(-2s(br*sse_fake + bn*kl_fake)) + mean_b exp(-2s(br*sse_rec + bn*kl_rec))) )
// out = {lossE, mean(kl_real), exp_elbo_rec, exp_elbo_fake}
__global__ __launch_bounds__(64) void sintro_loss_e_fwd_k(const float* __restrict__ sse_real, const float* __restrict__ kl_real,
                                                          const float* __restrict__ sse_fake, const float* __restrict__ kl_fake,
                                                          const float* __restrict__ sse_rec, const float* __restrict__ kl_rec,
                                                          int B, float s, float br, float bn, float bk,
                                                          float* __restrict__ out) {
    double a = 0.0, k = 0.0, ef = 0.0, er = 0.0;
    for (int b = threadIdx.x; b < B; b += 64) {
        a += (double)sse_real[b];
        k += (double)kl_real[b];
        ef += (double)expf(-2.f * s * (br * sse_fake[b] + bn * kl_fake[b]));
        er += (double)expf(-2.f * s * (br * sse_rec[b] + bn * kl_rec[b]));
    }
    a = wave_sum_d(a), k = wave_sum_d(k), ef = wave_sum_d(ef), er = wave_sum_d(er);
    if (threadIdx.x == 0) {
        const float m_rec = (float)(a / B), m_kl = (float)(k / B), m_ef = (float)(ef / B), m_er = (float)(er / B);
        out[0] = 10.f * (s * (br * m_rec + bk * m_kl) + 0.5f * (m_ef + m_er));
        out[1] = m_kl;
        out[2] = m_er;
        out[3] = m_ef;
    }
}
// gradients of lossE (times the incoming g) with respect to the six per-sample vectors
__global__ __launch_bounds__(64) void sintro_loss_e_bwd_k(const float* __restrict__ g, const float* __restrict__ sse_fake,
                                                          const float* __restrict__ kl_fake, const float* __restrict__ sse_rec,
                                                          const float* __restrict__ kl_rec, int B, float s, float br,
                                                          float bn, float bk, float* __restrict__ d /* [6][B] */) {
    const float g0 = 10.f * g[0];
    for (int b = threadIdx.x; b < B; b += 64) {
        const float ef = expf(-2.f * s * (br * sse_fake[b] + bn * kl_fake[b]));
        const float er = expf(-2.f * s * (br * sse_rec[b] + bn * kl_rec[b]));
        d[0 * B + b] = g0 * s * br / B;
        d[1 * B + b] = g0 * s * bk / B;
        d[2 * B + b] = g0 * 0.5f / B * ef * (-2.f * s * br);
        d[3 * B + b] = g0 * 0.5f / B * ef * (-2.f * s * bn);
        d[4 * B + b] = g0 * 0.5f / B * er * (-2.f * s * br);
        d[5 * B + b] = g0 * 0.5f / B * er * (-2.f * s * bn);
    }
}
// D-step, utils/my_trainer.py:301-321:
//   lossD = 10 * s * ( br*mean(sse_real) + 0.5*bk*(mean(kl_rec) + mean(kl_fake))
//                      + gr*0.5*br*(mean(sse_rec_rec) + mean(sse_fake_rec)) )
// out = {lossD, mean(kl_fake), mean(kl_rec), mean(sse_real)}
__global__ __launch_bounds__(64) void sintro_loss_d_fwd_k(const float* __restrict__ sse_real, const float* __restrict__ kl_rec,
                                                          const float* __restrict__ kl_fake, const float* __restrict__ sse_rr,
                                                          const float* __restrict__ sse_fr, int B, float s, float br,
                                                          float bk, float gr, float* __restrict__ out) {
    double a = 0.0, kr = 0.0, kf = 0.0, rr = 0.0, fr = 0.0;
    for (int b = threadIdx.x; b < B; b += 64) {
        a += (double)sse_real[b];
        kr += (double)kl_rec[b];
        kf += (double)kl_fake[b];
        rr += (double)sse_rr[b];
        fr += (double)sse_fr[b];
    }
    a = wave_sum_d(a), kr = wave_sum_d(kr), kf = wave_sum_d(kf), rr = wave_sum_d(rr), fr = wave_sum_d(fr);
    if (threadIdx.x == 0) {
        const float m_a = (float)(a / B), m_kr = (float)(kr / B), m_kf = (float)(kf / B);
        const float m_rr = (float)(rr / B), m_fr = (float)(fr / B);
        out[0] = 10.f * (s * (br * m_a + 0.5f * bk * (m_kr + m_kf) + gr * 0.5f * br * (m_rr + m_fr)));
        out[1] = m_kf;
        out[2] = m_kr;
        out[3] = m_a;
    }
}
__global__ __launch_bounds__(64) void sintro_loss_d_bwd_k(const float* __restrict__ g, int B, float s, float br, float bk,
                                                          float gr, float* __restrict__ d /* [5][B] */) {
    const float g0 = 10.f * g[0] * s / B;
    for (int b = threadIdx.x; b < B; b += 64) {
        d[0 * B + b] = g0 * br;
        d[1 * B + b] = g0 * 0.5f * bk;
        d[2 * B + b] = g0 * 0.5f * bk;
        d[3 * B + b] = g0 * gr * 0.5f * br;
        d[4 * B + b] = g0 * gr * 0.5f * br;
    }
}

__global__ __launch_bounds__(256) void adam_k(float* __restrict__ p, const float* __restrict__ g,
                                              float* __restrict__ m, float* __restrict__ v, long n,
                                              float b1, float b2, float eps, float step_size,
                                              float inv_sqrt_bc2) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = fmaf(b1, m[i], (1.f - b1) * gi);
        const float vi = fmaf(b2, v[i], (1.f - b2) * gi * gi);
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
}

// Adam whose learning rate and step count live in device memory: a captured launch (HIP graph) then
// follows the schedule and the bias corrections from replay to replay.
__global__ __launch_bounds__(256) void adam_dev_k(float* __restrict__ p, const float* __restrict__ g,
                                                  float* __restrict__ m, float* __restrict__ v, long n,
                                                  float b1, float b2, float eps, const float* __restrict__ lr,
                                                  const long long* __restrict__ step) {
    const double t = (double)*step;
    const float step_size = (float)((double)*lr / (1.0 - pow((double)b1, t)));
    const float inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)b2, t)));
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = fmaf(b1, m[i], (1.f - b1) * gi);
        const float vi = fmaf(b2, v[i], (1.f - b2) * gi * gi);
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
}
__global__ void counter_add_k(long long* c, long long inc) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *c += inc;
}

__global__ __launch_bounds__(256) void scale_k(float a, float* x, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        x[i] *= a;
}
// x and y must not overlap (siv_scale is the in-place form)
__global__ __launch_bounds__(256) void axpby_k(float a, const float* __restrict__ x, float b,
                                               float* __restrict__ y, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long)gridDim.x * blockDim.x)
        y[i] = (b == 0.f) ? a * x[i] : fmaf(a, x[i], b * y[i]);
}

// ---- bf16 storage of tensors kept for the backward pass ---------------------------------
// fp32 -> bf16 with round-to-nearest-even (NaN stays NaN), eight elements per thread and step
// (32 bytes in, 16 bytes out); bf16 -> fp32 is exact.
__global__ __launch_bounds__(256) void f32_to_bf16_k(const float* __restrict__ x, uint16_t* __restrict__ h, long n) {
    const long n8 = n >> 3;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const float4 a = reinterpret_cast<const float4*>(x)[2 * i], b = reinterpret_cast<const float4*>(x)[2 * i + 1];
        uint4 o;
        o.x = bf16_rne(a.x) | (bf16_rne(a.y) << 16); o.y = bf16_rne(a.z) | (bf16_rne(a.w) << 16);
        o.z = bf16_rne(b.x) | (bf16_rne(b.y) << 16); o.w = bf16_rne(b.z) | (bf16_rne(b.w) << 16);
        reinterpret_cast<uint4*>(h)[i] = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) h[(n8 << 3) + threadIdx.x] = (uint16_t)bf16_rne(x[(n8 << 3) + threadIdx.x]);
}
__global__ __launch_bounds__(256) void bf16_to_f32_k(const uint16_t* __restrict__ h, float* __restrict__ x, long n) {
    const long n8 = n >> 3;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const uint4 v = reinterpret_cast<const uint4*>(h)[i];
        reinterpret_cast<float4*>(x)[2 * i] = make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u),
                                                         __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
        reinterpret_cast<float4*>(x)[2 * i + 1] = make_float4(__uint_as_float(v.z << 16), __uint_as_float(v.z & 0xffff0000u),
                                                             __uint_as_float(v.w << 16), __uint_as_float(v.w & 0xffff0000u));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) x[(n8 << 3) + threadIdx.x] = __uint_as_float((uint32_t)h[(n8 << 3) + threadIdx.x] << 16);
}

// ---- Philox-4x32-10 (siv_common.h) --------------------------------------------------
__device__ __forceinline__ float u01(uint32_t r) {  // (0,1]
    return ((float)(r >> 8) + 1.0f) * (1.0f / 16777216.0f);
}

// epoch (optional, device memory): added to the stream id in steps of 2^20, so that a launch captured in a
// HIP graph draws fresh numbers on every replay once the counter has been advanced
__global__ __launch_bounds__(256) void randn_k(float* __restrict__ out, long n, uint64_t seed,
                                               uint64_t sid, const unsigned long long* __restrict__ epoch) {
    if (epoch) sid += *epoch * SIV_EPOCH_STRIDE;
    const long quads = (n + 3) / 4;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < quads;
         q += (long)gridDim.x * blockDim.x) {
        const u4 r = philox4x32_10((uint64_t)q, sid, seed);
        const float r0 = sqrtf(-2.f * logf(u01(r.x))), t0 = 6.28318530717958647692f * u01(r.y);
        const float r1 = sqrtf(-2.f * logf(u01(r.z))), t1 = 6.28318530717958647692f * u01(r.w);
        const float v[4] = {r0 * cosf(t0), r0 * sinf(t0), r1 * cosf(t1), r1 * sinf(t1)};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (q * 4 + j < n) out[q * 4 + j] = v[j];
    }
}
__global__ __launch_bounds__(256) void dropout_keep_k(uint8_t* __restrict__ keep, long n, float p,
                                                      uint64_t seed, uint64_t sid,
                                                      const unsigned long long* __restrict__ epoch) {
    if (epoch) sid += *epoch * SIV_EPOCH_STRIDE;
    const long quads = (n + 3) / 4;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < quads;
         q += (long)gridDim.x * blockDim.x) {
        const u4 r = philox4x32_10((uint64_t)q, sid, seed);
        const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (q * 4 + j < n)
                keep[q * 4 + j] = philox_keeps(rr[j], p) ? 1 : 0;
    }
}

}  // namespace siv

using namespace siv;

extern "C" int siv_reparam_fwd(const float* mu, const float* logvar, const float* eps,
                               float eps_const, float* z, long n, siv_stream_t stream) {
    SIV_CHECK_ARG(mu && logvar && z && n > 0, "reparam_fwd: bad argument");
    reparam_fwd_k<<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(mu, logvar, eps, eps_const, z, n);
    SIV_CHECK_LAUNCH("reparam_fwd");
    return SIV_OK;
}
extern "C" int siv_reparam_bwd(const float* dz, const float* logvar, const float* eps,
                               float eps_const, float* dmu, float* dlogvar, long n,
                               siv_stream_t stream) {
    SIV_CHECK_ARG(dz && logvar && n > 0, "reparam_bwd: bad argument");
    reparam_bwd_k<<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(dz, logvar, eps, eps_const, dmu,
                                                               dlogvar, n);
    SIV_CHECK_LAUNCH("reparam_bwd");
    return SIV_OK;
}
extern "C" int siv_kl_fwd(const float* mu, const float* logvar, float* out, int B, long n,
                          siv_stream_t stream) {
    SIV_CHECK_ARG(mu && logvar && out && B > 0 && n > 0, "kl_fwd: bad argument");
    kl_fwd_k<<<B, 256, 0, (hipStream_t)stream>>>(mu, logvar, out, n);
    SIV_CHECK_LAUNCH("kl_fwd");
    return SIV_OK;
}
extern "C" int siv_kl_bwd(const float* g, const float* mu, const float* logvar, float* dmu,
                          float* dlogvar, int B, long n, siv_stream_t stream) {
    SIV_CHECK_ARG(g && mu && logvar && dmu && dlogvar && B > 0 && n > 0, "kl_bwd: bad argument");
    kl_bwd_k<<<ew_grid((long)B * n), 256, 0, (hipStream_t)stream>>>(g, mu, logvar, dmu, dlogvar, B, n);
    SIV_CHECK_LAUNCH("kl_bwd");
    return SIV_OK;
}
extern "C" int siv_sse_fwd(const float* x, const float* y, float* out, int B, long n, void* ws,
                           size_t ws_bytes, siv_stream_t stream) {
    SIV_CHECK_ARG(x && y && out && B > 0 && n > 0, "sse_fwd: bad argument");
    long chunks = cdiv(n, 256 * 16);
    if (chunks > 512) chunks = 512;
    SIV_CHECK_WS((size_t)B * chunks * sizeof(float), ws_bytes, "sse_fwd");
    hipStream_t s = (hipStream_t)stream;
    sse_partial_k<<<dim3((unsigned)chunks, (unsigned)B), 256, 0, s>>>(x, y, (float*)ws, n);
    SIV_CHECK_LAUNCH("sse_fwd(partial)");
    sse_finalize_k<<<(B + 63) / 64, 64, 0, s>>>((const float*)ws, (int)chunks, out, B);
    SIV_CHECK_LAUNCH("sse_fwd(finalize)");
    return SIV_OK;
}
// ---- allocation-free variant of siv_sse_fwd --------------------------------------------------------
// The per-block partial sums live in a file-scope device array instead of a caller workspace, so the call
// needs no device memory beyond its inputs and output.  Each GPU the library runs on holds its own copy
// of the array (SIV_SSE_NOALLOC_MAX_B * 512 floats = 512 KiB) at its own address until the process ends;
// the address is looked up once per device, with that device current, and the kernels receive it as an
// ordinary pointer argument.  Every partial a call reads was written by that call, so nothing has to be
// cleared between calls; two calls that overlap in time on one GPU (different streams) race on the array.
#define SIV_SSE_NOALLOC_MAX_B 256
#define SIV_SSE_NOALLOC_MAX_DEVICES 64
static __device__ __align__(256) float sse_noalloc_scratch[SIV_SSE_NOALLOC_MAX_B * 512];
static std::atomic<float*> g_sse_noalloc_ptr[SIV_SSE_NOALLOC_MAX_DEVICES];

extern "C" int siv_sse_fwd_noalloc(const float* x, const float* y, float* out, int B, long n,
                                   siv_stream_t stream) {
    SIV_CHECK_ARG(x && y && out && B > 0 && n > 0, "sse_fwd_noalloc: bad argument");
    SIV_CHECK_ARG(B <= SIV_SSE_NOALLOC_MAX_B,
                  "sse_fwd_noalloc: batch %d exceeds the built-in scratch (at most %d samples per call); use "
                  "siv_sse_fwd with a workspace", B, SIV_SSE_NOALLOC_MAX_B);
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SIV_SSE_NOALLOC_MAX_DEVICES) {
        set_error("sse_fwd_noalloc: no current HIP device, or device index %d beyond %d", dev,
                  SIV_SSE_NOALLOC_MAX_DEVICES - 1);
        return SIV_EHIP;
    }
    float* part = g_sse_noalloc_ptr[dev].load(std::memory_order_acquire);
    if (!part) {
        // first call on this device: the array's name is only a host-side handle, its device address
        // comes from the runtime (a different one on every GPU)
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(sse_noalloc_scratch));
        if (e != hipSuccess || !p) {
            set_error("sse_fwd_noalloc: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
            return SIV_EHIP;
        }
        part = (float*)p;
        g_sse_noalloc_ptr[dev].store(part, std::memory_order_release);
    }
    long chunks = cdiv(n, 256 * 16);
    if (chunks > 512) chunks = 512;
    hipStream_t s = (hipStream_t)stream;
    sse_partial_k<<<dim3((unsigned)chunks, (unsigned)B), 256, 0, s>>>(x, y, part, n);
    SIV_CHECK_LAUNCH("sse_fwd_noalloc(partial)");
    sse_finalize_k<<<(B + 63) / 64, 64, 0, s>>>(part, (int)chunks, out, B);
    SIV_CHECK_LAUNCH("sse_fwd_noalloc(finalize)");
    return SIV_OK;
}
extern "C" int siv_sse_bwd(const float* g, const float* x, const float* y, float* dx, float* dy,
                           int B, long n, siv_stream_t stream) {
    SIV_CHECK_ARG(g && x && y && (dx || dy) && B > 0 && n > 0, "sse_bwd: bad argument");
    sse_bwd_k<<<ew_grid((long)B * n), 256, 0, (hipStream_t)stream>>>(g, x, y, dx, dy, B, n);
    SIV_CHECK_LAUNCH("sse_bwd");
    return SIV_OK;
}
extern "C" int siv_sintro_loss_e_fwd(const float* sse_real, const float* kl_real, const float* sse_fake,
                                     const float* kl_fake, const float* sse_rec, const float* kl_rec, int B,
                                     float scale, float beta_rec, float beta_neg, float beta_kl, float* out4,
                                     siv_stream_t stream) {
    SIV_CHECK_ARG(sse_real && kl_real && sse_fake && kl_fake && sse_rec && kl_rec && out4 && B > 0,
                  "sintro_loss_e_fwd: bad argument");
    sintro_loss_e_fwd_k<<<1, 64, 0, (hipStream_t)stream>>>(sse_real, kl_real, sse_fake, kl_fake, sse_rec, kl_rec, B,
                                                           scale, beta_rec, beta_neg, beta_kl, out4);
    SIV_CHECK_LAUNCH("sintro_loss_e_fwd");
    return SIV_OK;
}
extern "C" int siv_sintro_loss_e_bwd(const float* g, const float* sse_fake, const float* kl_fake,
                                     const float* sse_rec, const float* kl_rec, int B, float scale, float beta_rec,
                                     float beta_neg, float beta_kl, float* d6, siv_stream_t stream) {
    SIV_CHECK_ARG(g && sse_fake && kl_fake && sse_rec && kl_rec && d6 && B > 0, "sintro_loss_e_bwd: bad argument");
    sintro_loss_e_bwd_k<<<1, 64, 0, (hipStream_t)stream>>>(g, sse_fake, kl_fake, sse_rec, kl_rec, B, scale, beta_rec,
                                                           beta_neg, beta_kl, d6);
    SIV_CHECK_LAUNCH("sintro_loss_e_bwd");
    return SIV_OK;
}
extern "C" int siv_sintro_loss_d_fwd(const float* sse_real, const float* kl_rec, const float* kl_fake,
                                     const float* sse_rec_rec, const float* sse_fake_rec, int B, float scale,
                                     float beta_rec, float beta_kl, float gamma_r, float* out4, siv_stream_t stream) {
    SIV_CHECK_ARG(sse_real && kl_rec && kl_fake && sse_rec_rec && sse_fake_rec && out4 && B > 0,
                  "sintro_loss_d_fwd: bad argument");
    sintro_loss_d_fwd_k<<<1, 64, 0, (hipStream_t)stream>>>(sse_real, kl_rec, kl_fake, sse_rec_rec, sse_fake_rec, B,
                                                           scale, beta_rec, beta_kl, gamma_r, out4);
    SIV_CHECK_LAUNCH("sintro_loss_d_fwd");
    return SIV_OK;
}
extern "C" int siv_sintro_loss_d_bwd(const float* g, int B, float scale, float beta_rec, float beta_kl, float gamma_r,
                                     float* d5, siv_stream_t stream) {
    SIV_CHECK_ARG(g && d5 && B > 0, "sintro_loss_d_bwd: bad argument");
    sintro_loss_d_bwd_k<<<1, 64, 0, (hipStream_t)stream>>>(g, B, scale, beta_rec, beta_kl, gamma_r, d5);
    SIV_CHECK_LAUNCH("sintro_loss_d_bwd");
    return SIV_OK;
}
extern "C" int siv_adam_step(float* p, const float* g, float* m, float* v, long n, float lr,
                             float beta1, float beta2, float eps, int step, siv_stream_t stream) {
    SIV_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "adam_step: bad argument");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    adam_k<<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, beta1, beta2, eps,
                                                        (float)((double)lr / bc1),
                                                        (float)(1.0 / sqrt(bc2)));
    SIV_CHECK_LAUNCH("adam_step");
    return SIV_OK;
}
extern "C" int siv_randn_at(float* out, long n, uint64_t seed, uint64_t stream_id, const uint64_t* epoch,
                            siv_stream_t stream) {
    SIV_CHECK_ARG(out && n > 0, "randn: bad argument");
    randn_k<<<ew_grid((n + 3) / 4), 256, 0, (hipStream_t)stream>>>(out, n, seed, stream_id,
                                                                  (const unsigned long long*)epoch);
    SIV_CHECK_LAUNCH("randn");
    return SIV_OK;
}
extern "C" int siv_randn(float* out, long n, uint64_t seed, uint64_t stream_id, siv_stream_t stream) {
    return siv_randn_at(out, n, seed, stream_id, nullptr, stream);
}
extern "C" int siv_dropout_keep_at(uint8_t* keep, long n, float p, uint64_t seed, uint64_t stream_id,
                                   const uint64_t* epoch, siv_stream_t stream) {
    SIV_CHECK_ARG(keep && n > 0 && p >= 0.f && p < 1.f, "dropout_keep: bad argument");
    dropout_keep_k<<<ew_grid((n + 3) / 4), 256, 0, (hipStream_t)stream>>>(keep, n, p, seed, stream_id,
                                                                         (const unsigned long long*)epoch);
    SIV_CHECK_LAUNCH("dropout_keep");
    return SIV_OK;
}
extern "C" int siv_dropout_keep(uint8_t* keep, long n, float p, uint64_t seed, uint64_t stream_id,
                                siv_stream_t stream) {
    return siv_dropout_keep_at(keep, n, p, seed, stream_id, nullptr, stream);
}
extern "C" int siv_adam_step_at(float* p, const float* g, float* m, float* v, long n, const float* lr,
                                float beta1, float beta2, float eps, const int64_t* step, siv_stream_t stream) {
    SIV_CHECK_ARG(p && g && m && v && lr && step && n > 0, "adam_step_at: bad argument");
    adam_dev_k<<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, beta1, beta2, eps, lr,
                                                            (const long long*)step);
    SIV_CHECK_LAUNCH("adam_step_at");
    return SIV_OK;
}
extern "C" int siv_counter_add(int64_t* counter, int64_t inc, siv_stream_t stream) {
    SIV_CHECK_ARG(counter != nullptr, "counter_add: null pointer");
    counter_add_k<<<1, 64, 0, (hipStream_t)stream>>>((long long*)counter, (long long)inc);
    SIV_CHECK_LAUNCH("counter_add");
    return SIV_OK;
}
extern "C" int siv_scale(float a, float* x, long n, siv_stream_t stream) {
    SIV_CHECK_ARG(x && n > 0, "scale: bad argument");
    scale_k<<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(a, x, n);
    SIV_CHECK_LAUNCH("scale");
    return SIV_OK;
}
extern "C" int siv_f32_to_bf16(const float* x, void* h, long n, siv_stream_t stream) {
    SIV_CHECK_ARG(x && h && n > 0, "f32_to_bf16: bad argument");
    SIV_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)h & 15) == 0, "f32_to_bf16: buffers must be 16-byte aligned");
    f32_to_bf16_k<<<ew_grid(cdiv(n, 8)), 256, 0, (hipStream_t)stream>>>(x, (uint16_t*)h, n);
    SIV_CHECK_LAUNCH("f32_to_bf16");
    return SIV_OK;
}
extern "C" int siv_bf16_to_f32(const void* h, float* x, long n, siv_stream_t stream) {
    SIV_CHECK_ARG(x && h && n > 0, "bf16_to_f32: bad argument");
    SIV_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)h & 15) == 0, "bf16_to_f32: buffers must be 16-byte aligned");
    bf16_to_f32_k<<<ew_grid(cdiv(n, 8)), 256, 0, (hipStream_t)stream>>>((const uint16_t*)h, x, n);
    SIV_CHECK_LAUNCH("bf16_to_f32");
    return SIV_OK;
}
extern "C" int siv_axpby(float a, const float* x, float b, float* y, long n, siv_stream_t stream) {
    SIV_CHECK_ARG(x && y && n > 0, "axpby: bad argument");
    SIV_CHECK_ARG(x + n <= y || y + n <= x, "axpby: x and y must not overlap (use siv_scale to scale in place)");
    axpby_k<<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(a, x, b, y, n);
    SIV_CHECK_LAUNCH("axpby");
    return SIV_OK;
}
