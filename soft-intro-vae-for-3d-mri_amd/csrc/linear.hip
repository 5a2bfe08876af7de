// Fully connected layers of the FC-latent model (batch of a few samples against a 19200-wide
// feature vector): skinny GEMMs whose cost is one pass over the weight matrix (92 MB at z = 600).
// Forward and input gradient run on v_mfma_f32_16x16x4_f32 with the batch as the 16 rows of the
// tile; what makes them fast is not the matrix pipe but the decomposition: thousands of waves,
// each streaming a short slab of the matrix with 16-byte loads, partial sums reduced in fixed
// order.  Also the elementwise add + activation of that model's skip connections.
#include "siv_common.h"

namespace siv {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int LB = 8;  // batch rows processed together by the VALU kernels

// ---- VALU fallbacks (K not a multiple of 4) -------------------------------------------------
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
// many slices, few elements: one wave per element, lanes stride over the slices, fixed-order wave sum
__global__ __launch_bounds__(256) void linear_dgrad_reduce_w(const float* __restrict__ part, int nslice,
                                                             long n, float* __restrict__ dx) {
    const int lane = threadIdx.x & 63;
    const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    float s = 0.f;
    for (int t = lane; t < nslice; t += 64) s += part[(long)t * n + i];
    s = wave_sum(s);
    if (lane == 0) dx[i] = s;
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

// ---- matrix-pipe kernels (K % 4 == 0) ---------------------------------------------------------
// Forward.  A wave owns 16 output columns o0.. and a K slab: lane (n, kk) loads w[o0+n][k..k+3] and
// x[b0+n][k..k+3] at k = slab + 16 s + 4 kk as float4 -- per row 64 contiguous bytes per step -- and
// component j of both feeds instruction j (K index 16 s + 4 kk + j on both sides).  One accumulator:
// rows = batch, columns = outputs.  part[slab][b][o].
__global__ __launch_bounds__(256) void linear_fwd_mfma_k(const float* __restrict__ x,
                                                         const float* __restrict__ w,
                                                         float* __restrict__ part, int B, int b0, int K,
                                                         int O, int ncg, int nslab, int slab_len) {
    const int lane = threadIdx.x & 63;
    const int n = lane & 15, kk = lane >> 4;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= ncg * nslab) return;
    const int cg = item % ncg, slab = item / ncg;
    const int o = cg * 16 + n, b = b0 + n;
    const bool ook = o < O, bok = b < B;
    const float* wr = w + (long)(ook ? o : 0) * K;
    const float* xr = x + (long)(bok ? b : 0) * K;
    const int k0 = slab * slab_len;
    const int k1 = k0 + slab_len < K ? k0 + slab_len : K;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kb = k0; kb < k1; kb += 64) {          // four steps per trip: eight loads in flight
        float4 wv[4], xv[4];
        bool kok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            // clamped, unconditional loads (no branch for the compiler to wait at); masked where used
            // a masked lane re-reads the slab's first four elements (k0 is a multiple of 4 below K and K is a
            // multiple of 4, so those always lie inside the row -- nothing is read past the end of a row)
            const int k = kb + 16 * u + 4 * kk;
            kok[u] = k < k1;
            const int kc = kok[u] ? k : k0;
            wv[u] = *reinterpret_cast<const float4*>(wr + kc);
            xv[u] = *reinterpret_cast<const float4*>(xr + kc);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool wok = kok[u] && ook, xok = kok[u] && bok;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xok ? xv[u].x : 0.f, wok ? wv[u].x : 0.f, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xok ? xv[u].y : 0.f, wok ? wv[u].y : 0.f, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xok ? xv[u].z : 0.f, wok ? wv[u].z : 0.f, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xok ? xv[u].w : 0.f, wok ? wv[u].w : 0.f, acc, 0, 0, 0);
        }
    }
    // lane (n, kk), register e: batch row b0 + 4 kk + e, output o
    if (ook) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int bb = b0 + 4 * kk + e;
            if (bb < B) part[((long)slab * B + bb) * O + o] = acc[e];
        }
    }
}
__global__ __launch_bounds__(256) void linear_fwd_reduce(const float* __restrict__ part, int nslab,
                                                         const float* __restrict__ bias, int B, int O,
                                                         float* __restrict__ y) {
    const long n = (long)B * O;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int t = 0; t < nslab; ++t) s += part[(long)t * n + i];
        y[i] = s + (bias ? bias[i % O] : 0.f);
    }
}

// Input gradient.  A wave owns 64 input columns kc0.. and a slab of output rows: lane (n, kk) loads
// w[o][kc0+4n .. +3] for o = slab + 4 s + kk -- 256 contiguous bytes per row and step -- and
// dy[b0+n][o]; component j feeds instruction j, whose column n stands for input column kc0 + 4 n + j.
// part[slab][b][k].
__global__ __launch_bounds__(256) void linear_dgrad_mfma_k(const float* __restrict__ dy,
                                                           const float* __restrict__ w,
                                                           float* __restrict__ part, int B, int b0, int K,
                                                           int O, int ncg, int nslab, int slab_len) {
    const int lane = threadIdx.x & 63;
    const int n = lane & 15, kk = lane >> 4;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= ncg * nslab) return;
    const int cg = item % ncg, slab = item / ncg;
    const int kcol = cg * 64 + 4 * n, b = b0 + n;
    const bool cok = kcol < K, bok = b < B;
    const float* wc = w + (cok ? kcol : 0);
    const float* dr = dy + (long)(bok ? b : 0) * O;
    const int o0 = slab * slab_len;
    const int o1 = o0 + slab_len < O ? o0 + slab_len : O;
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ob = o0; ob < o1; ob += 16) {          // four steps per trip
        float4 wv[4];
        float dv[4];
        bool ook[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int o = ob + 4 * u + kk;
            ook[u] = o < o1;
            const int oc = ook[u] ? o : o0;
            wv[u] = *reinterpret_cast<const float4*>(wc + (long)oc * K);
            dv[u] = dr[oc];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float d = (ook[u] && bok) ? dv[u] : 0.f;
            const bool wok = ook[u] && cok;
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(d, wok ? wv[u].x : 0.f, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(d, wok ? wv[u].y : 0.f, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(d, wok ? wv[u].z : 0.f, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(d, wok ? wv[u].w : 0.f, acc[3], 0, 0, 0);
        }
    }
    if (cok) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int bb = b0 + 4 * kk + e;
            if (bb < B)
                *reinterpret_cast<float4*>(part + ((long)slab * B + bb) * K + kcol) =
                    make_float4(acc[0][e], acc[1][e], acc[2][e], acc[3][e]);
        }
    }
}

// ---- weight gradient: dw[o][k] = sum_b dy[b][o] * x[b][k], a rank-B update -- one streaming write --------
// A thread keeps the float4 x[b][k..k+3] of up to LB batch rows in registers and walks a strip of
// output rows; dy[b][o] is uniform over the block (scalar loads).
__global__ __launch_bounds__(256) void linear_wgrad_v(const float* __restrict__ x,
                                                      const float* __restrict__ dy,
                                                      float* __restrict__ dw, int B, int b0, int K, int O,
                                                      int rows_per_block, int accumulate) {
    const int k = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (k >= K) return;
    float4 xr[LB];
#pragma unroll
    for (int j = 0; j < LB; ++j)
        xr[j] = b0 + j < B ? *reinterpret_cast<const float4*>(x + (long)(b0 + j) * K + k)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
    const int o0 = blockIdx.y * rows_per_block;
    const int o1 = o0 + rows_per_block < O ? o0 + rows_per_block : O;
    for (int o = o0; o < o1; ++o) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < LB; ++j) {
            const float d = b0 + j < B ? dy[(long)(b0 + j) * O + o] : 0.f;
            s.x = fmaf(d, xr[j].x, s.x); s.y = fmaf(d, xr[j].y, s.y);
            s.z = fmaf(d, xr[j].z, s.z); s.w = fmaf(d, xr[j].w, s.w);
        }
        float4* dst = reinterpret_cast<float4*>(dw + (long)o * K + k);
        if (accumulate) {
            const float4 old = *dst;
            s.x += old.x; s.y += old.y; s.z += old.z; s.w += old.w;
        }
        *dst = s;
    }
}

// dw[o][k] = sum_b dy[b][o] * x[b][k] (any K)
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

// slab length (a multiple of `unit`) that splits `len` into about `want` slabs
static inline int slab_len_for(int len, long want, int unit) {
    if (want < 1) want = 1;
    long l = cdiv(cdiv(len, want), unit) * unit;
    if (l < 4L * unit) l = 4L * unit;
    return (int)l;
}
constexpr long LIN_WAVES = 4096;    // 16 waves per CU keep the weight stream saturated

extern "C" int siv_linear_fwd(const float* x, const float* w, const float* bias, float* y, int B,
                              int K, int O, void* ws, size_t ws_bytes, siv_stream_t stream) {
    SIV_CHECK_ARG(x && w && y, "linear_fwd: null pointer");
    SIV_CHECK_ARG(B > 0 && K > 0 && O > 0, "linear_fwd: bad shape B=%d K=%d O=%d", B, K, O);
    hipStream_t s = (hipStream_t)stream;
    if (K % 4 != 0 || conv_path() == 1) {
        linear_fwd_k<<<(O + 3) / 4, 256, 0, s>>>(x, w, bias, y, B, K, O);
        SIV_CHECK_LAUNCH("linear_fwd");
        return SIV_OK;
    }
    const int ncg = (O + 15) / 16;
    const int slab = slab_len_for(K, LIN_WAVES / ncg, 16);
    const int nslab = (K + slab - 1) / slab;
    SIV_CHECK_WS((size_t)nslab * B * O * sizeof(float), ws_bytes, "linear_fwd");
    for (int b0 = 0; b0 < B; b0 += 16) {
        linear_fwd_mfma_k<<<(ncg * nslab + 3) / 4, 256, 0, s>>>(x, w, (float*)ws, B, b0, K, O, ncg, nslab, slab);
        SIV_CHECK_LAUNCH("linear_fwd(matrix)");
    }
    linear_fwd_reduce<<<lin_grid((long)B * O), 256, 0, s>>>((const float*)ws, nslab, bias, B, O, y);
    SIV_CHECK_LAUNCH("linear_fwd(reduce)");
    return SIV_OK;
}

extern "C" int siv_linear_dgrad(const float* dy, const float* w, float* dx, int B, int K, int O,
                                void* ws, size_t ws_bytes, siv_stream_t stream) {
    SIV_CHECK_ARG(dy && w && dx, "linear_dgrad: null pointer");
    SIV_CHECK_ARG(B > 0 && K > 0 && O > 0, "linear_dgrad: bad shape B=%d K=%d O=%d", B, K, O);
    hipStream_t s = (hipStream_t)stream;
    if (K % 4 == 0 && conv_path() != 1) {
        const int ncg = (K + 63) / 64;
        const int slab = slab_len_for(O, LIN_WAVES / ncg, 4);
        const int nslab = (O + slab - 1) / slab;
        SIV_CHECK_WS((size_t)nslab * B * K * sizeof(float), ws_bytes, "linear_dgrad");
        for (int b0 = 0; b0 < B; b0 += 16) {
            linear_dgrad_mfma_k<<<(ncg * nslab + 3) / 4, 256, 0, s>>>(dy, w, (float*)ws, B, b0, K, O, ncg, nslab,
                                                                      slab);
            SIV_CHECK_LAUNCH("linear_dgrad(matrix)");
        }
        const long n = (long)B * K;
        if (nslab >= 64 && n <= 65536)
            linear_dgrad_reduce_w<<<(unsigned)cdiv(n, 4), 256, 0, s>>>((const float*)ws, nslab, n, dx);
        else
            linear_dgrad_reduce<<<lin_grid(n), 256, 0, s>>>((const float*)ws, nslab, n, dx);
        SIV_CHECK_LAUNCH("linear_dgrad(reduce)");
        return SIV_OK;
    }
    const int kblocks = (K + 255) / 256;
    int nslice = 1024 / kblocks;
    if (nslice < 1) nslice = 1;
    if (nslice > O) nslice = O;
    const int rps = (O + nslice - 1) / nslice;
    nslice = (O + rps - 1) / rps;
    SIV_CHECK_WS((size_t)nslice * B * K * sizeof(float), ws_bytes, "linear_dgrad");
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
    if (K % 4 == 0 && conv_path() != 1) {
        const int kb = (K / 4 + 255) / 256;
        int rpb = (int)cdiv((long)O * kb, 8192);      // ~8000 blocks
        if (rpb < 1) rpb = 1;
        for (int b0 = 0; b0 < B; b0 += LB) {
            linear_wgrad_v<<<dim3(kb, (O + rpb - 1) / rpb), 256, 0, s>>>(x, dy, dw, B, b0, K, O, rpb,
                                                                         accumulate || b0 > 0);
            SIV_CHECK_LAUNCH("linear_wgrad");
        }
    } else {
        linear_wgrad_k<<<lin_grid((long)O * K), 256, 0, s>>>(x, dy, dw, B, K, O, accumulate);
        SIV_CHECK_LAUNCH("linear_wgrad");
    }
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
