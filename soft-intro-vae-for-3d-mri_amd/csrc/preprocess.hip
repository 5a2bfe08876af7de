// Input pipeline on the device: the per-volume intensity normalisation the reference applies in
// its Dataset (clip to [0, 4 sigma], then min-max to [0, 1]) and the flip / shift augmentation.
// One reduction pass and one apply pass over each raw volume; volumes of a batch are independent.
#include "siv_common.h"

namespace siv {

constexpr int PP_BLOCKS = 64;   // reduction blocks per volume (one wave combines them)

// part[(b * PP_BLOCKS + blk) * 4 + {0,1,2,3}] = sum, sum of squares, min, max of the block's slice
__global__ __launch_bounds__(256) void pp_reduce_k(const float* __restrict__ v, long n,
                                                   double* __restrict__ part) {
    __shared__ double sh[4][4];
    const float* p = v + (long)blockIdx.y * n;
    double s = 0.0, ss = 0.0;
    float mn = INFINITY, mx = -INFINITY;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float x = p[i];
        s += (double)x;
        ss += (double)x * (double)x;
        mn = fminf(mn, x);
        mx = fmaxf(mx, x);
    }
    s = wave_sum_d(s);
    ss = wave_sum_d(ss);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, o, 64));
        mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        sh[w][0] = s; sh[w][1] = ss; sh[w][2] = (double)mn; sh[w][3] = (double)mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* o = part + ((long)blockIdx.y * gridDim.x + blockIdx.x) * 4;
        o[0] = (sh[0][0] + sh[1][0]) + (sh[2][0] + sh[3][0]);
        o[1] = (sh[0][1] + sh[1][1]) + (sh[2][1] + sh[3][1]);
        o[2] = fmin(fmin(sh[0][2], sh[1][2]), fmin(sh[2][2], sh[3][2]));
        o[3] = fmax(fmax(sh[0][3], sh[1][3]), fmax(sh[2][3], sh[3][3]));
    }
}

// out = (clip(v, 0, cut * std) - lo) / (hi - lo), lo / hi = min / max of the clipped volume.
// A constant volume divides 0 by 0 and yields NaN, as numpy does in the reference.
__global__ __launch_bounds__(256) void pp_apply_k(const float* __restrict__ v, long n,
                                                  const double* __restrict__ part, int nblk, float cut,
                                                  float* __restrict__ out) {
    __shared__ float prm[3];
    if (threadIdx.x < 64) {
        const double* pb = part + (long)blockIdx.y * nblk * 4;
        double s = 0.0, ss = 0.0, mn = INFINITY, mx = -INFINITY;
        for (int b = threadIdx.x; b < nblk; b += 64) {
            s += pb[b * 4 + 0];
            ss += pb[b * 4 + 1];
            mn = fmin(mn, pb[b * 4 + 2]);
            mx = fmax(mx, pb[b * 4 + 3]);
        }
        s = wave_sum_d(s);
        ss = wave_sum_d(ss);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn = fmin(mn, __shfl_xor(mn, o, 64));
            mx = fmax(mx, __shfl_xor(mx, o, 64));
        }
        if (threadIdx.x == 0) {
            const double mean = s / (double)n;
            double var = ss / (double)n - mean * mean;
            if (var < 0.0) var = 0.0;
            const float top = cut * (float)sqrt(var);
            prm[0] = top;
            prm[1] = fminf(fmaxf((float)mn, 0.f), top);   // min of the clipped volume
            prm[2] = fminf(fmaxf((float)mx, 0.f), top);   // max of the clipped volume
        }
    }
    __syncthreads();
    const float top = prm[0], lo = prm[1], range = prm[2] - prm[1];
    const float* p = v + (long)blockIdx.y * n;
    float* o = out + (long)blockIdx.y * n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        o[i] = (fminf(fmaxf(p[i], 0.f), top) - lo) / range;
}

// out[b][d][h][w] = in[b][sd(d)][sh(h)][sw(w)] with s(i) = flip ? n-1-(i-shift) : i-shift, zero
// where the source index falls outside the volume.  params: int[B][6] = flipD,flipH,flipW,shiftD,
// shiftH,shiftW per volume.
__global__ __launch_bounds__(256) void flip_shift_k(const float* __restrict__ in,
                                                    const int* __restrict__ params,
                                                    float* __restrict__ out, int D, int H, int W) {
    const int* pr = params + blockIdx.y * 6;
    const long n = (long)D * H * W;
    const float* p = in + (long)blockIdx.y * n;
    float* o = out + (long)blockIdx.y * n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int w = (int)(i % W), h = (int)((i / W) % H), d = (int)(i / ((long)W * H));
        int sd = d - pr[3], shh = h - pr[4], sw = w - pr[5];
        if (pr[0]) sd = D - 1 - sd;
        if (pr[1]) shh = H - 1 - shh;
        if (pr[2]) sw = W - 1 - sw;
        float val = 0.f;
        if (sd >= 0 && sd < D && shh >= 0 && shh < H && sw >= 0 && sw < W)
            val = p[((long)sd * H + shh) * W + sw];
        o[i] = val;
    }
}


// Trilinear resampling under a per-volume affine map of voxel indices:
//   out[b][d][h][w] = sum over the 8 neighbours of p = A_b * (d, h, w, 1) of weight * value,
// where a neighbour outside the volume contributes the pad value (the result is continuous in p).
// mats: float[B][12], the 3x4 matrix A_b row-major, mapping OUTPUT indices to INPUT indices.
// pad_mode 0: pad_value; 1: the minimum of volume b, taken from the partials of pp_reduce_k.
__global__ __launch_bounds__(256) void affine_resample_k(const float* __restrict__ in,
                                                         const float* __restrict__ mats,
                                                         float* __restrict__ out, int D, int H, int W,
                                                         int pad_mode, float pad_value,
                                                         const double* __restrict__ part, int nblk) {
    __shared__ float prm[13];
    if (threadIdx.x < 12) prm[threadIdx.x] = mats[blockIdx.y * 12 + threadIdx.x];
    if (threadIdx.x >= 64 && threadIdx.x < 128) {
        float mn = pad_value;
        if (pad_mode == 1) {
            double m = INFINITY;
            for (int b = threadIdx.x - 64; b < nblk; b += 64) m = fmin(m, part[((long)blockIdx.y * nblk + b) * 4 + 2]);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = fmin(m, __shfl_xor(m, o, 64));
            mn = (float)m;
        }
        if (threadIdx.x == 64) prm[12] = mn;
    }
    __syncthreads();
    const float pad = prm[12];
    const long n = (long)D * H * W;
    const float* p = in + (long)blockIdx.y * n;
    float* o = out + (long)blockIdx.y * n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float w = (float)(i % W), h = (float)((i / W) % H), d = (float)(i / ((long)W * H));
        const float pz = fmaf(prm[0], d, fmaf(prm[1], h, fmaf(prm[2], w, prm[3])));
        const float py = fmaf(prm[4], d, fmaf(prm[5], h, fmaf(prm[6], w, prm[7])));
        const float px = fmaf(prm[8], d, fmaf(prm[9], h, fmaf(prm[10], w, prm[11])));
        float val = pad;
        // further than one voxel outside on any axis: all 8 neighbours are padding
        if (pz > -1.f && pz < (float)D && py > -1.f && py < (float)H && px > -1.f && px < (float)W) {
            const float fz = floorf(pz), fy = floorf(py), fx = floorf(px);
            const int z0 = (int)fz, y0 = (int)fy, x0 = (int)fx;
            const float tz = pz - fz, ty = py - fy, tx = px - fx;
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int zz = z0 + (c >> 2), yy = y0 + ((c >> 1) & 1), xx = x0 + (c & 1);
                const float wgt = ((c >> 2) ? tz : 1.f - tz) * (((c >> 1) & 1) ? ty : 1.f - ty) *
                                  ((c & 1) ? tx : 1.f - tx);
                const bool ok = zz >= 0 && zz < D && yy >= 0 && yy < H && xx >= 0 && xx < W;
                const float v = ok ? p[((long)zz * H + yy) * W + xx] : pad;
                acc = fmaf(wgt, v, acc);
            }
            val = acc;
        }
        o[i] = val;
    }
}

}  // namespace siv

using namespace siv;

extern "C" int siv_preprocess(const float* voxels, float* out, int B, long n, float cut_range,
                              void* ws, size_t ws_bytes, siv_stream_t stream) {
    SIV_CHECK_ARG(voxels && out, "preprocess: null pointer");
    SIV_CHECK_ARG(B > 0 && n > 0, "preprocess: bad shape B=%d n=%ld", B, n);
    SIV_CHECK_ARG(cut_range > 0.f, "preprocess: cut_range must be positive");
    SIV_CHECK_WS((size_t)B * PP_BLOCKS * 4 * sizeof(double), ws_bytes, "preprocess");
    SIV_CHECK_ARG(((uintptr_t)ws & 7) == 0, "preprocess: workspace must be 8-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    pp_reduce_k<<<dim3(PP_BLOCKS, B), 256, 0, s>>>(voxels, n, (double*)ws);
    SIV_CHECK_LAUNCH("preprocess(reduce)");
    long g = cdiv(n, 256 * 4);
    if (g > 1024) g = 1024;
    pp_apply_k<<<dim3((unsigned)g, B), 256, 0, s>>>(voxels, n, (const double*)ws, PP_BLOCKS, cut_range, out);
    SIV_CHECK_LAUNCH("preprocess(apply)");
    return SIV_OK;
}

extern "C" int siv_flip_shift(const float* in, const int* params, float* out, int B, int D, int H,
                              int W, siv_stream_t stream) {
    SIV_CHECK_ARG(in && params && out && in != out, "flip_shift: bad pointer (in-place is not supported)");
    SIV_CHECK_ARG(B > 0 && D > 0 && H > 0 && W > 0, "flip_shift: bad shape");
    long g = cdiv((long)D * H * W, 256 * 4);
    if (g > 1024) g = 1024;
    flip_shift_k<<<dim3((unsigned)g, B), 256, 0, (hipStream_t)stream>>>(in, params, out, D, H, W);
    SIV_CHECK_LAUNCH("flip_shift");
    return SIV_OK;
}

extern "C" int siv_affine_resample(const float* in, const float* mats, float* out, int B, int D, int H,
                                   int W, int pad_mode, float pad_value, void* ws, size_t ws_bytes,
                                   siv_stream_t stream) {
    SIV_CHECK_ARG(in && mats && out && in != out, "affine_resample: bad pointer (in-place is not supported)");
    SIV_CHECK_ARG(B > 0 && D > 0 && H > 0 && W > 0, "affine_resample: bad shape");
    SIV_CHECK_ARG(pad_mode == 0 || pad_mode == 1, "affine_resample: pad_mode must be 0 (constant) or 1 (minimum)");
    hipStream_t s = (hipStream_t)stream;
    const long n = (long)D * H * W;
    if (pad_mode == 1) {
        SIV_CHECK_WS((size_t)B * PP_BLOCKS * 4 * sizeof(double), ws_bytes, "affine_resample");
        SIV_CHECK_ARG(((uintptr_t)ws & 7) == 0, "affine_resample: workspace must be 8-byte aligned");
        pp_reduce_k<<<dim3(PP_BLOCKS, B), 256, 0, s>>>(in, n, (double*)ws);
        SIV_CHECK_LAUNCH("affine_resample(reduce)");
    }
    long g = cdiv(n, 256 * 4);
    if (g > 1024) g = 1024;
    affine_resample_k<<<dim3((unsigned)g, B), 256, 0, s>>>(in, mats, out, D, H, W, pad_mode, pad_value,
                                                           (const double*)ws, PP_BLOCKS);
    SIV_CHECK_LAUNCH("affine_resample");
    return SIV_OK;
}
