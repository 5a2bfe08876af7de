// 3x3x3 convolution of a nearest-neighbour x2 upsampled volume without the redundant arithmetic.
//
// After the upsample, output voxel 2i+a (a = 0/1 its parity along an axis) reads the upsampled
// input at 2i+a-1, 2i+a, 2i+a+1, i.e. the LOW-RESOLUTION input at
//     a = 0:  i-1, i, i        a = 1:  i, i, i+1
// so along every axis only two distinct inputs are touched and the three taps collapse to two with
// pre-summed weights (a = 0: {w0, w1+w2};  a = 1: {w0+w1, w2}).  In 3-D each of the 8 output parity
// classes is a 2x2x2 convolution of the low-resolution input: 8 multiply-adds per output and channel
// pair instead of 27.  Arithmetic stays fp32; the only change in rounding is the pre-summation of
// the weights.
//
//   M rows  = low-resolution voxels of a brick (128 or 120), N = 64 output channels
//   K       = 8 collapsed taps x Cin per parity class
//   a block = one brick x one parity along D; the 4 (H, W) parity classes are 4 accumulator sets that
//             share one staged low-resolution halo
#include "siv_common.h"
#include "split_bf16.h"

#ifndef SIV_UPD
#define SIV_UPD 4      // weight-fragment prefetch distance in taps (divides 32)
#endif

namespace siv {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// LBD x LBH x LBW: low-resolution brick
template <int LBD_, int LBH_, int LBW_>
struct UpCfg {
    static constexpr int LBD = LBD_, LBH = LBH_, LBW = LBW_;
    static constexpr int CK = 8, CKP = 12;
    static constexpr int M = LBD * LBH * LBW;             // rows in use (<= 128)
    static constexpr int WM = 2, WN = 2, NW = 4, NTHR = 256, NT = 64, AM = (M + 63) / 64;
    static constexpr int HD = LBD + 1, HH = LBH + 2, HW = LBW + 2;
    static constexpr int HV = HD * HH * HW;               // halo voxels
    static constexpr int NH = (HV * 2 + NTHR - 1) / NTHR; // float4 halo items per thread and chunk
    static constexpr int V_FLOATS = HV * CKP;
    static constexpr int TAPS = 32;                       // (collapsed tap, H parity, W parity)
    static constexpr int SLICE = NT * 8;
    static constexpr int EPI_STRIDE = 40, EPI_FLOATS = 32 * EPI_STRIDE;
    static constexpr int LDS_FLOATS = V_FLOATS > NW * EPI_FLOATS ? V_FLOATS : NW * EPI_FLOATS;
    static constexpr size_t LDS_BYTES = (size_t)LDS_FLOATS * sizeof(float);
    static_assert(M <= 128 && M > 32, "one or two row tiles per wave, two waves along M");
};

// Epilogue shared by the two forward kernels: per row tile and parity class, transpose the 32x32 result
// through this wave's slice of LDS and write 16-byte channel runs.  When stat_part is given the block also
// emits, per output channel, the sum and the sum of squares of its (bias-free) outputs.
template <int LBD, int LBH, int LBW, int AMX>
__device__ __forceinline__ void up_fwd_store(f32x16 (&acc)[4][AMX], float* lds, const float* __restrict__ bias,
                                             float* __restrict__ y, float* __restrict__ stat_part, int n, int z0,
                                             int y0, int x0, int ad, int co0, int D, int H, int W, int Cout) {
    using C = UpCfg<LBD, LBH, LBW>;
    static_assert(AMX == C::AM, "accumulator sets per wave");
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;
    const int Dl = D >> 1, Hl = H >> 1, Wl = W >> 1;
    float* tile = lds + wave * C::EPI_FLOATS;
    const int cq = lane & 7;
    const int co = co0 + wn * 32 + 4 * cq;
    float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias && co < Cout) bq = *reinterpret_cast<const float4*>(bias + co);
    const int orig_b = blockIdx.x;
    float s1 = 0.f, s2 = 0.f;     // this lane's column (channel co0 + wn*32 + li), its 16 rows per tile
#pragma unroll
    for (int a = 0; a < C::AM; ++a) {
#pragma unroll
        for (int pos = 0; pos < 4; ++pos) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[pos][a][r];
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                tile[row * C::EPI_STRIDE + li] = v;
                if (stat_part) {
                    const int m = (wm * C::AM + a) * 32 + row;
                    const int gz = z0 + m / (LBH * LBW), gy = y0 + (m / LBW) % LBH, gx = x0 + m % LBW;
                    if (m < C::M && gz < Dl && gy < Hl && gx < Wl) {
                        s1 += v;
                        s2 = fmaf(v, v, s2);
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int row = ps * 8 + (lane >> 3);
                const int m = (wm * C::AM + a) * 32 + row;
                const int gz = z0 + m / (LBH * LBW), gy = y0 + (m / LBW) % LBH, gx = x0 + m % LBW;
                float4 v = *reinterpret_cast<const float4*>(tile + row * C::EPI_STRIDE + 4 * cq);
                v.x += bq.x; v.y += bq.y; v.z += bq.z; v.w += bq.w;
                if (m < C::M && gz < Dl && gy < Hl && gx < Wl && co < Cout) {
                    const long o = (((long)n * D + 2 * gz + ad) * H + 2 * gy + (pos >> 1)) * W + 2 * gx + (pos & 1);
                    *reinterpret_cast<float4*>(y + o * Cout + co) = v;
                }
            }
        }
    }
    if (stat_part) {
        s1 += __shfl_xor(s1, 32, 64);      // the two lane halves hold the same column
        s2 += __shfl_xor(s2, 32, 64);
        __syncthreads();
        float* red = lds;                  // [2 (wm)][2 (sum, sumsq)][64 co]
        if (lh == 0) {
            red[(wm * 2 + 0) * 64 + wn * 32 + li] = s1;
            red[(wm * 2 + 1) * 64 + wn * 32 + li] = s2;
        }
        __syncthreads();
        if (tid < 128) {
            const int k = tid >> 6, c = tid & 63;
            if (co0 + c < Cout)
                stat_part[((long)orig_b * 2 + k) * Cout + co0 + c] = red[k * 64 + c] + red[(2 + k) * 64 + c];
        }
    }
}

// x: [N][D/2][H/2][W/2][Cin]; y: [N][D][H][W][Cout]; D, H, W even.  wp: see k2_up_pack_weights.
template <int LBD, int LBH, int LBW>
__global__ __launch_bounds__(256, 2) void k2_up_fwd(const float* __restrict__ x,
                                                    const float* __restrict__ wp,
                                                    const float* __restrict__ bias,
                                                    float* __restrict__ y, int N, int D, int H, int W,
                                                    int Cin, int Cout, int nbd, int nbh, int nbw,
                                                    float* __restrict__ stat_part) {
    using C = UpCfg<LBD, LBH, LBW>;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;

    // each XCD walks one contiguous eighth of the block list; the two D parities of a brick are
    // neighbours in it (they read the same input)
    int b = blockIdx.x;
    if ((gridDim.x & 7) == 0) b = (b & 7) * (gridDim.x >> 3) + (b >> 3);
    const int ad = b & 1; b >>= 1;
    const int bx = b % nbw; b /= nbw;
    const int by = b % nbh; b /= nbh;
    const int bz = b % nbd;
    const int n = b / nbd;
    const int z0 = bz * LBD, y0 = by * LBH, x0 = bx * LBW;      // low-resolution origin of the brick
    const int co0 = blockIdx.y * C::NT;
    const int Dl = D >> 1, Hl = H >> 1, Wl = W >> 1;
    const int nchunk = Cin / C::CK;

    f32x16 acc[4][C::AM];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int a = 0; a < C::AM; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][a][r] = 0.f;

    // LDS offset of this lane's A row for collapsed tap (0,0,0), parities (0,0)
    int aoff[C::AM];
#pragma unroll
    for (int a = 0; a < C::AM; ++a) {
        int m = (wm * C::AM + a) * 32 + li;
        if (m >= C::M) m = 0;                   // padding row: reads a valid voxel, never stored
        const int dz = m / (LBH * LBW), dy = (m / LBW) % LBH, dx = m % LBW;
        aoff[a] = ((dz * C::HH + dy) * C::HW + dx) * C::CKP + 4 * lh;
    }
    // this lane's B fragment inside a packed [64 co][8 ci] slice
    const float* wlane = wp + ((long)(blockIdx.y * 2 + ad) * nchunk) * C::TAPS * C::SLICE + (wn * 32 + li) * 8 + 4 * lh;
    constexpr int PD = SIV_UPD;
    static_assert(C::TAPS % PD == 0, "ring index must not depend on the chunk");
    float4 bring[PD];

    // halo staging: item = (halo voxel, channel half); unconditional loads (out-of-volume voxels read
    // voxel 0 and are zeroed when consumed), addresses hoisted out of the chunk loop
    float4 hreg[C::NH];
    long hoff[C::NH];
    unsigned hmask = 0;
#pragma unroll
    for (int i = 0; i < C::NH; ++i) {
        const int idx = tid + C::NTHR * i;
        const int hv = idx >> 1, q = idx & 1;
        const int hx = hv % C::HW, hy = (hv / C::HW) % C::HH, hz = hv / (C::HW * C::HH);
        const int gz = z0 + ad - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        const bool ok = hv < C::HV && gz >= 0 && gz < Dl && gy >= 0 && gy < Hl && gx >= 0 && gx < Wl;
        hoff[i] = ok ? ((((long)n * Dl + gz) * Hl + gy) * Wl + gx) * Cin + 4 * q : 0;
        hmask |= (unsigned)ok << i;
    }
    auto load_halo = [&](int c0) {
#pragma unroll
        for (int i = 0; i < C::NH; ++i) hreg[i] = *reinterpret_cast<const float4*>(x + hoff[i] + c0);
    };
    load_halo(0);
#pragma unroll
    for (int i = 0; i < PD; ++i) bring[i] = *reinterpret_cast<const float4*>(wlane + i * C::SLICE);
    // pin the first loads: the loop is then entered with nothing in flight and the compiler counts,
    // per wait, the younger loads that may stay outstanding (loads return in order)
#define SIV_PIN(V) asm volatile("" : "+v"((V).x), "+v"((V).y), "+v"((V).z), "+v"((V).w))
#pragma unroll
    for (int i = 0; i < C::NH; ++i) SIV_PIN(hreg[i]);
#pragma unroll
    for (int i = 0; i < PD; ++i) SIV_PIN(bring[i]);
#undef SIV_PIN

    for (int cc = 0; cc < nchunk; ++cc) {
        __syncthreads();  // previous chunk's MFMAs are done with the halo
#pragma unroll
        for (int i = 0; i < C::NH; ++i) {
            const int idx = tid + C::NTHR * i;
            if (!((hmask >> i) & 1)) hreg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((idx >> 1) < C::HV)
                *reinterpret_cast<float4*>(lds + (idx >> 1) * C::CKP + 4 * (idx & 1)) = hreg[i];
        }
        __syncthreads();
        load_halo((cc + 1 < nchunk ? cc + 1 : cc) * C::CK);
        const float* wc = wlane + (long)cc * C::TAPS * C::SLICE;
        float4 anext[C::AM];
#pragma unroll
        for (int a = 0; a < C::AM; ++a) anext[a] = *reinterpret_cast<const float4*>(lds + aoff[a]);
#pragma unroll
        for (int t = 0; t < C::TAPS; ++t) {
            const int pos = t & 3;               // (H parity, W parity): the accumulator set
            float4 av[C::AM];
            const float4 bv = bring[t % PD];
#pragma unroll
            for (int a = 0; a < C::AM; ++a) av[a] = anext[a];
            bring[t % PD] = *reinterpret_cast<const float4*>(wc + (t + PD) * C::SLICE);
            if (t + 1 < C::TAPS) {
                const int t1 = t + 1;
                const int ah = (t1 >> 1) & 1, aw = t1 & 1, sd = t1 >> 4, sh = (t1 >> 3) & 1, sw = (t1 >> 2) & 1;
                const int tapoff = ((sd * C::HH + ah + sh) * C::HW + aw + sw) * C::CKP;
#pragma unroll
                for (int a = 0; a < C::AM; ++a)
                    anext[a] = *reinterpret_cast<const float4*>(lds + aoff[a] + tapoff);
            }
            __builtin_amdgcn_sched_barrier(0);
#define SIV_USTEP(F)                                                                        \
    _Pragma("unroll") for (int a = 0; a < C::AM; ++a)                                       \
        acc[pos][a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a].F, bv.F, acc[pos][a], 0, 0, 0);
            SIV_USTEP(x)
            SIV_USTEP(y)
            SIV_USTEP(z)
            SIV_USTEP(w)
#undef SIV_USTEP
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    up_fwd_store<LBD, LBH, LBW>(acc, lds, bias, y, stat_part, n, z0, y0, x0, ad, co0, D, H, W, Cout);
}

// Along one axis, the original taps k that collapse onto source s (0/1) for output parity a:
//   a = 0: s = 0 <- {0}, s = 1 <- {1, 2};   a = 1: s = 0 <- {0, 1}, s = 1 <- {2}
__device__ __forceinline__ void collapsed_range(int a, int s, int& lo, int& hi) {
    lo = a == 0 ? (s == 0 ? 0 : 1) : (s == 0 ? 0 : 2);
    hi = a == 0 ? (s == 0 ? 0 : 2) : (s == 0 ? 1 : 2);
}

// Packed, pre-summed weights: wp[ct][ad][cc][t = (sd, sh, sw, ah, aw)][co (64)][ci (8)] from
// w[Kout][27][Kin]; zero padded in co.
__global__ __launch_bounds__(256) void k2_up_pack_weights(const float* __restrict__ w,
                                                          float* __restrict__ wp, int Kin, int Kout) {
    const int nchunk = Kin / 8, ntile = (Kout + 63) / 64;
    const long total = (long)ntile * 2 * nchunk * 32 * 64 * 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7);
        long r = i >> 3;
        const int co = (int)(r & 63); r >>= 6;
        const int t = (int)(r & 31); r >>= 5;
        const int cc = (int)(r % nchunk); r /= nchunk;
        const int ad = (int)(r & 1);
        const int ct = (int)(r >> 1);
        const int aw = t & 1, ah = (t >> 1) & 1, sw = (t >> 2) & 1, sh = (t >> 3) & 1, sd = t >> 4;
        const int kco = ct * 64 + co, kci = cc * 8 + j;
        float v = 0.f;
        if (kco < Kout) {
            int d0, d1, h0, h1, w0, w1;
            collapsed_range(ad, sd, d0, d1);
            collapsed_range(ah, sh, h0, h1);
            collapsed_range(aw, sw, w0, w1);
            for (int kd = d0; kd <= d1; ++kd)
                for (int kh = h0; kh <= h1; ++kh)
                    for (int kw = w0; kw <= w1; ++kw) v += w[((long)kco * 27 + (kd * 3 + kh) * 3 + kw) * Kin + kci];
        }
        wp[i] = v;
    }
}

// ---------------------------------------------------------------------------------
// The same convolution in product form over (H, W).  Along one axis the two outputs of a
// low-resolution voxel j are
//   y[2j]   = w0 x[j-1] + (w1 + w2) x[j] = w0 (x[j-1] - x[j]) + S x[j]
//   y[2j+1] = (w0 + w1) x[j] + w2 x[j+1] = w2 (x[j+1] - x[j]) + S x[j],      S = w0 + w1 + w2
// i.e. three products (P-, P0, P+) for two outputs.  Over (H, W) that is 9 products per
// low-resolution voxel instead of 16, with the z axis collapsed as before (2 taps per D parity):
// 18 multiply-adds per 4 outputs instead of 32.  The differences are formed from the raw halo while
// the operand is fetched (nine LDS reads and twelve subtractions per row tile and plane feed 36
// matrix instructions); the nine accumulator sets are folded into the four parity outputs in the
// epilogue: y[ah][aw] = P[s(ah)][s(aw)] + P[s(ah)][0] + P[0][s(aw)] + P[0][0].
// One row tile per wave (64-row bricks): 9 x 16 accumulator registers.
// wp: see k2_up_pack_weights_prod.
// ---------------------------------------------------------------------------------
template <int LBD, int LBH, int LBW>
__global__ __launch_bounds__(256, 2) void k2_up_fwd_prod(const float* __restrict__ x,
                                                         const float* __restrict__ wp,
                                                         const float* __restrict__ bias,
                                                         float* __restrict__ y, int N, int D, int H, int W,
                                                         int Cin, int Cout, int nbd, int nbh, int nbw,
                                                         float* __restrict__ stat_part) {
    using C = UpCfg<LBD, LBH, LBW>;
    static_assert(C::AM == 1, "one row tile per wave");
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;

    int b = blockIdx.x;
    if ((gridDim.x & 7) == 0) b = (b & 7) * (gridDim.x >> 3) + (b >> 3);
    const int ad = b & 1; b >>= 1;
    const int bx = b % nbw; b /= nbw;
    const int by = b % nbh; b /= nbh;
    const int bz = b % nbd;
    const int n = b / nbd;
    const int z0 = bz * LBD, y0 = by * LBH, x0 = bx * LBW;
    const int co0 = blockIdx.y * C::NT;
    const int Dl = D >> 1, Hl = H >> 1, Wl = W >> 1;
    const int nchunk = Cin / C::CK;

    f32x16 acc[9];             // [ph * 3 + pw], p = 0, 1, 2 standing for (-, 0, +)
#pragma unroll
    for (int p = 0; p < 9; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    int m = wm * 32 + li;
    if (m >= C::M) m = 0;
    const int aoff = (((m / (LBH * LBW)) * C::HH + (m / LBW) % LBH) * C::HW + m % LBW) * C::CKP + 4 * lh;
    constexpr int NTAP = 18, PD = 3;
    // Weights and halo go through buffer descriptors: per-lane offsets stay constant, everything that moves
    // is scalar, and a halo voxel outside the volume (offset beyond the descriptor) reads zeros -- no 64-bit
    // address arithmetic and no masking on the vector ALU, whose instructions take matrix-pipe time here.
    const __amdgpu_buffer_rsrc_t wrs = buf_rsrc(wp, (long)gridDim.y * 2 * nchunk * NTAP * C::SLICE * 4);
    const unsigned wvo = ((wn * 32 + li) * 8 + 4 * lh) * 4;
    const int wso0 = (int)(blockIdx.y * 2 + ad) * nchunk * NTAP * C::SLICE * 4;
    float4 bring[PD];

    const __amdgpu_buffer_rsrc_t xrs = buf_rsrc(x + (long)n * Dl * Hl * Wl * Cin, (long)Dl * Hl * Wl * Cin * 4);
    float4 hreg[C::NH];
    unsigned hoff[C::NH];
#pragma unroll
    for (int i = 0; i < C::NH; ++i) {
        const int idx = tid + C::NTHR * i;
        const int hv = idx >> 1, q = idx & 1;
        const int hx = hv % C::HW, hy = (hv / C::HW) % C::HH, hz = hv / (C::HW * C::HH);
        const int gz = z0 + ad - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        const bool ok = hv < C::HV && gz >= 0 && gz < Dl && gy >= 0 && gy < Hl && gx >= 0 && gx < Wl;
        hoff[i] = ok ? (unsigned)((((gz * Hl + gy) * Wl + gx) * Cin + 4 * q) * 4) : 0x80000000u;
    }
    auto load_halo = [&](int c0) {
#pragma unroll
        for (int i = 0; i < C::NH; ++i) hreg[i] = buf_load16(xrs, hoff[i], c0 * 4);
    };
    load_halo(0);
#pragma unroll
    for (int i = 0; i < PD; ++i) bring[i] = buf_load16(wrs, wvo, wso0 + i * C::SLICE * 4);
#define SIV_PIN(V) asm volatile("" : "+v"((V).x), "+v"((V).y), "+v"((V).z), "+v"((V).w))
#pragma unroll
    for (int i = 0; i < C::NH; ++i) SIV_PIN(hreg[i]);
#pragma unroll
    for (int i = 0; i < PD; ++i) SIV_PIN(bring[i]);
#undef SIV_PIN

    // four-channel differences as two packed two-channel subtractions (vector instructions take matrix-pipe
    // time on a SIMD that runs fp32 matrix instructions, so their count matters)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    struct f32x4 { f32x2 lo, hi; };
    // (spelled as the instruction: the compiler splits a packed subtraction whose halves feed different
    // matrix instructions into two scalar ones)
    auto pk_sub = [](f32x2 a, f32x2 c) {
        f32x2 r;
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(c));
        return r;
    };
    auto sub4 = [&](f32x4 a, f32x4 c) { return f32x4{pk_sub(a.lo, c.lo), pk_sub(a.hi, c.hi)}; };
    for (int cc = 0; cc < nchunk; ++cc) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < C::NH; ++i) {
            const int idx = tid + C::NTHR * i;
            if ((idx >> 1) < C::HV)
                *reinterpret_cast<float4*>(lds + (idx >> 1) * C::CKP + 4 * (idx & 1)) = hreg[i];
        }
        __syncthreads();
        load_halo((cc + 1 < nchunk ? cc + 1 : cc) * C::CK);
        const int wso = wso0 + cc * NTAP * C::SLICE * 4;
#pragma unroll
        for (int sd = 0; sd < 2; ++sd) {
            // 3x3 neighbourhood of the row's voxel in plane sd, differences along W, then along H
            f32x4 d[3][3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float* row = lds + aoff + ((sd * C::HH + r) * C::HW) * C::CKP;
                const f32x4 c0 = *reinterpret_cast<const f32x4*>(row);
                const f32x4 c1 = *reinterpret_cast<const f32x4*>(row + C::CKP);
                const f32x4 c2 = *reinterpret_cast<const f32x4*>(row + 2 * C::CKP);
                d[r][0] = sub4(c0, c1);
                d[r][1] = c1;
                d[r][2] = sub4(c2, c1);
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ph = t / 3, pw = t % 3, tt = sd * 9 + t;
                const f32x4 av = ph == 0 ? sub4(d[0][pw], d[1][pw]) : ph == 1 ? d[1][pw] : sub4(d[2][pw], d[1][pw]);
                const float4 bv = bring[tt % PD];
                bring[tt % PD] = buf_load16(wrs, wvo, wso + (tt + PD) * C::SLICE * 4);
                __builtin_amdgcn_sched_barrier(0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.lo.x, bv.x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.lo.y, bv.y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.hi.x, bv.z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.hi.y, bv.w, acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // fold the nine product sets into the four (H, W) parity outputs
    f32x16 out4[4][1];
#pragma unroll
    for (int pos = 0; pos < 4; ++pos) {
        const int sh = (pos >> 1) ? 2 : 0, sw = (pos & 1) ? 2 : 0;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            out4[pos][0][r] = (acc[sh * 3 + sw][r] + acc[sh * 3 + 1][r]) + (acc[3 + sw][r] + acc[4][r]);
    }
    up_fwd_store<LBD, LBH, LBW>(out4, lds, bias, y, stat_part, n, z0, y0, x0, ad, co0, D, H, W, Cout);
}

// Weights of the product form: wp[ct][ad][cc][t = sd * 9 + ph * 3 + pw][co (64)][ci (8)] = sum over the
// taps kd that collapse onto (ad, sd) and over kh in T(ph), kw in T(pw) of w[co][kd][kh][kw][ci], with
// T(-) = {0}, T(0) = {0, 1, 2}, T(+) = {2}; zero padded in co.
__global__ __launch_bounds__(256) void k2_up_pack_weights_prod(const float* __restrict__ w,
                                                               float* __restrict__ wp, int Kin, int Kout) {
    const int nchunk = Kin / 8, ntile = (Kout + 63) / 64;
    const long total = (long)ntile * 2 * nchunk * 18 * 64 * 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7);
        long r = i >> 3;
        const int co = (int)(r & 63); r >>= 6;
        const int t = (int)(r % 18); r /= 18;
        const int cc = (int)(r % nchunk); r /= nchunk;
        const int ad = (int)(r & 1);
        const int ct = (int)(r >> 1);
        const int sd = t / 9, ph = (t % 9) / 3, pw = t % 3;
        const int kco = ct * 64 + co, kci = cc * 8 + j;
        float v = 0.f;
        if (kco < Kout) {
            int d0, d1;
            collapsed_range(ad, sd, d0, d1);
            const int h0 = ph == 2 ? 2 : 0, h1 = ph == 0 ? 0 : 2;
            const int w0 = pw == 2 ? 2 : 0, w1 = pw == 0 ? 0 : 2;
            for (int kd = d0; kd <= d1; ++kd)
                for (int kh = h0; kh <= h1; ++kh)
                    for (int kw = w0; kw <= w1; ++kw) v += w[((long)kco * 27 + (kd * 3 + kh) * 3 + kw) * Kin + kci];
        }
        wp[i] = v;
    }
}

// The same convolution with every fp32 product formed from three bf16 pieces per operand on the bf16
// matrix cores (scheme and operand packing: conv_wino_b3.hip).  The staged halo holds [h | m | l] pieces
// of 8 channels per voxel (48 bytes, the fp32 footprint); wp: see k2_up_pack_weights_b3.
template <int LBD, int LBH, int LBW>
__global__ __launch_bounds__(256, 2) void k2_up_fwd_b3(const float* __restrict__ x,
                                                       const uint4* __restrict__ wp,
                                                    const float* __restrict__ bias,
                                                    float* __restrict__ y, int N, int D, int H, int W,
                                                    int Cin, int Cout, int nbd, int nbh, int nbw,
                                                    float* __restrict__ stat_part) {
    using C = UpCfg<LBD, LBH, LBW>;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;

    // each XCD walks one contiguous eighth of the block list; the two D parities of a brick are
    // neighbours in it (they read the same input)
    int b = blockIdx.x;
    if ((gridDim.x & 7) == 0) b = (b & 7) * (gridDim.x >> 3) + (b >> 3);
    const int ad = b & 1; b >>= 1;
    const int bx = b % nbw; b /= nbw;
    const int by = b % nbh; b /= nbh;
    const int bz = b % nbd;
    const int n = b / nbd;
    const int z0 = bz * LBD, y0 = by * LBH, x0 = bx * LBW;      // low-resolution origin of the brick
    const int co0 = blockIdx.y * C::NT;
    const int Dl = D >> 1, Hl = H >> 1, Wl = W >> 1;
    const int nchunk = Cin / C::CK;

    f32x16 acc[4][C::AM];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int a = 0; a < C::AM; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][a][r] = 0.f;

    // LDS offsets of this lane's three A operands (pieces h|h, m|h, l|m by lane half), tap 0, parities (0,0)
    int aoff1[C::AM], aoff2[C::AM], aoff3[C::AM];
#pragma unroll
    for (int a = 0; a < C::AM; ++a) {
        int m = (wm * C::AM + a) * 32 + li;
        if (m >= C::M) m = 0;                   // padding row: reads a valid voxel, never stored
        const int dz = m / (LBH * LBW), dy = (m / LBW) % LBH, dx = m % LBW;
        const int row = ((dz * C::HH + dy) * C::HW + dx) * C::CKP;
        aoff1[a] = row;
        aoff2[a] = row + (lh ? 0 : 4);
        aoff3[a] = row + (lh ? 4 : 8);
    }
    // this lane's two B operands of a tap: slices of [64 co][2 lane halves][8 ci] bf16, 256 uint4 per tap
    const uint4* wlane = wp + ((long)(blockIdx.y * 2 + ad) * nchunk) * C::TAPS * 256 + (wn * 32 + li) * 2 + lh;
    constexpr int PD = 4;
    uint4 bxr[PD], byr[PD];

    // halo staging: item = (halo voxel, channel half); unconditional loads (out-of-volume voxels read
    // voxel 0 and are zeroed when consumed), addresses hoisted out of the chunk loop
    float4 hreg[C::NH];
    long hoff[C::NH];
    unsigned hmask = 0;
#pragma unroll
    for (int i = 0; i < C::NH; ++i) {
        const int idx = tid + C::NTHR * i;
        const int hv = idx >> 1, q = idx & 1;
        const int hx = hv % C::HW, hy = (hv / C::HW) % C::HH, hz = hv / (C::HW * C::HH);
        const int gz = z0 + ad - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        const bool ok = hv < C::HV && gz >= 0 && gz < Dl && gy >= 0 && gy < Hl && gx >= 0 && gx < Wl;
        hoff[i] = ok ? ((((long)n * Dl + gz) * Hl + gy) * Wl + gx) * Cin + 4 * q : 0;
        hmask |= (unsigned)ok << i;
    }
    auto load_halo = [&](int c0) {
#pragma unroll
        for (int i = 0; i < C::NH; ++i) hreg[i] = *reinterpret_cast<const float4*>(x + hoff[i] + c0);
    };
    load_halo(0);
#pragma unroll
    for (int i = 0; i < PD; ++i) {
        bxr[i] = wlane[i * 256];
        byr[i] = wlane[i * 256 + 128];
    }
    // pin the first loads: the loop is then entered with nothing in flight and the compiler counts,
    // per wait, the younger loads that may stay outstanding (loads return in order)
#define SIV_PIN(V) asm volatile("" : "+v"((V).x), "+v"((V).y), "+v"((V).z), "+v"((V).w))
#pragma unroll
    for (int i = 0; i < C::NH; ++i) SIV_PIN(hreg[i]);
#pragma unroll
    for (int i = 0; i < PD; ++i) { SIV_PIN(bxr[i]); SIV_PIN(byr[i]); }
#undef SIV_PIN

    for (int cc = 0; cc < nchunk; ++cc) {
        __syncthreads();  // previous chunk's MFMAs are done with the halo
#pragma unroll
        for (int i = 0; i < C::NH; ++i) {
            const int idx = tid + C::NTHR * i;
            if (!((hmask >> i) & 1)) hreg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((idx >> 1) < C::HV) {
                uint2 ph, pm, pl;
                split3(hreg[i], ph, pm, pl);
                float* d = lds + (idx >> 1) * C::CKP + 2 * (idx & 1);
                *reinterpret_cast<uint2*>(d) = ph;
                *reinterpret_cast<uint2*>(d + 4) = pm;
                *reinterpret_cast<uint2*>(d + 8) = pl;
            }
        }
        __syncthreads();
        load_halo((cc + 1 < nchunk ? cc + 1 : cc) * C::CK);
        const uint4* wc = wlane + (long)cc * C::TAPS * 256;
        bf16x8 n1, n2, n3;            // A operands, read one (tap, row tile) unit ahead of their MFMAs
        n1 = *reinterpret_cast<const bf16x8*>(lds + aoff1[0]);
        n2 = *reinterpret_cast<const bf16x8*>(lds + aoff2[0]);
        n3 = *reinterpret_cast<const bf16x8*>(lds + aoff3[0]);
#pragma unroll
        for (int u = 0; u < C::TAPS * C::AM; ++u) {
            const int t = u / C::AM, a = u % C::AM;
            const int pos = t & 3;               // (H parity, W parity): the accumulator set
            const bf16x8 a1 = n1, a2 = n2, a3 = n3;
            const bf16x8 bx8 = __builtin_bit_cast(bf16x8, bxr[t % PD]);
            const bf16x8 by8 = __builtin_bit_cast(bf16x8, byr[t % PD]);
            if (u + 1 < C::TAPS * C::AM) {
                const int t1 = (u + 1) / C::AM, a1i = (u + 1) % C::AM;
                const int ah = (t1 >> 1) & 1, aw = t1 & 1, sd = t1 >> 4, sh = (t1 >> 3) & 1, sw = (t1 >> 2) & 1;
                const int tapoff = ((sd * C::HH + ah + sh) * C::HW + aw + sw) * C::CKP;
                n1 = *reinterpret_cast<const bf16x8*>(lds + aoff1[a1i] + tapoff);
                n2 = *reinterpret_cast<const bf16x8*>(lds + aoff2[a1i] + tapoff);
                n3 = *reinterpret_cast<const bf16x8*>(lds + aoff3[a1i] + tapoff);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[pos][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, bx8, acc[pos][a], 0, 0, 0);   // al*bh | am*bm
            acc[pos][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, by8, acc[pos][a], 0, 0, 0);   // am*bh | ah*bl
            acc[pos][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bx8, acc[pos][a], 0, 0, 0);   // ah*bh | ah*bm
            __builtin_amdgcn_sched_barrier(0);
            if (a == C::AM - 1) {
                bxr[t % PD] = wc[(t + PD) * 256];
                byr[t % PD] = wc[(t + PD) * 256 + 128];
            }
        }
    }

    up_fwd_store<LBD, LBH, LBW>(acc, lds, bias, y, stat_part, n, z0, y0, x0, ad, co0, D, H, W, Cout);
}

// Packed, pre-summed weights as bf16 pieces: per (co tile, ad, ci chunk, tap) two operand slices of
// [64 co][2 lane halves][8 ci]: slice 0 = (h | m), slice 1 = (h | l).
__global__ __launch_bounds__(256) void k2_up_pack_weights_b3(const float* __restrict__ w,
                                                             __bf16* __restrict__ wp, int Kin, int Kout) {
    const int nchunk = Kin / 8, ntile = Kout / 64;
    const long total = (long)ntile * 2 * nchunk * 32 * 2 * 64 * 2 * 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7);
        long r = i >> 3;
        const int half = (int)(r & 1); r >>= 1;
        const int co = (int)(r & 63); r >>= 6;
        const int which = (int)(r & 1); r >>= 1;
        const int t = (int)(r & 31); r >>= 5;
        const int cc = (int)(r % nchunk); r /= nchunk;
        const int ad = (int)(r & 1);
        const int ct = (int)(r >> 1);
        const int aw = t & 1, ah = (t >> 1) & 1, sw = (t >> 2) & 1, sh = (t >> 3) & 1, sd = t >> 4;
        const int kco = ct * 64 + co, kci = cc * 8 + j;
        int d0, d1, h0, h1, w0, w1;
        collapsed_range(ad, sd, d0, d1);
        collapsed_range(ah, sh, h0, h1);
        collapsed_range(aw, sw, w0, w1);
        float v = 0.f;
        for (int kd = d0; kd <= d1; ++kd)
            for (int kh = h0; kh <= h1; ++kh)
                for (int kw = w0; kw <= w1; ++kw) v += w[((long)kco * 27 + (kd * 3 + kh) * 3 + kw) * Kin + kci];
        __bf16 h, m, l;
        split3_scalar(v, h, m, l);
        wp[i] = half == 0 ? h : (which == 0 ? m : l);
    }
}

static bool up_wide10(int W) { return (W / 2) % 10 == 0 && (W / 2) % 8 != 0; }

// Output channels in tiles of 64; a last partial tile (the 16- and 32-channel layers of the FC-latent decoder)
// is zero padded in the packed weights and masked in the stores (fp32 kernels; the split-bf16 packing needs
// whole tiles).
bool k2_up_fwd_supported(int Cin, int Cout) { return Cin % 8 == 0 && Cout % 16 == 0; }
size_t k2_up_fwd_ws_bytes(int Cin, int Cout) {
    // sized for the split-bf16 packing (two 2 KB operand slices per tap), which is the larger one
    return ((size_t)((Cout + 63) / 64) * 2 * (Cin / 8) * 32 + 4) * 4096;
}
// number of [2][Cout] partial-statistics rows the kernel emits (D, H, W: output extents)
// the product form from 32 K output voxels upwards (and on request); its bricks hold one low-resolution plane
static bool up_use_prod(int N, int D, int H, int W, bool split, bool force) {
    return !split && (force || (long)N * D * H * W >= (1L << 15));
}
long k2_up_fwd_blocks(int N, int D, int H, int W, bool split, bool force_prod) {
    if (up_use_prod(N, D, H, W, split, force_prod)) {
        if (up_wide10(W)) return (long)N * (D / 2) * cdiv(H / 2, 6) * cdiv(W / 2, 10) * 2;
        return (long)N * (D / 2) * cdiv(H / 2, 8) * cdiv(W / 2, 8) * 2;
    }
    if (up_wide10(W)) return (long)N * (D / 2) * cdiv(H / 2, 12) * cdiv(W / 2, 10) * 2;
    return (long)N * cdiv(D / 2, 2) * cdiv(H / 2, 8) * cdiv(W / 2, 8) * 2;
}

template <int LBD, int LBH, int LBW>
static int up_launch_prod(const float* x, const void* wp, const float* bias, float* y, int N, int D, int H,
                          int W, int Cin, int Cout, hipStream_t s, float* stat_part) {
    using C = UpCfg<LBD, LBH, LBW>;
    SIV_CHECK_ARG((long)(D / 2) * (H / 2) * (W / 2) * Cin < (1L << 29),
                  "conv3d_k3(upsampled, product form): sample too large for 32-bit byte offsets");
    const int nbd = (int)cdiv(D / 2, LBD), nbh = (int)cdiv(H / 2, LBH), nbw = (int)cdiv(W / 2, LBW);
    const long nblk = (long)N * nbd * nbh * nbw * 2;
    if (nblk > 0x7fffffffL) {
        set_error("conv3d_k3(upsampled): too many bricks");
        return SIV_EINVAL;
    }
    k2_up_fwd_prod<LBD, LBH, LBW><<<dim3((unsigned)nblk, (unsigned)((Cout + 63) / 64)), 256, C::LDS_BYTES, s>>>(
        x, (const float*)wp, bias, y, N, D, H, W, Cin, Cout, nbd, nbh, nbw, stat_part);
    SIV_CHECK_LAUNCH("conv3d_k3(upsampled, product form)");
    return SIV_OK;
}

template <int LBD, int LBH, int LBW>
static int up_launch_cfg(const float* x, const void* wp, const float* bias, float* y, int N, int D, int H,
                         int W, int Cin, int Cout, hipStream_t s, float* stat_part, bool split) {
    using C = UpCfg<LBD, LBH, LBW>;
    const int nbd = (int)cdiv(D / 2, LBD), nbh = (int)cdiv(H / 2, LBH), nbw = (int)cdiv(W / 2, LBW);
    const long nblk = (long)N * nbd * nbh * nbw * 2;
    if (nblk > 0x7fffffffL) {
        set_error("conv3d_k3(upsampled): too many bricks");
        return SIV_EINVAL;
    }
    const dim3 grid((unsigned)nblk, (unsigned)((Cout + 63) / 64));
    if (split)
        k2_up_fwd_b3<LBD, LBH, LBW><<<grid, 256, C::LDS_BYTES, s>>>(x, (const uint4*)wp, bias, y, N, D, H, W, Cin,
                                                                    Cout, nbd, nbh, nbw, stat_part);
    else
        k2_up_fwd<LBD, LBH, LBW><<<grid, 256, C::LDS_BYTES, s>>>(x, (const float*)wp, bias, y, N, D, H, W, Cin, Cout,
                                                                 nbd, nbh, nbw, stat_part);
    SIV_CHECK_LAUNCH("conv3d_k3(upsampled)");
    return SIV_OK;
}

// x: [N][D/2][H/2][W/2][Cin] -> y: [N][D][H][W][Cout].  stat_part (optional): receives
// [k2_up_fwd_blocks()][2][Cout] per-block sums / sums of squares of the bias-free output; it must not
// overlap the first k2_up_fwd_ws_bytes() bytes of ws.  split: products from bf16 pieces.
int k2_up_fwd_launch(const float* x, const float* w, const float* bias, float* y, int N, int D, int H,
                     int W, int Cin, int Cout, void* ws, size_t ws_bytes, hipStream_t s, float* stat_part,
                     bool split, bool force_prod) {
    SIV_CHECK_WS(k2_up_fwd_ws_bytes(Cin, Cout), ws_bytes, "conv3d_k3(upsampled)");
    if (Cout % 64 != 0) split = false;
    bool need_pack;
    if (up_use_prod(N, D, H, W, split, force_prod)) {
        const long total = (long)((Cout + 63) / 64) * 2 * (Cin / 8) * 18 * 512;
        long g = cdiv(total, 256);
        if (g > 4096) g = 4096;
        void* wpk = pack_slot(w, PF_UP_PROD, 0, 0, Cin, Cout, k2_up_fwd_ws_bytes(Cin, Cout), ws, &need_pack);
        if (need_pack) {
            k2_up_pack_weights_prod<<<(unsigned)g, 256, 0, s>>>(w, (float*)wpk, Cin, Cout);
            SIV_CHECK_PACK_LAUNCH("conv3d_k3(pack upsampled weights, product form)", wpk);
        }
        if (up_wide10(W)) return up_launch_prod<1, 6, 10>(x, wpk, bias, y, N, D, H, W, Cin, Cout, s, stat_part);
        return up_launch_prod<1, 8, 8>(x, wpk, bias, y, N, D, H, W, Cin, Cout, s, stat_part);
    }
    void* wpk = pack_slot(w, split ? PF_UP_B3 : PF_UP, 0, 0, Cin, Cout, k2_up_fwd_ws_bytes(Cin, Cout), ws, &need_pack);
    if (need_pack) {
        if (split) {
            const long total = (long)(Cout / 64) * 2 * (Cin / 8) * 32 * 2 * 64 * 2 * 8;
            long g = cdiv(total, 256);
            if (g > 4096) g = 4096;
            k2_up_pack_weights_b3<<<(unsigned)g, 256, 0, s>>>(w, (__bf16*)wpk, Cin, Cout);
        } else {
            const long total = (long)((Cout + 63) / 64) * 2 * (Cin / 8) * 32 * 512;
            long g = cdiv(total, 256);
            if (g > 4096) g = 4096;
            k2_up_pack_weights<<<(unsigned)g, 256, 0, s>>>(w, (float*)wpk, Cin, Cout);
        }
        SIV_CHECK_PACK_LAUNCH("conv3d_k3(pack upsampled weights)", wpk);
    }
    if (up_wide10(W)) return up_launch_cfg<1, 12, 10>(x, wpk, bias, y, N, D, H, W, Cin, Cout, s, stat_part, split);
    return up_launch_cfg<2, 8, 8>(x, wpk, bias, y, N, D, H, W, Cin, Cout, s, stat_part, split);
}

// ---------------------------------------------------------------------------------
// weight gradient of the convolution of an upsampled input
// ---------------------------------------------------------------------------------
// With the taps collapsed as above, the weight gradient splits by output parity class p = (ad,ah,aw):
//   G[p][s][co][ci] = sum_i dy[2i + a][co] * x[i + a - 1 + s][ci],   s in {0,1}^3
//   dw[co][kd][kh][kw][ci] = sum_p G[p][s(ad,kd), s(ah,kh), s(aw,kw)]   (s(a,k): a=0: 0,1,1; a=1: 0,0,1)
// i.e. 8 classes x 8 taps over the low-resolution voxels: 8 multiply-adds per output voxel and channel
// pair instead of 27.
//
// Block = (range of 2x4xBW low-resolution bricks) x (parity class) x (64 co x 64 ci tile).
// Wave w owns co half (w&1), ci half (w>>1) and the 8 collapsed taps.  Voxels are the K dimension.
// MODE 1, 2 -- narrow layers (Cin <= 32 and Cout <= 32), as in k3_wgrad_mfma: the four waves split the
// brick's voxels (two of its eight rows each, a partial slab per wave); MODE 2 (Cin <= 16): a tile's 32
// columns hold two taps x 16 input channels, four matrix instructions per k-step instead of eight.
template <int BW_>
struct UwCfg {
    static constexpr int BD = 2, BH = 4, BW = BW_;
    static constexpr int M = BD * BH * BW;                 // 64 (80) voxels = 32 (40) k-steps
    static constexpr int HD = BD + 1, HH = BH + 1, HW = BW + 1;
    static constexpr int HV = HD * HH * HW;                // 135 (165) low-resolution halo voxels
    static constexpr int CP = 64;
    static constexpr int NDY = M * 16 / 256;               // float4 of dy per thread
    static constexpr int NX = (HV * 16 + 255) / 256;       // float4 of x per thread (last partial)
    static constexpr size_t LDS_BYTES = (size_t)(M * CP + HV * CP) * sizeof(float);
};

template <int BW_>
struct UwStage {
    using C = UwCfg<BW_>;
    float4 dyr[C::NDY];
    float4 xr[C::NX];

    // D, H, W: extents of dy (high resolution, even); x is [N][D/2][H/2][W/2][Cin]
    __device__ __forceinline__ void load(const float* __restrict__ x, const float* __restrict__ dy,
                                         long br, int ad, int ah, int aw, int co0, int ci0, int D, int H,
                                         int W, int Cin, int Cout, int nbd, int nbh, int nbw) {
        const int tid = threadIdx.x;
        const int Dl = D >> 1, Hl = H >> 1, Wl = W >> 1;
        long b = br;
        const int bx = (int)(b % nbw); b /= nbw;
        const int by = (int)(b % nbh); b /= nbh;
        const int bz = (int)(b % nbd);
        const int n = (int)(b / nbd);
        const int z0 = bz * C::BD, y0 = by * C::BH, x0 = bx * C::BW;
#pragma unroll
        for (int i = 0; i < C::NDY; ++i) {
            const int idx = tid + 256 * i;
            const int m = idx >> 4, q = idx & 15;
            const int gz = z0 + m / (C::BH * C::BW), gy = y0 + (m / C::BW) % C::BH, gx = x0 + m % C::BW;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gz < Dl && gy < Hl && gx < Wl && co0 + 4 * q < Cout)
                v = *reinterpret_cast<const float4*>(
                    dy + ((((long)n * D + 2 * gz + ad) * H + 2 * gy + ah) * W + 2 * gx + aw) * Cout + co0 + 4 * q);
            dyr[i] = v;
        }
#pragma unroll
        for (int i = 0; i < C::NX; ++i) {
            const int idx = tid + 256 * i;
            const int hv = idx >> 4, q = idx & 15;
            const int hx = hv % C::HW, hy = (hv / C::HW) % C::HH, hz = hv / (C::HW * C::HH);
            const int gz = z0 + hz + ad - 1, gy = y0 + hy + ah - 1, gx = x0 + hx + aw - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (hv < C::HV && ci0 + 4 * q < Cin && gz >= 0 && gz < Dl && gy >= 0 && gy < Hl && gx >= 0 &&
                gx < Wl)
                v = *reinterpret_cast<const float4*>(
                    x + ((((long)n * Dl + gz) * Hl + gy) * Wl + gx) * Cin + ci0 + 4 * q);
            xr[i] = v;
        }
    }
    __device__ __forceinline__ void store(float* dyl, float* xl) const {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < C::NDY; ++i) {
            const int idx = tid + 256 * i;
            *reinterpret_cast<float4*>(dyl + (idx >> 4) * C::CP + 4 * (idx & 15)) = dyr[i];
        }
#pragma unroll
        for (int i = 0; i < C::NX; ++i) {
            const int idx = tid + 256 * i;
            if ((idx >> 4) < C::HV)
                *reinterpret_cast<float4*>(xl + (idx >> 4) * C::CP + 4 * (idx & 15)) = xr[i];
        }
    }
};

template <int BW_, int MODE>
__global__ __launch_bounds__(256, 2) void k2_up_wgrad(const float* __restrict__ x,
                                                      const float* __restrict__ dy,
                                                      float* __restrict__ part, int N, int D, int H, int W,
                                                      int Cin, int Cout, int nbd, int nbh, int nbw,
                                                      int bricks_per_block) {
    using C = UwCfg<BW_>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* dyl = lds;                 // [M voxels][64 co]
    float* xl = lds + C::M * C::CP;   // [HV halo voxels][64 ci]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int cs = MODE ? 0 : wave & 1, is = MODE ? 0 : wave >> 1;
    const int p = blockIdx.y, ad = p >> 2, ah = (p >> 1) & 1, aw = p & 1;
    const int ntci = (Cin + 63) / 64;
    const int co0 = (blockIdx.z / ntci) * 64, ci0 = (blockIdx.z % ntci) * 64;
    const bool active = co0 + cs * 32 < Cout && ci0 + is * 32 < Cin;
    const long nbricks = (long)N * nbd * nbh * nbw;
    const long b0 = (long)blockIdx.x * bricks_per_block;
    long b1 = b0 + bricks_per_block;
    if (b1 > nbricks) b1 = nbricks;

    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    UwStage<BW_> st;
    if (b0 < b1) st.load(x, dy, b0, ad, ah, aw, co0, ci0, D, H, W, Cin, Cout, nbd, nbh, nbw);
    for (long br = b0; br < b1; ++br) {
        __syncthreads();              // the previous brick's MFMAs are done with LDS
        st.store(dyl, xl);
        __syncthreads();
        if (br + 1 < b1)              // next brick's global loads fly during this brick's MFMAs
            st.load(x, dy, br + 1, ad, ah, aw, co0, ci0, D, H, W, Cin, Cout, nbd, nbh, nbw);
        const float* ap = dyl + lh * C::CP + cs * 32 + li;
        const float* bp = xl + lh * C::CP + is * 32 + li;
        if constexpr (MODE != 0) {
            // this wave's rows 2 wave, 2 wave + 1 of the brick's BD x BH rows: BW / 2 k-steps each
            constexpr int HB = C::BW / 2;
            static_assert(C::BD * C::BH == 8, "two brick rows per wave");
            const int row = 2 * wave;
            const float* aw_ = ap + 2 * (row * HB) * C::CP;
            const int wbase = ((row / C::BH) * C::HH + row % C::BH) * C::HW;
            if constexpr (MODE == 2) {
                // column li = (tap 2 p2 + (li >> 4), input channel li & 15) of tile p2
                const float* bq[4];
#pragma unroll
                for (int p2 = 0; p2 < 4; ++p2) {
                    const int t = 2 * p2 + (li >> 4);
                    bq[p2] = xl + lh * C::CP + (li & 15) + (wbase + ((t >> 2) * C::HH + ((t >> 1) & 1)) * C::HW + (t & 1)) * C::CP;
                }
#pragma unroll
                for (int j = 0; j < 2 * HB; ++j) {
                    const int off = ((j / HB) * C::HW + 2 * (j % HB)) * C::CP;
                    const float a = aw_[2 * j * C::CP];
#pragma unroll
                    for (int p2 = 0; p2 < 4; ++p2)
                        acc[p2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq[p2][off], acc[p2], 0, 0, 0);
                }
            } else {
                const float* bw_ = bp + wbase * C::CP;
#pragma unroll 1
                for (int jr = 0; jr < 2; ++jr) {                  // (one brick row at a time: register budget)
#pragma unroll
                    for (int jc = 0; jc < HB; ++jc) {
                        const float a = aw_[2 * (jr * HB + jc) * C::CP];
#pragma unroll
                        for (int t = 0; t < 8; ++t)
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                a, bw_[(jr * C::HW + 2 * jc + ((t >> 2) * C::HH + ((t >> 1) & 1)) * C::HW + (t & 1)) * C::CP],
                                acc[t], 0, 0, 0);
                    }
                }
            }
        } else if (active)
#pragma unroll 4
        for (int kk = 0; kk < C::M / 2; ++kk) {
            // voxels 2kk + lh: dx = 2*(kk % (BW/2)) + lh, dy = (kk / (BW/2)) % BH, dz = kk / (BW/2*BH)
            constexpr int HB = C::BW / 2;
            const int hbase = ((kk / (HB * C::BH)) * C::HH + (kk / HB) % C::BH) * C::HW + 2 * (kk % HB);
            const float a = ap[2 * kk * C::CP];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const float bvv = bp[(hbase + ((t >> 2) * C::HH + ((t >> 1) & 1)) * C::HW + (t & 1)) * C::CP];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bvv, acc[t], 0, 0, 0);
            }
        }
    }

    // partial slab: part[blockIdx.x][co][p * 8 + s][ci]; narrow layers: one slab per wave, part[blockIdx.x * 4 + wave]
    float* pp = part + (MODE ? (long)blockIdx.x * 4 + wave : (long)blockIdx.x) * Cout * 64 * Cin;
    if constexpr (MODE == 2) {
#pragma unroll
        for (int p2 = 0; p2 < 4; ++p2) {
            const int t = 2 * p2 + (li >> 4), ci = li & 15;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (co < Cout && ci < Cin) pp[((long)co * 64 + p * 8 + t) * Cin + ci] = acc[p2][r];
            }
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + cs * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int ci = ci0 + is * 32 + li;
            if (co < Cout && ci < Cin) pp[((long)co * 64 + p * 8 + t) * Cin + ci] = acc[t][r];
        }
    }
}

// Two-stage, fixed-order reduction of the partial slabs.  Stage 1 (slab_sum_inplace) adds slabs
// 1..nslab-1 into slab 0, one element per thread (coalesced, nslab loads deep); stage 2 maps the 64
// (class, collapsed tap) sums onto the 27 taps:
//   dw[co][k][ci] = (accumulate ? dw : 0) + sum_p slab0[co][p * 8 + s(p, k)][ci]
__global__ __launch_bounds__(256) void slab_sum_inplace(float* __restrict__ part, int nslab, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 s = *reinterpret_cast<const float4*>(part + 4 * i);
        int q = 1;
        for (; q + 8 <= nslab; q += 8) {              // eight slabs in flight, added in slab order
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(part + (long)(q + u) * 4 * n4 + 4 * i);
#pragma unroll
            for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        for (; q < nslab; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(part + (long)q * 4 * n4 + 4 * i);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        *reinterpret_cast<float4*>(part + 4 * i) = s;
    }
}

__global__ __launch_bounds__(256) void up_wgrad_taps(const float* __restrict__ g, int Cin, int Cout,
                                                     float* __restrict__ dw, int accumulate) {
    const long n = (long)Cout * 27 * Cin;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Cin);
        const int k = (int)((i / Cin) % 27);
        const long co = i / ((long)Cin * 27);
        const int kd = k / 9, kh = (k / 3) % 3, kw = k % 3;
        float s = 0.f;
        for (int p = 0; p < 8; ++p) {
            const int ad = p >> 2, ah = (p >> 1) & 1, aw = p & 1;
            const int sd = ad == 0 ? (kd > 0) : (kd > 1), sh = ah == 0 ? (kh > 0) : (kh > 1),
                      sw = aw == 0 ? (kw > 0) : (kw > 1);
            s += g[(co * 64 + p * 8 + sd * 4 + sh * 2 + sw) * Cin + ci];
        }
        dw[i] = accumulate ? dw[i] + s : s;
    }
}

bool k2_up_wgrad_supported(int Cin, int Cout) { return Cin % 4 == 0 && Cout % 4 == 0 && Cin >= 4 && Cout >= 8; }

// x: [N][D/2][H/2][W/2][Cin], dy: [N][D][H][W][Cout], dw: [Cout][27][Cin]
int k2_up_wgrad_launch(const float* x, const float* dy, float* dw, int N, int D, int H, int W, int Cin,
                       int Cout, int accumulate, void* ws, size_t ws_bytes, hipStream_t s) {
    const int Dl = D / 2, Hl = H / 2, Wl = W / 2;
    const bool wide = (Wl % 10 == 0) && (Wl % 8 != 0);
    const int nbd = (Dl + 1) / 2, nbh = (Hl + 3) / 4, nbw = wide ? Wl / 10 : (Wl + 7) / 8;
    const long nbricks = (long)N * nbd * nbh * nbw;
    const int tiles = ((Cout + 63) / 64) * ((Cin + 63) / 64);
    const long wsize = (long)Cout * 64 * Cin;
    // 8*tiles*nsplit blocks at 2 resident blocks per CU: at most two full rounds
    long nsplit = 1024 / (8L * tiles);
    if (nsplit < 1) nsplit = 1;
    if (nsplit > nbricks) nsplit = nbricks;
    const int mode = (Cin <= 32 && Cout <= 32) ? (Cin <= 16 ? 2 : 1) : 0;
    const int wave_slabs = mode ? 4 : 1;                            // narrow layers: a partial slab per wave
    const long max_by_ws = (long)(ws_bytes / (wsize * sizeof(float))) / wave_slabs;
    if (nsplit > max_by_ws) nsplit = max_by_ws;
    if (nsplit < 1) {
        set_error("conv3d_k3_wgrad(upsampled): workspace too small (need %zu, have %zu)",
                  (size_t)wsize * sizeof(float), ws_bytes);
        return SIV_EWORKSPACE;
    }
    const int bpb = (int)cdiv(nbricks, nsplit);
    nsplit = cdiv(nbricks, bpb);
    dim3 grid((unsigned)nsplit, 8, (unsigned)tiles);
#define SIV_UWM(BWW, MD)                                                                            \
    k2_up_wgrad<BWW, MD><<<grid, 256, UwCfg<BWW>::LDS_BYTES, s>>>(x, dy, (float*)ws, N, D, H, W, Cin, Cout, nbd, \
                                                                  nbh, nbw, bpb)
#define SIV_UW(BWW)                                                                                 \
    do {                                                                                            \
        if (mode == 2) SIV_UWM(BWW, 2); else if (mode == 1) SIV_UWM(BWW, 1); else SIV_UWM(BWW, 0);    \
    } while (0)
    if (wide) SIV_UW(10); else SIV_UW(8);
#undef SIV_UW
#undef SIV_UWM
    SIV_CHECK_LAUNCH("conv3d_k3_wgrad(upsampled)");
    long g = cdiv(wsize / 4, 256);        // Cin % 4 == 0: whole float4s
    if (g > 4096) g = 4096;
    if (nsplit * wave_slabs > 1) {
        slab_sum_inplace<<<(unsigned)g, 256, 0, s>>>((float*)ws, (int)nsplit * wave_slabs, wsize / 4);
        SIV_CHECK_LAUNCH("conv3d_k3_wgrad(upsampled slab sum)");
    }
    g = cdiv((long)Cout * 27 * Cin, 256);
    if (g > 4096) g = 4096;
    up_wgrad_taps<<<(unsigned)g, 256, 0, s>>>((const float*)ws, Cin, Cout, dw, accumulate);
    SIV_CHECK_LAUNCH("conv3d_k3_wgrad(upsampled taps)");
    return SIV_OK;
}

// ---------------------------------------------------------------------------------
// input gradient of the convolution of an upsampled input
// ---------------------------------------------------------------------------------
// The gradient with respect to the LOW-RESOLUTION input is the 2x2x2 sum-pool of the gradient with
// respect to the upsampled one, which makes it a stride-2 convolution of dy with a 4x4x4 kernel:
//   dx[j][ci] = sum_{u in {-1,0,1,2}^3} sum_co q[u][co][ci] * dy[2j + u][co]
// where along each axis q[-1] = w2, q[0] = w1 + w2, q[1] = w0 + w1, q[2] = w0 (wk = tap k of that
// axis): 64 taps per low-resolution voxel = 8 multiply-adds per dy voxel and channel pair instead of
// 27, and no full-resolution intermediate.
//
//   M rows  = low-resolution voxels of a brick (256 or 240), N = 64 input channels of the convolution
//   K       = 64 taps x Cout, walked as (8-channel chunk, u along D): each step stages the dy planes
//             2z + u of the brick (W-parity split, so that the stride-2 reads are unit-stride in LDS)
//             and runs the 16 (u along H, u along W) taps on them
// SIV_UPD_LINE=0 (build flag) keeps the four-plane box kernel at every size
#ifndef SIV_UPD_LINE
#define SIV_UPD_LINE 1
#endif

// box form: log2 of the number of 8-channel chunks whose steps are grouped (see k2_up_dgrad)
__host__ __device__ inline int up_dgrad_group_shift(int nchunk) { return nchunk % 4 == 0 ? 2 : nchunk % 2 == 0 ? 1 : 0; }

template <int LBD_, int LBH_, int LBW_>
struct UdCfg {
    static constexpr int LBD = LBD_, LBH = LBH_, LBW = LBW_;
    static constexpr int CK = 8, CKP = 12;
    static constexpr int M = LBD * LBH * LBW;             // rows in use (<= 256)
    static constexpr int NTHR = 256, NT = 64, AM = 4;     // 2 x 2 waves, 4 row tiles per wave
    static constexpr int HH = 2 * LBH + 2, HWF = 2 * LBW + 2, HWH = LBW + 1;
    static constexpr int PITCH = LBW == 8 ? 240 : 272;    // floats per halo row; = 16 mod 32 (bank spread)
    static constexpr int HV = LBD * HH * HWF;             // dy voxels staged per step
    static constexpr int NH = (HV * 2 + NTHR - 1) / NTHR;
    static constexpr int V_FLOATS = LBD * HH * PITCH;
    static constexpr int TAPS = 16, SLICE = NT * 8;
    static constexpr int EPI_STRIDE = 40, EPI_FLOATS = 32 * EPI_STRIDE;
    static constexpr int SLACK = PITCH + HWH * CKP;         // what the box sums of the last halo row touch behind the image
    static constexpr int PARK = NH > 8 ? 4 : 0;             // load offsets per thread parked in LDS instead of registers
    static constexpr size_t LDS_BYTES = (size_t)(V_FLOATS + SLACK + PARK * NTHR) * sizeof(float);
    static_assert(PITCH >= 2 * HWH * CKP && PITCH % 32 == 16, "row pitch");
    static_assert(M <= 256 && M > 224, "four row tiles per wave, two waves along M");
    static_assert(V_FLOATS >= 4 * EPI_FLOATS, "epilogue tiles fit");
};

// dy: [N][D][H][W][Cout] (D, H, W even); dx: [N][D/2][H/2][W/2][Cin]; wp: see k2_up_pack_dgrad.
// SPLIT: every fp32 product formed from three bf16 pieces per operand on the bf16 matrix cores (scheme
// and operand packing: conv_wino_b3.hip); the staged dy voxels then hold [h | m | l] pieces of 8 channels.
// BOX (fp32 only): along H and W the four collapsed taps are regrouped around the ORIGINAL taps,
//   sum_u q[u] dy[2j+u] = w0 (dy[2j+1] + dy[2j+2]) + w1 (dy[2j] + dy[2j+1]) + w2 (dy[2j-1] + dy[2j]),
// so a tap (kh, kw) multiplies the 2x2 in-plane box sum of dy that starts at (2j+1-kh, 2j+1-kw): 9 taps
// per staged plane instead of 16 (36 instead of 64 per low-resolution voxel).  The box sums are formed
// once per step, in place: the thread that staged voxel (y, x) adds its three neighbours out of the raw
// image and writes the sum back where the voxel was (vector instructions do not overlap fp32 matrix
// instructions on a SIMD, so summing per tap and row tile -- 432 additions per wave and step -- cost a
// third of the step).  Layout and epilogue are unchanged; only the order of the fp32 additions differs
// from the 16-tap form.
// SIV_UDSTAMP (diagnostic build only, tools/ubench/up_dgrad_stamps.hip): lane 0 of every wave of the first
// blocks records the shader clock around the phases of steps 8..11 into a buffer nothing else reads.
#ifdef SIV_UDSTAMP
__device__ unsigned long long ud_stamps[4096 * 4 * 32];
#define SIV_UDS(i)                                                                                           \
    do {                                                                                                     \
        if (lane == 0 && blockIdx.x < 4096 && step >= 8 && step < 12)                                        \
            ud_stamps[((long)blockIdx.x * 4 + wave) * 32 + (step - 8) * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define SIV_UDS(i)
#endif

template <int LBD, int LBH, int LBW, bool SPLIT, bool BOX = false>
__global__ __launch_bounds__(256, 2) void k2_up_dgrad(const float* __restrict__ dy,
                                                      const void* __restrict__ wpv,
                                                      float* __restrict__ dx, int N, int D, int H, int W,
                                                      int Cin, int Cout, int nbd, int nbh, int nbw) {
    using C = UdCfg<LBD, LBH, LBW>;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;

    int b = blockIdx.x;
    if ((gridDim.x & 7) == 0) b = (b & 7) * (gridDim.x >> 3) + (b >> 3);
    const int bx = b % nbw; b /= nbw;
    const int by = b % nbh; b /= nbh;
    const int bz = b % nbd;
    const int n = b / nbd;
    const int z0 = bz * LBD, y0 = by * LBH, x0 = bx * LBW;      // low-resolution origin of the brick
    const int ci0 = blockIdx.y * C::NT;
    const int Dl = D >> 1, Hl = H >> 1, Wl = W >> 1;
    const int nstep = (Cout / C::CK) * 4;                       // (chunk, u along D)
    // gridDim.z > 1 (box form only): block z walks its share of the steps and leaves a partial dx in slab z
    // of a buffer of gridDim.z outputs (added up by the caller's next kernel)
    const int s_begin = BOX ? (int)blockIdx.z * (nstep / (int)gridDim.z) : 0;
    const int s_end = BOX ? s_begin + nstep / (int)gridDim.z : nstep;
    if (BOX) dx += (long)blockIdx.z * N * (D >> 1) * (H >> 1) * (W >> 1) * Cin;

    // LDS offset of this lane's A row for tap (u_h, u_w) = (-1, -1); with SPLIT three of them, one per
    // A operand (pieces h|h, m|h, l|m by lane half)
    int aoff[C::AM], aoff2[C::AM], aoff3[C::AM];
#pragma unroll
    for (int a = 0; a < C::AM; ++a) {
        int m = (wm * C::AM + a) * 32 + li;
        if (m >= C::M) m = 0;                   // padding row: reads a valid voxel, never stored
        const int dz = m / (LBH * LBW), dyy = (m / LBW) % LBH, dxx = m % LBW;
        const int row = (dz * C::HH + 2 * dyy) * C::PITCH + dxx * C::CKP;
        aoff[a] = SPLIT ? row : row + 4 * lh;
        aoff2[a] = row + (lh ? 0 : 4);
        aoff3[a] = row + (lh ? 4 : 8);
    }
    constexpr int PD = BOX ? 3 : 4;             // divides the taps of a step: the ring index restarts with it
    static_assert((BOX ? 9 : C::TAPS) % PD == 0, "ring index must not depend on the step");
    // fp32: one [64 ci][8 co] slice per tap; SPLIT: two operand slices of [64 ci][2 lane halves][8 co] bf16
    constexpr int NTAP = BOX ? 9 : C::TAPS;
    const float* wlane = (const float*)wpv + (long)blockIdx.y * nstep * NTAP * C::SLICE + (wn * 32 + li) * 8 + 4 * lh;
    const uint4* wlq = (const uint4*)wpv + (long)blockIdx.y * nstep * C::TAPS * 256 + (wn * 32 + li) * 2 + lh;
    float4 bring[PD];
    uint4 bxr[PD], byr[PD];

    // staging: item = (dy voxel of the staged planes, channel half).  Offsets are relative to the
    // brick's first voxel in plane 2*z0 (always inside the tensor); the plane parity u is added per step.
    const long plane = (long)H * W * Cout;
    const float* base = dy + (((long)n * D + 2 * z0) * H + 2 * y0) * (long)W * Cout + (long)2 * x0 * Cout;
    float4 hreg[C::NH];
    int hoff[C::NH], hdst[C::NH];
    unsigned hmask = 0, curmask = 0;
    int hpz[C::NH];
    // BOX: the loads go through a buffer descriptor of sample n.  A voxel outside the sample in z lands
    // below or beyond the descriptor's range by itself, one outside in y or x is sent there (UP_OUTSIDE),
    // and either reads zeros: no masks, no selects, one 32-bit addition per load.
    const __amdgpu_buffer_rsrc_t drs = buf_rsrc(dy + (long)n * D * H * W * Cout, BOX ? (long)D * H * W * Cout * 4 : 0);
    const __amdgpu_buffer_rsrc_t wrs = buf_rsrc(wpv, BOX ? (long)gridDim.y * nstep * 9 * C::SLICE * 4 : 0);
    const unsigned wvo = ((wn * 32 + li) * 8 + 4 * lh) * 4;
    const int wso0 = (int)blockIdx.y * nstep * 9 * C::SLICE * 4;
    auto wstep = [&](int step) {    // BOX: byte offset of the nine slices of a step
        return wso0 + step * 9 * C::SLICE * 4;
    };
    constexpr unsigned UP_OUTSIDE = 0xC0000000u;
    // BOX: LDS offset of the voxel to the right minus this voxel's.  Voxel hv = (tid >> 1) + 128 i and the
    // halo row length is even, so the parity of the column is the same for every item of a thread.
    static_assert(C::HWF % 2 == 0 && C::NTHR % 4 == 0, "column parity per thread");
    const int hnx = ((tid >> 1) & 1) ? (1 - C::HWH) * C::CKP : C::HWH * C::CKP;
#pragma unroll
    for (int i = 0; i < C::NH; ++i) {
        const int idx = tid + C::NTHR * i;
        const int hv = idx >> 1, q = idx & 1;
        const int hx = hv % C::HWF, hy = (hv / C::HWF) % C::HH, pz = hv / (C::HWF * C::HH);
        const int gy = 2 * y0 - 1 + hy, gx = 2 * x0 - 1 + hx;
        const bool ok = hv < C::HV && gy >= 0 && gy < H && gx >= 0 && gx < W;
        if constexpr (BOX) {
            hoff[i] = ok ? (int)((((2 * (z0 + pz) * H + gy) * W + gx) * Cout + 4 * q) * 4) : (int)UP_OUTSIDE;
            hpz[i] = 0;
        } else {
            hoff[i] = ok ? ((2 * pz * H + (hy - 1)) * W + (hx - 1)) * Cout + 4 * q : 0;
            hpz[i] = ok ? 2 * (z0 + pz) : -4;       // high-resolution plane for u = 0 (-4: never valid)
        }
        hdst[i] = (pz * C::HH + hy) * C::PITCH + ((hx & 1) * C::HWH + (hx >> 1)) * C::CKP + (SPLIT ? 2 : 4) * q;
        hmask |= (unsigned)(hv < C::HV) << i;
    }
    // BOX with eleven items per thread: the offsets of the last four live in LDS (one 16-byte read per step);
    // in registers they were spilled to scratch memory and reloaded in front of every step's loads.
    constexpr int NHR = BOX ? C::NH - C::PARK : C::NH;
    int* const park = reinterpret_cast<int*>(lds + C::V_FLOATS + C::SLACK) + 4 * tid;
    if constexpr (BOX && C::PARK > 0)
        *reinterpret_cast<int4*>(park) = make_int4(hoff[NHR], hoff[NHR + 1], hoff[NHR + 2], hoff[NHR + 3]);
    // BOX: the steps of up to four chunks -- the 128 bytes of a voxel that arrive as one cache line -- are taken
    // together, step = ((group * 4 + u index) * G + chunk in group): the four steps that use a line follow each
    // other (in the order (chunk, u) a line was fetched again sixteen steps later, by then from HBM: 8.5x the
    // bytes of dy per launch, which bounded the kernel).
    const int gsh = BOX ? up_dgrad_group_shift(Cout / C::CK) : 0;
    auto load_halo = [&](int step) {
        const int c0 = BOX ? (((step >> (2 + gsh)) << gsh) + (step & ((1 << gsh) - 1))) * C::CK : (step >> 2) * C::CK;
        const int u = ((step >> gsh) & 3) - 1;
        unsigned mk = 0;
        if constexpr (BOX) {
            const int delta = (u * H * W * Cout + c0) * 4;
            int4 pk = make_int4(0, 0, 0, 0);
            if constexpr (C::PARK > 0) pk = *reinterpret_cast<const int4*>(park);
#pragma unroll
            for (int i = 0; i < NHR; ++i) hreg[i] = buf_load16(drs, (unsigned)(hoff[i] + delta), 0);
            if constexpr (C::PARK > 0) {
                hreg[NHR + 0] = buf_load16(drs, (unsigned)(pk.x + delta), 0);
                hreg[NHR + 1] = buf_load16(drs, (unsigned)(pk.y + delta), 0);
                hreg[NHR + 2] = buf_load16(drs, (unsigned)(pk.z + delta), 0);
                hreg[NHR + 3] = buf_load16(drs, (unsigned)(pk.w + delta), 0);
            }
            return ~0u;
        }
#pragma unroll
        for (int i = 0; i < C::NH; ++i) {
            const int gz = hpz[i] + u;
            const bool ok = gz >= 0 && gz < D;
            const long off = ok ? (long)hoff[i] + u * plane + c0 : 0;
            hreg[i] = *reinterpret_cast<const float4*>(base + off);
            mk |= (unsigned)ok << i;
        }
        return mk;
    };
    unsigned nextmask = load_halo(s_begin);
#define SIV_PIN(V) asm volatile("" : "+v"((V).x), "+v"((V).y), "+v"((V).z), "+v"((V).w))
#pragma unroll
    for (int i = 0; i < PD; ++i) {
        if constexpr (SPLIT) {
            bxr[i] = wlq[i * 256];
            byr[i] = wlq[i * 256 + 128];
        } else if constexpr (BOX) {
            bring[i] = buf_load16(wrs, wvo, wstep(s_begin) + i * C::SLICE * 4);
        } else {
            bring[i] = *reinterpret_cast<const float4*>(wlane + i * C::SLICE);
        }
    }
#pragma unroll
    for (int i = 0; i < C::NH; ++i) SIV_PIN(hreg[i]);
#pragma unroll
    for (int i = 0; i < PD; ++i) {
        if constexpr (SPLIT) { SIV_PIN(bxr[i]); SIV_PIN(byr[i]); } else { SIV_PIN(bring[i]); }
    }
#undef SIV_PIN

    f32x16 acc[C::AM];
#pragma unroll
    for (int a = 0; a < C::AM; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    for (int step = s_begin; step < s_end; ++step) {
        curmask = nextmask;
        SIV_UDS(0);
        __syncthreads();  // previous step's MFMAs are done with the planes
        SIV_UDS(1);
#pragma unroll
        for (int i = 0; i < C::NH; ++i) {
            if (!((curmask >> i) & 1)) hreg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((hmask >> i) & 1) {
                if constexpr (SPLIT) {
                    uint2 ph, pm, pl;
                    split3(hreg[i], ph, pm, pl);
                    float* d = lds + hdst[i];          // hdst carries 2 * q here
                    *reinterpret_cast<uint2*>(d) = ph;
                    *reinterpret_cast<uint2*>(d + 4) = pm;
                    *reinterpret_cast<uint2*>(d + 8) = pl;
                } else {
                    *reinterpret_cast<float4*>(lds + hdst[i]) = hreg[i];
                }
            }
        }
        __syncthreads();
        SIV_UDS(2);
        if constexpr (BOX) {
            // raw image -> box image, in place: this voxel + right + below + below right
#pragma unroll
            for (int i = 0; i < C::NH; ++i) {
                if ((hmask >> i) & 1) {
                    // (the last row and column of the halo have no box of their own: what they add up --
                    // the first row of the next plane, the slack row behind the image, column 1 -- is
                    // never read)
                    const float* d = lds + hdst[i];
                    const float4 r1 = *reinterpret_cast<const float4*>(d + hnx);
                    const float4 r2 = *reinterpret_cast<const float4*>(d + C::PITCH);
                    const float4 r3 = *reinterpret_cast<const float4*>(d + C::PITCH + hnx);
                    hreg[i].x = (hreg[i].x + r1.x) + (r2.x + r3.x);
                    hreg[i].y = (hreg[i].y + r1.y) + (r2.y + r3.y);
                    hreg[i].z = (hreg[i].z + r1.z) + (r2.z + r3.z);
                    hreg[i].w = (hreg[i].w + r1.w) + (r2.w + r3.w);
                }
                if (i % 2 == 1) __builtin_amdgcn_sched_barrier(0);     // two items' reads in flight, not all
            }
            SIV_UDS(3);
            __syncthreads();  // every neighbour has been read
            SIV_UDS(4);
#pragma unroll
            for (int i = 0; i < C::NH; ++i)
                if ((hmask >> i) & 1) *reinterpret_cast<float4*>(lds + hdst[i]) = hreg[i];
            __syncthreads();
        }
        SIV_UDS(5);
        nextmask = load_halo(step + 1 < s_end ? step + 1 : step);
        SIV_UDS(6);
        if constexpr (SPLIT) {
            const uint4* wq = wlq + (long)step * C::TAPS * 256;
            bf16x8 n1, n2, n3;            // A operands, read one (tap, row tile) unit ahead of their MFMAs
            n1 = *reinterpret_cast<const bf16x8*>(lds + aoff[0]);
            n2 = *reinterpret_cast<const bf16x8*>(lds + aoff2[0]);
            n3 = *reinterpret_cast<const bf16x8*>(lds + aoff3[0]);
#pragma unroll
            for (int u = 0; u < C::TAPS * C::AM; ++u) {
                const int t = u / C::AM, a = u % C::AM;
                const bf16x8 a1 = n1, a2 = n2, a3 = n3;
                const bf16x8 bx8 = __builtin_bit_cast(bf16x8, bxr[t % PD]);
                const bf16x8 by8 = __builtin_bit_cast(bf16x8, byr[t % PD]);
                if (u + 1 < C::TAPS * C::AM) {
                    const int t1 = (u + 1) / C::AM, a1i = (u + 1) % C::AM;
                    const int uh = t1 >> 2, uw = t1 & 3;
                    const int tapoff = uh * C::PITCH + ((uw & 1) * C::HWH + (uw >> 1)) * C::CKP;
                    n1 = *reinterpret_cast<const bf16x8*>(lds + aoff[a1i] + tapoff);
                    n2 = *reinterpret_cast<const bf16x8*>(lds + aoff2[a1i] + tapoff);
                    n3 = *reinterpret_cast<const bf16x8*>(lds + aoff3[a1i] + tapoff);
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, bx8, acc[a], 0, 0, 0);   // al*bh | am*bm
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, by8, acc[a], 0, 0, 0);   // am*bh | ah*bl
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bx8, acc[a], 0, 0, 0);   // ah*bh | ah*bm
                __builtin_amdgcn_sched_barrier(0);
                if (a == C::AM - 1) {
                    bxr[t % PD] = wq[(t + PD) * 256];
                    byr[t % PD] = wq[(t + PD) * 256 + 128];
                }
            }
            continue;
        }
        const float* wc = wlane + (long)step * NTAP * C::SLICE;
        if constexpr (BOX) {
            // the box that starts at row e_h, column e_w of the W-parity-split image
            auto boxoff = [&](int t) {
                const int eh = t / 3, px = t % 3;
                return eh * C::PITCH + ((px & 1) * C::HWH + (px >> 1)) * C::CKP;
            };
            const int wso = wstep(step);
            // Row tile a's four matrix instructions run back to back (the accumulator is ready again when
            // the next one issues), so its operand register is free behind them and takes tap t + 1 at
            // once: 16 operand registers instead of 32.
            float4 av[C::AM];
#pragma unroll
            for (int a = 0; a < C::AM; ++a) av[a] = *reinterpret_cast<const float4*>(lds + aoff[a] + boxoff(0));
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float4 bv = bring[t % PD];
                bring[t % PD] = buf_load16(wrs, wvo, wso + (t + PD) * C::SLICE * 4);    // runs into the next step's slices
#pragma unroll
                for (int a = 0; a < C::AM; ++a) {
                    __builtin_amdgcn_sched_barrier(0);
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a].x, bv.x, acc[a], 0, 0, 0);
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a].y, bv.y, acc[a], 0, 0, 0);
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a].z, bv.z, acc[a], 0, 0, 0);
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a].w, bv.w, acc[a], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (t + 1 < 9) av[a] = *reinterpret_cast<const float4*>(lds + aoff[a] + boxoff(t + 1));
                }
            }
            SIV_UDS(7);
            continue;
        }
        float4 anext[C::AM];
#pragma unroll
        for (int a = 0; a < C::AM; ++a) anext[a] = *reinterpret_cast<const float4*>(lds + aoff[a]);
#pragma unroll
        for (int t = 0; t < C::TAPS; ++t) {
            float4 av[C::AM];
            const float4 bv = bring[t % PD];
#pragma unroll
            for (int a = 0; a < C::AM; ++a) av[a] = anext[a];
            bring[t % PD] = *reinterpret_cast<const float4*>(wc + (t + PD) * C::SLICE);
            if (t + 1 < C::TAPS) {
                const int uh = (t + 1) >> 2, uw = (t + 1) & 3;
                const int tapoff = uh * C::PITCH + ((uw & 1) * C::HWH + (uw >> 1)) * C::CKP;
#pragma unroll
                for (int a = 0; a < C::AM; ++a)
                    anext[a] = *reinterpret_cast<const float4*>(lds + aoff[a] + tapoff);
            }
            __builtin_amdgcn_sched_barrier(0);
#define SIV_DSTEP(F)                                                                        \
    _Pragma("unroll") for (int a = 0; a < C::AM; ++a)                                       \
        acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a].F, bv.F, acc[a], 0, 0, 0);
            SIV_DSTEP(x)
            SIV_DSTEP(y)
            SIV_DSTEP(z)
            SIV_DSTEP(w)
#undef SIV_DSTEP
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // store: transpose each 32x32 tile through this wave's slice of LDS, write 16-byte channel runs
    float* tile = lds + wave * C::EPI_FLOATS;
    const int cq = lane & 7;
    const int ci = ci0 + wn * 32 + 4 * cq;
#pragma unroll
    for (int a = 0; a < C::AM; ++a) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) tile[((r & 3) + 8 * (r >> 2) + 4 * lh) * C::EPI_STRIDE + li] = acc[a][r];
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int row = ps * 8 + (lane >> 3);
            const int m = (wm * C::AM + a) * 32 + row;
            const int gz = z0 + m / (LBH * LBW), gy = y0 + (m / LBW) % LBH, gx = x0 + m % LBW;
            const float4 v = *reinterpret_cast<const float4*>(tile + row * C::EPI_STRIDE + 4 * cq);
            if (m < C::M && gz < Dl && gy < Hl && gx < Wl && ci < Cin)
                *reinterpret_cast<float4*>(dx + ((((long)n * Dl + gz) * Hl + gy) * Wl + gx) * Cin + ci) = v;
        }
    }
}

// ---------------------------------------------------------------------------------
// the box form, one whole cache line per staged voxel
// ---------------------------------------------------------------------------------
// k2_up_dgrad<.., BOX> stages four dy planes with 8 channels per voxel and step: 32 of the 128 bytes that a
// voxel's cache line delivers, and the line comes back for the other chunks and for the second (z, u) pair
// that uses the plane after hundreds of kilobytes of other lines -- measured at 64 -> 64 @80x96x80: 10.2 GB
// of 128-byte requests from the L2 to the fabric per launch for 1.26 GB of dy, 5.5 TB/s, which bounded the
// kernel.  Here a step stages ONE plane 2 (z0 + dz) + u with 32 channels per voxel (whole lines, used once per
// (dz, u)) and multiplies it into the row tiles of plane dz alone:
//   M rows  = 4 x 8 x 8 low-resolution voxels; wave (wm, wn) owns row tile wm of each plane dz (4 accumulators)
//             for the input-channel half wn
//   K       = walked as (32-channel group, u along D, dz): 9 box taps x four 8-channel groups = 36 units of
//             4 matrix instructions per step and wave, as many as in the four-plane form
// Weights: the box packing in the kernel's step order (k2_up_pack_dgrad_box), whose 36 slices per (group, u)
// are re-read for the four dz.  Staging, box sums in place and the epilogue follow the four-plane form.
// SIV_UL_EXP: timing-only ablation mask (never set in the library build; results are wrong with it):
// 1 = no weight loads in the multiply loop, 2 = no dy loads, 4 = no box phase (neighbour reads, sums, two
// barriers).  Measured at 64 -> 64 @80x96x80, batch 8: full 1.69 ms, 1: 1.63, 2: 1.59, 4: 1.54, all three
// 1.35 (the matrix instructions alone: 1.15 ms; 2400 blocks on 768 slots leave a last round an eighth full).
#ifndef SIV_UL_EXP
#define SIV_UL_EXP 0
#endif
// SIV_UL_BLOCKS blocks per CU (3: 168 registers, no parked offsets, sums pinned where they are formed),
// SIV_UL_PD weight slices in flight (divides 36), SIV_UL_PARK load offsets kept in LDS (0 or 4).
// Measured (blocks, slices, parked): (2, 4, 4) 1.68 ms, (2, 6, 4) 1.69, (2, 9, 4) 1.69, (3, 3, 0) 1.65.
#ifndef SIV_UL_BLOCKS
#define SIV_UL_BLOCKS 3
#endif
#ifndef SIV_UL_PD
#define SIV_UL_PD 3
#endif
#ifndef SIV_UL_PARK
#define SIV_UL_PARK 0
#endif
struct UlCfg {
    static constexpr int LBD = 4, LBH = 8, LBW = 8;
    static constexpr int CG = 32, CKP = 36;                 // channels per voxel and step; floats per voxel (bank spread)
    static constexpr int NTHR = 256, NT = 64, AM = 4;
    static constexpr int HH = 2 * LBH + 2, HWF = 2 * LBW + 2, HWH = LBW + 1;
    static constexpr int PITCH = 656;                       // floats per halo row: >= 2 * HWH * CKP, = 16 mod 32
    static constexpr int HV = HH * HWF;                     // dy voxels staged per step (324)
    static constexpr int NH = (HV * 8 + NTHR - 1) / NTHR;   // 16-byte items per thread (11)
    static constexpr int V_FLOATS = HH * PITCH;
    static constexpr int UNITS = 36, SLICE = NT * 8;
    static constexpr int EPI_STRIDE = 40, EPI_FLOATS = 32 * EPI_STRIDE;
    static constexpr int SLACK = PITCH + HWH * CKP + CG;    // what the box sums of the last halo row touch behind the image
    static constexpr int PARK = SIV_UL_PARK;                // load offsets per thread parked in LDS (0 or 4)
    static constexpr size_t LDS_BYTES = (size_t)(V_FLOATS + SLACK + PARK * NTHR) * sizeof(float);
    static_assert(PITCH >= 2 * HWH * CKP && PITCH % 32 == 16, "row pitch");
    static_assert(V_FLOATS >= 4 * EPI_FLOATS, "epilogue tiles fit");
};

__global__ __launch_bounds__(256, SIV_UL_BLOCKS) void k2_up_dgrad_line(const float* __restrict__ dy,
                                                           const float* __restrict__ wp,
                                                           float* __restrict__ dx, int N, int D, int H, int W,
                                                           int Cin, int Cout, int nbd, int nbh, int nbw) {
    using C = UlCfg;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;

    int b = blockIdx.x;
    if ((gridDim.x & 7) == 0) b = (b & 7) * (gridDim.x >> 3) + (b >> 3);
    const int bx = b % nbw; b /= nbw;
    const int by = b % nbh; b /= nbh;
    const int bz = b % nbd;
    const int n = b / nbd;
    const int z0 = bz * C::LBD, y0 = by * C::LBH, x0 = bx * C::LBW;      // low-resolution origin of the brick
    const int ci0 = blockIdx.y * C::NT;
    const int Dl = D >> 1, Hl = H >> 1, Wl = W >> 1;
    const int nseq = (Cout / C::CG) * 4;                        // (channel group, u along D); four steps (dz) each
    const int nstep = nseq * 4;

    // LDS offset of this lane's A row (row tile wm of the staged plane) for box (0, 0), channel group 0
    const int mrow = wm * 32 + li;
    const int aoff = (2 * (mrow / C::LBW)) * C::PITCH + (mrow % C::LBW) * C::CKP + 4 * lh;
    auto unitoff = [&](int u) {      // unit u = (8-channel group kq, box tap t = e_h * 3 + p_x)
        const int kq = u / 9, t = u % 9;
        const int eh = t / 3, px = t % 3;
        return eh * C::PITCH + ((px & 1) * C::HWH + (px >> 1)) * C::CKP + 8 * kq;
    };

    constexpr int PD = SIV_UL_PD;    // weight slices in flight; divides the 36 units of a step
    static_assert(C::UNITS % PD == 0, "ring index must not depend on the step");
    float4 bring[PD];
    const __amdgpu_buffer_rsrc_t drs = buf_rsrc(dy + (long)n * D * H * W * Cout, (long)D * H * W * Cout * 4);
    const __amdgpu_buffer_rsrc_t wrs = buf_rsrc(wp, (long)gridDim.y * nseq * C::UNITS * C::SLICE * 4);
    const unsigned wvo = ((wn * 32 + li) * 8 + 4 * lh) * 4;
    const int wso0 = (int)blockIdx.y * nseq * C::UNITS * C::SLICE * 4;
    constexpr unsigned UP_OUTSIDE = 0xC0000000u;

    // staging: item = (voxel of the staged plane, 16-byte piece of its 32 channels).  A thread keeps its piece
    // q and walks the 18 x 18 halo voxels in a regular pattern -- items 0..8: column v & 15 of rows 2 i + (v >> 4);
    // items 9, 10: columns 16 and 17 of rows (v >> 1) + 16 (i - 9), v = tid >> 3 -- so that the LDS addresses of
    // all items are two registers plus constants.  Global offsets are those of plane 2 * z0; the step adds the
    // plane (2 dz + u) and the channel group, and a plane outside the sample lands outside the descriptor by itself.
    static_assert(C::HH == 18 && C::HWF == 18 && C::NH == 11, "item pattern");
    float4 hreg[C::NH];
    int hoff[C::NH];
    const int sv = tid >> 3, sq = tid & 7;
    // item 10 covers rows 16 and 17 only: the other threads repeat their item 9 (same source, same destination,
    // same box sum) instead of branching
    const bool last_ok = tid < 32;
    auto item_row = [&](int i) { return i < 9 ? 2 * i + (sv >> 4) : (sv >> 1) + 16 * (i - 9); };
    const int colA = sv & 15, colB = 16 + (sv & 1);
    const int hdstA = (sv >> 4) * C::PITCH + ((colA & 1) * C::HWH + (colA >> 1)) * C::CKP + 4 * sq;
    const int hdstB = (sv >> 1) * C::PITCH + ((colB & 1) * C::HWH + (colB >> 1)) * C::CKP + 4 * sq;
    const int hdstL = last_ok ? hdstB + 16 * C::PITCH : hdstB;
    auto hdst = [&](int i) { return i < 9 ? hdstA + 2 * i * C::PITCH : i == 9 ? hdstB : hdstL; };
    // LDS offset of the voxel to the right minus this voxel's: the column parity is that of v for every item
    const int hnx = (sv & 1) ? (1 - C::HWH) * C::CKP : C::HWH * C::CKP;
#pragma unroll
    for (int i = 0; i < C::NH; ++i) {
        const int hy = i == 10 && !last_ok ? item_row(9) : item_row(i), hx = i < 9 ? colA : colB;
        const int gy = 2 * y0 - 1 + hy, gx = 2 * x0 - 1 + hx;
        const bool ok = hy < C::HH && gy >= 0 && gy < H && gx >= 0 && gx < W;
        hoff[i] = ok ? (int)((((2 * z0 * H + gy) * W + gx) * Cout + 4 * sq) * 4) : (int)UP_OUTSIDE;
    }
    constexpr int NHR = C::NH - C::PARK;
    int* const park = reinterpret_cast<int*>(lds + C::V_FLOATS + C::SLACK) + 4 * tid;
    if constexpr (C::PARK > 0)
        *reinterpret_cast<int4*>(park) = make_int4(hoff[NHR], hoff[NHR + 1], hoff[NHR + 2], hoff[NHR + 3]);
    auto load_halo = [&](int step) {
        const int dz = step & 3, u = ((step >> 2) & 3) - 1, c0 = (step >> 4) * C::CG;
        const int delta = ((2 * dz + u) * H * W * Cout + c0) * 4;
        int4 pk = make_int4(0, 0, 0, 0);
        if constexpr (C::PARK > 0) pk = *reinterpret_cast<const int4*>(park);
#pragma unroll
        for (int i = 0; i < NHR; ++i) hreg[i] = buf_load16(drs, (unsigned)(hoff[i] + delta), 0);
        if constexpr (C::PARK > 0) {
            hreg[NHR + 0] = buf_load16(drs, (unsigned)(pk.x + delta), 0);
            hreg[NHR + 1] = buf_load16(drs, (unsigned)(pk.y + delta), 0);
            hreg[NHR + 2] = buf_load16(drs, (unsigned)(pk.z + delta), 0);
            hreg[NHR + 3] = buf_load16(drs, (unsigned)(pk.w + delta), 0);
        }
    };
    load_halo(0);
#pragma unroll
    for (int i = 0; i < PD; ++i) bring[i] = buf_load16(wrs, wvo, wso0 + i * C::SLICE * 4);
#define SIV_PIN(V) asm volatile("" : "+v"((V).x), "+v"((V).y), "+v"((V).z), "+v"((V).w))
#pragma unroll
    for (int i = 0; i < C::NH; ++i) SIV_PIN(hreg[i]);
#pragma unroll
    for (int i = 0; i < PD; ++i) SIV_PIN(bring[i]);

    f32x16 acc[C::AM];
#pragma unroll
    for (int a = 0; a < C::AM; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    for (int seq = 0; seq < nseq; ++seq) {
#pragma unroll
        for (int dz = 0; dz < 4; ++dz) {
            const int step = seq * 4 + dz;
            SIV_UDS(0);
            __syncthreads();  // previous step's matrix instructions are done with the plane
            SIV_UDS(1);
#pragma unroll
            for (int i = 0; i < C::NH; ++i)
                *reinterpret_cast<float4*>(lds + hdst(i)) = hreg[i];
            __syncthreads();
            SIV_UDS(2);
            if (!(SIV_UL_EXP & 4)) {
            // raw image -> box image, in place: this voxel + right + below + below right (the last row and
            // column of the halo have no box of their own; what they add up is never read)
#pragma unroll
            for (int i = 0; i < C::NH; ++i) {
                const float* d = lds + hdst(i);
                const float4 r1 = *reinterpret_cast<const float4*>(d + hnx);
                const float4 r2 = *reinterpret_cast<const float4*>(d + C::PITCH);
                const float4 r3 = *reinterpret_cast<const float4*>(d + C::PITCH + hnx);
                hreg[i].x = (hreg[i].x + r1.x) + (r2.x + r3.x);
                hreg[i].y = (hreg[i].y + r1.y) + (r2.y + r3.y);
                hreg[i].z = (hreg[i].z + r1.z) + (r2.z + r3.z);
                hreg[i].w = (hreg[i].w + r1.w) + (r2.w + r3.w);
                // the sums are formed here, not where they are stored: three items' reads in flight, not all
                if (SIV_UL_BLOCKS > 2 && i >= 2) SIV_PIN(hreg[i - 2]);
                if (i % 2 == 1) __builtin_amdgcn_sched_barrier(0);
            }
            SIV_UDS(3);
            __syncthreads();  // every neighbour has been read
            SIV_UDS(4);
#pragma unroll
            for (int i = 0; i < C::NH; ++i)
                *reinterpret_cast<float4*>(lds + hdst(i)) = hreg[i];
            __syncthreads();
            }
            SIV_UDS(5);
            if (!(SIV_UL_EXP & 2)) load_halo(step + 1 < nstep ? step + 1 : step);
            SIV_UDS(6);

            // the 36 slices of this (group, u); the ring runs on into the next step's (the same ones for dz < 3)
            const int wcur = wso0 + seq * C::UNITS * C::SLICE * 4;
            const int wnxt = dz < 3 ? wcur : wcur + C::UNITS * C::SLICE * 4;
            float4 av = *reinterpret_cast<const float4*>(lds + aoff + unitoff(0));
#pragma unroll
            for (int u = 0; u < C::UNITS; ++u) {
                const float4 bv = bring[u % PD];
                if (!(SIV_UL_EXP & 1))
                bring[u % PD] = u + PD < C::UNITS ? buf_load16(wrs, wvo, wcur + (u + PD) * C::SLICE * 4)
                                                  : buf_load16(wrs, wvo, wnxt + (u + PD - C::UNITS) * C::SLICE * 4);
                const float4 a4 = av;
                if (u + 1 < C::UNITS) av = *reinterpret_cast<const float4*>(lds + aoff + unitoff(u + 1));
                __builtin_amdgcn_sched_barrier(0);
                acc[dz] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, bv.x, acc[dz], 0, 0, 0);
                acc[dz] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, bv.y, acc[dz], 0, 0, 0);
                acc[dz] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, bv.z, acc[dz], 0, 0, 0);
                acc[dz] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, bv.w, acc[dz], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            SIV_UDS(7);
        }
    }

    // store: transpose each 32x32 tile through this wave's slice of LDS, write 16-byte channel runs
    float* tile = lds + wave * C::EPI_FLOATS;
    const int cq = lane & 7;
    const int ci = ci0 + wn * 32 + 4 * cq;
#pragma unroll
    for (int a = 0; a < C::AM; ++a) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) tile[((r & 3) + 8 * (r >> 2) + 4 * lh) * C::EPI_STRIDE + li] = acc[a][r];
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int row = ps * 8 + (lane >> 3);
            const int pm = wm * 32 + row;                       // voxel of plane a
            const int gz = z0 + a, gy = y0 + pm / C::LBW, gx = x0 + pm % C::LBW;
            const float4 v = *reinterpret_cast<const float4*>(tile + row * C::EPI_STRIDE + 4 * cq);
            if (gz < Dl && gy < Hl && gx < Wl && ci < Cin)
                *reinterpret_cast<float4*>(dx + ((((long)n * Dl + gz) * Hl + gy) * Wl + gx) * Cin + ci) = v;
        }
    }
}
#undef SIV_PIN

// Packed 4x4x4 stride-2 kernel: wp[ct][cc][ud][t = (uh, uw)][ci (64)][co (8)] from w[Kout][27][Kin];
// along an axis, offset u = -1, 0, 1, 2 (index 0..3) sums the original taps {2}, {1,2}, {0,1}, {0}.
__global__ __launch_bounds__(256) void k2_up_pack_dgrad(const float* __restrict__ w, float* __restrict__ wp,
                                                        int Kin, int Kout) {
    const int nchunk = Kout / 8, ntile = (Kin + 63) / 64;
    const long total = (long)ntile * nchunk * 64 * 64 * 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7);
        long r = i >> 3;
        const int ci = (int)(r & 63); r >>= 6;
        const int t = (int)(r & 15); r >>= 4;
        const int ud = (int)(r & 3); r >>= 2;
        const int cc = (int)(r % nchunk);
        const int ct = (int)(r / nchunk);
        const int uh = t >> 2, uw = t & 3;
        const int kci = ct * 64 + ci, kco = cc * 8 + j;
        float v = 0.f;
        if (kci < Kin) {
            // index u -> taps [lo, hi]: 0 -> {2}, 1 -> {1,2}, 2 -> {0,1}, 3 -> {0}
            const int d0 = ud == 0 ? 2 : ud == 1 ? 1 : 0, d1 = ud <= 1 ? 2 : ud == 2 ? 1 : 0;
            const int h0 = uh == 0 ? 2 : uh == 1 ? 1 : 0, h1 = uh <= 1 ? 2 : uh == 2 ? 1 : 0;
            const int w0 = uw == 0 ? 2 : uw == 1 ? 1 : 0, w1 = uw <= 1 ? 2 : uw == 2 ? 1 : 0;
            for (int kd = d0; kd <= d1; ++kd)
                for (int kh = h0; kh <= h1; ++kh)
                    for (int kw = w0; kw <= w1; ++kw) v += w[((long)kco * 27 + (kd * 3 + kh) * 3 + kw) * Kin + kci];
        }
        wp[i] = v;
    }
}

// The box form: wp[ct][step][t = e_h * 3 + e_w][ci (64)][co (8)] = sum over the taps kd that collapse
// onto ud of w[co][kd][2 - e_h][2 - e_w][ci] (the box that starts e rows below 2j - 1 belongs to tap 2 - e),
// the steps in the kernel's order: ((chunk group * 4 + ud) * G + chunk in group), G = 1 << up_dgrad_group_shift.
__global__ __launch_bounds__(256) void k2_up_pack_dgrad_box(const float* __restrict__ w, float* __restrict__ wp,
                                                            int Kin, int Kout) {
    const int nchunk = Kout / 8, ntile = (Kin + 63) / 64;
    const long total = (long)ntile * nchunk * 4 * 9 * 64 * 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7);
        long r = i >> 3;
        const int ci = (int)(r & 63); r >>= 6;
        const int t = (int)(r % 9); r /= 9;
        const int step = (int)(r % (nchunk * 4));
        const int ct = (int)(r / (nchunk * 4));
        const int gsh = up_dgrad_group_shift(nchunk);
        const int ud = (step >> gsh) & 3;
        const int cc = ((step >> (2 + gsh)) << gsh) + (step & ((1 << gsh) - 1));
        const int kh = 2 - t / 3, kw = 2 - t % 3;
        const int kci = ct * 64 + ci, kco = cc * 8 + j;
        float v = 0.f;
        if (kci < Kin) {
            const int d0 = ud == 0 ? 2 : ud == 1 ? 1 : 0, d1 = ud <= 1 ? 2 : ud == 2 ? 1 : 0;
            for (int kd = d0; kd <= d1; ++kd) v += w[((long)kco * 27 + (kd * 3 + kh) * 3 + kw) * Kin + kci];
        }
        wp[i] = v;
    }
}

// The same kernel as bf16 pieces: per (ci tile, co chunk, ud, tap) two operand slices of
// [64 ci][2 lane halves][8 co]: slice 0 = (h | m), slice 1 = (h | l).
__global__ __launch_bounds__(256) void k2_up_pack_dgrad_b3(const float* __restrict__ w, __bf16* __restrict__ wp,
                                                           int Kin, int Kout) {
    const int nchunk = Kout / 8, ntile = Kin / 64;
    const long total = (long)ntile * nchunk * 64 * 2 * 64 * 2 * 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7);
        long r = i >> 3;
        const int half = (int)(r & 1); r >>= 1;
        const int ci = (int)(r & 63); r >>= 6;
        const int which = (int)(r & 1); r >>= 1;
        const int t = (int)(r & 15); r >>= 4;
        const int ud = (int)(r & 3); r >>= 2;
        const int cc = (int)(r % nchunk);
        const int ct = (int)(r / nchunk);
        const int uh = t >> 2, uw = t & 3;
        const int kci = ct * 64 + ci, kco = cc * 8 + j;
        const int d0 = ud == 0 ? 2 : ud == 1 ? 1 : 0, d1 = ud <= 1 ? 2 : ud == 2 ? 1 : 0;
        const int h0 = uh == 0 ? 2 : uh == 1 ? 1 : 0, h1 = uh <= 1 ? 2 : uh == 2 ? 1 : 0;
        const int w0 = uw == 0 ? 2 : uw == 1 ? 1 : 0, w1 = uw <= 1 ? 2 : uw == 2 ? 1 : 0;
        float v = 0.f;
        for (int kd = d0; kd <= d1; ++kd)
            for (int kh = h0; kh <= h1; ++kh)
                for (int kw = w0; kw <= w1; ++kw) v += w[((long)kco * 27 + (kd * 3 + kh) * 3 + kw) * Kin + kci];
        __bf16 h, m, l;
        split3_scalar(v, h, m, l);
        wp[i] = half == 0 ? h : (which == 0 ? m : l);
    }
}

// Input channels (the channels of dx) in tiles of 64; a last partial tile is zero padded in the packed weights
// and masked in the stores (fp32 kernels; the split-bf16 packing needs whole tiles).
bool k2_up_dgrad_supported(int Cin, int Cout) { return Cout % 8 == 0 && Cin % 16 == 0; }
size_t k2_up_dgrad_ws_bytes(int Cin, int Cout) {   // sized for the split-bf16 packing (4 KB per tap)
    return ((size_t)((Cin + 63) / 64) * (Cout / 8) * 64 + 4) * 4096;
}

// dx = part[0] + part[1] (+ part[2] + part[3]), 16 bytes per thread and pass
template <int KP>
__global__ __launch_bounds__(256) void up_dgrad_ksum_k(const float* __restrict__ part, float* __restrict__ dx, long quads) {
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < quads; q += (long)gridDim.x * 256) {
        const float4* p = reinterpret_cast<const float4*>(part) + q;
        float4 a = p[0], b = p[quads];
        a = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
        if constexpr (KP == 4) {
            const float4 c = p[2 * quads], d = p[3 * quads];
            a = make_float4(a.x + (c.x + d.x), a.y + (c.y + d.y), a.z + (c.z + d.z), a.w + (c.w + d.w));
        }
        reinterpret_cast<float4*>(dx)[q] = a;
    }
}
// Parts of the step loop (the contraction over the convolution's output channels and the taps along D) a
// launch of the box form is cut into: 4 where the bricks do not fill the 2 x 256 block slots (256 -> 128
// @20x24x20, batch 8: 160 blocks of 64 steps -> 640 of 16), 2 where a second round would be at most 60 % full
// (128 -> 64 @40x48x40: 640 blocks -> 1280 half blocks = 2.5 rounds of half the length); 1 otherwise.
#ifndef SIV_UPD_KSPLIT
#define SIV_UPD_KSPLIT 1
#endif
static int up_dgrad_parts(long nblk_total, int Cout, long dx_elems) {
    const int nchunk = Cout / 8;
    if (!SIV_UPD_KSPLIT || up_dgrad_group_shift(nchunk) != 2) return 1;         // whole groups of four chunks per part
    const long slots = 512;
    const int ngroup = nchunk / 4;
    if (nblk_total * 5 < slots * 2 && ngroup % 2 == 0) return ngroup % 4 == 0 ? 4 : 2;
    // one round that is 40..100 % full (the same layer on a merged batch of two passes): the number of parts
    // that shortens the launch by at least a fifth -- rounds of blocks x the length of a block
    if (nblk_total <= slots && ngroup % 2 == 0 && 2 * dx_elems * 4 <= (96L << 20)) {
        int best = 1;
        long best_cost = 4;                                        // cost in quarters of an unsplit block
        for (int p = 2; p <= (ngroup % 4 == 0 ? 4 : 2); p *= 2) {
            const long cost = cdiv(nblk_total * p, slots) * (4 / p);
            if (cost < best_cost && cost * 5 <= 4 * 4 && (long)p * dx_elems * 4 <= (96L << 20)) best = p, best_cost = cost;
        }
        if (best > 1) return best;
    }
    const long rounds = cdiv(nblk_total, slots);
    if (nblk_total > slots && rounds <= 3 && nblk_total * 10 <= (rounds * 10 - 4) * slots
        && cdiv(2 * nblk_total, slots) < 2 * rounds && ngroup % 2 == 0 && 2 * dx_elems * 4 <= (96L << 20))
        return 2;
    return 1;
}

template <int LBD, int LBH, int LBW, bool SPLIT, bool BOX = false>
static int up_dgrad_cfg(const float* dy, const void* wp, float* dx, int N, int D, int H, int W, int Cin,
                        int Cout, hipStream_t s, void* kws = nullptr, size_t kws_bytes = 0) {
    using C = UdCfg<LBD, LBH, LBW>;
    const int nbd = (int)cdiv(D / 2, LBD), nbh = (int)cdiv(H / 2, LBH), nbw = (int)cdiv(W / 2, LBW);
    const long nblk = (long)N * nbd * nbh * nbw;
    if (nblk > 0x7fffffffL || (long)(2 * LBD + 2) * H * W * Cout > 0x7fffffffL) {
        set_error("conv3d_k3_dgrad(upsampled): volume too large for 32-bit brick offsets");
        return SIV_EINVAL;
    }
    // more than 64 KB of dynamic LDS is an opt-in
    if (hipFuncSetAttribute((const void*)k2_up_dgrad<LBD, LBH, LBW, SPLIT, BOX>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)C::LDS_BYTES) != hipSuccess) {
        set_error("conv3d_k3_dgrad(upsampled): cannot reserve %zu bytes of LDS", C::LDS_BYTES);
        return SIV_EHIP;
    }
    const long dx_elems = (long)N * (D / 2) * (H / 2) * (W / 2) * Cin;
    int parts = BOX && Cin % 4 == 0 ? up_dgrad_parts(nblk * ((Cin + 63) / 64), Cout, dx_elems) : 1;
    if (parts > 1 && (kws == nullptr || kws_bytes < (size_t)parts * dx_elems * sizeof(float) || (((uintptr_t)kws) & 15) != 0))
        parts = 1;
    k2_up_dgrad<LBD, LBH, LBW, SPLIT, BOX><<<dim3((unsigned)nblk, (unsigned)((Cin + 63) / 64), (unsigned)parts), 256, C::LDS_BYTES, s>>>(
        dy, wp, parts > 1 ? (float*)kws : dx, N, D, H, W, Cin, Cout, nbd, nbh, nbw);
    SIV_CHECK_LAUNCH("conv3d_k3_dgrad(upsampled)");
    if (parts > 1) {
        const long quads = dx_elems / 4;
        long g = cdiv(quads, 256L * 4);
        if (g > 2048) g = 2048;
        if (parts == 4) up_dgrad_ksum_k<4><<<(unsigned)g, 256, 0, s>>>((const float*)kws, dx, quads);
        else up_dgrad_ksum_k<2><<<(unsigned)g, 256, 0, s>>>((const float*)kws, dx, quads);
        SIV_CHECK_LAUNCH("conv3d_k3_dgrad(upsampled, sum of the parts)");
    }
    return SIV_OK;
}

// dy: [N][D][H][W][Cout] -> dx: [N][D/2][H/2][W/2][Cin]; w: the convolution's weights [Cout][27][Cin].
// split: products from bf16 pieces; force_box: the 9-tap box form whatever the volume (fp32 only).
int k2_up_dgrad_launch(const float* dy, const float* w, float* dx, int N, int D, int H, int W, int Cin,
                       int Cout, void* ws, size_t ws_bytes, hipStream_t s, bool split, bool force_box) {
    // the box form on request and from 32 K output voxels upwards (1.35x faster at the benchmark sizes);
    // tiny volumes keep the 16-tap form
    if (Cin % 64 != 0) split = false;
    // (also where the box form's step loop can be cut into parts: a small volume is a handful of bricks)
    const bool w10 = up_wide10(W);
    const long nbricks = (long)N * cdiv(D / 2, w10 ? 2 : 4) * cdiv(H / 2, w10 ? 12 : 8) * cdiv(W / 2, w10 ? 10 : 8);
    const long dx_elems = (long)N * (D / 2) * (H / 2) * (W / 2) * Cin;
    const int box_parts = split || Cin % 4 != 0 ? 1 : up_dgrad_parts(nbricks * ((Cin + 63) / 64), Cout, dx_elems);
    const bool box = !split && (force_box || (long)N * D * H * W >= (1L << 15) || box_parts > 1);
    SIV_CHECK_WS(k2_up_dgrad_ws_bytes(Cin, Cout), ws_bytes, "conv3d_k3_dgrad(upsampled)");
    bool need_pack;
    void* wpk = pack_slot(w, split ? PF_UPD_B3 : box ? PF_UPD_BOX : PF_UPD, 0, 1, Cin, Cout,
                          k2_up_dgrad_ws_bytes(Cin, Cout), ws, &need_pack);
    if (need_pack) {
        if (split) {
            const long total = (long)(Cin / 64) * (Cout / 8) * 64 * 2 * 64 * 2 * 8;
            long g = cdiv(total, 256);
            if (g > 4096) g = 4096;
            k2_up_pack_dgrad_b3<<<(unsigned)g, 256, 0, s>>>(w, (__bf16*)wpk, Cin, Cout);
        } else if (box) {
            const long total = (long)((Cin + 63) / 64) * (Cout / 8) * 36 * 512;
            long g = cdiv(total, 256);
            if (g > 4096) g = 4096;
            k2_up_pack_dgrad_box<<<(unsigned)g, 256, 0, s>>>(w, (float*)wpk, Cin, Cout);
        } else {
            const long total = (long)((Cin + 63) / 64) * (Cout / 8) * 64 * 512;
            long g = cdiv(total, 256);
            if (g > 4096) g = 4096;
            k2_up_pack_dgrad<<<(unsigned)g, 256, 0, s>>>(w, (float*)wpk, Cin, Cout);
        }
        SIV_CHECK_PACK_LAUNCH("conv3d_k3_dgrad(pack upsampled weights)", wpk);
    }
    if (split) {
        if (up_wide10(W)) return up_dgrad_cfg<2, 12, 10, true>(dy, wpk, dx, N, D, H, W, Cin, Cout, s);
        return up_dgrad_cfg<4, 8, 8, true>(dy, wpk, dx, N, D, H, W, Cin, Cout, s);
    }
    if (box) {
        void* kws = ws_tail(ws, ws_bytes, k2_up_dgrad_ws_bytes(Cin, Cout));
        const size_t kws_bytes = ws_tail_bytes(ws_bytes, k2_up_dgrad_ws_bytes(Cin, Cout));
        if (up_wide10(W)) return up_dgrad_cfg<2, 12, 10, false, true>(dy, wpk, dx, N, D, H, W, Cin, Cout, s, kws, kws_bytes);
        // whole cache lines per staged voxel where the channel groups are whole (32) and the lane offsets,
        // shifted by up to nine planes, stay clear of the marker for voxels outside the sample
        if (SIV_UPD_LINE && box_parts == 1 && Cout % 32 == 0 && (long)(D + 10) * H * W * Cout * 4 < (1L << 30)) {
            using C = UlCfg;
            const int nbd = (int)cdiv(D / 2, C::LBD), nbh = (int)cdiv(H / 2, C::LBH), nbw = (int)cdiv(W / 2, C::LBW);
            const long nblk = (long)N * nbd * nbh * nbw;
            if (nblk > 0x7fffffffL) {
                set_error("conv3d_k3_dgrad(upsampled): too many bricks");
                return SIV_EINVAL;
            }
            if (hipFuncSetAttribute((const void*)k2_up_dgrad_line, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)C::LDS_BYTES) != hipSuccess) {
                set_error("conv3d_k3_dgrad(upsampled): cannot reserve %zu bytes of LDS", C::LDS_BYTES);
                return SIV_EHIP;
            }
            k2_up_dgrad_line<<<dim3((unsigned)nblk, (unsigned)((Cin + 63) / 64)), 256, C::LDS_BYTES, s>>>(
                dy, (const float*)wpk, dx, N, D, H, W, Cin, Cout, nbd, nbh, nbw);
            SIV_CHECK_LAUNCH("conv3d_k3_dgrad(upsampled, whole lines)");
            return SIV_OK;
        }
        return up_dgrad_cfg<4, 8, 8, false, true>(dy, wpk, dx, N, D, H, W, Cin, Cout, s, kws, kws_bytes);
    }
    if (up_wide10(W)) return up_dgrad_cfg<2, 12, 10, false>(dy, wpk, dx, N, D, H, W, Cin, Cout, s);
    return up_dgrad_cfg<4, 8, 8, false>(dy, wpk, dx, N, D, H, W, Cin, Cout, s);
}

}  // namespace siv
