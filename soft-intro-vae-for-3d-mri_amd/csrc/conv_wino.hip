// 3x3x3 convolution, forward / input gradient, with the Winograd F(2,3) minimal-filtering
// identity applied along W and the fp32 matrix cores doing the channel contraction.
//
// For two adjacent outputs y0, y1 of a row and the four inputs d0..d3 they depend on,
//   y0 = m0 + m1 + m2,  y1 = m1 - m2 - m3,   with
//   m0 = (d0 - d2) g0,  m1 = (d1 + d2)(g0 + g1 + g2)/2,  m2 = (d2 - d1)(g0 - g1 + g2)/2,  m3 = (d1 - d3) g2
// so a 3-tap row costs 4 multiplications per output pair instead of 6: the 27 taps per output of
// the direct kernel become 9 (kd,kh) x 4 positions per PAIR = 18 per output.  Everything stays
// fp32; the only change in rounding is the additions of the two transforms (the /2 is exact).
//
//   M rows  = output pairs of a 4x8x8-voxel brick (128 pairs), N = 64 output channels
//   K       = 9 (kd,kh) x Cin per Winograd position; 4 positions = 4 independent accumulators
//
// The input transform is applied while the halo is staged into LDS (V[pos][halo row][pair][ci]),
// the weight transform when the weights are packed, the output transform on the accumulators.
#include "siv_common.h"

// SIV_WEXP: timing-only ablation mask used by tools/ubench (never set in the library build):
// 1 = no weight-fragment loads, 2 = no LDS fragment reads, 4 = no halo loads, 8 = no output stores,
// 16 = no input transform / LDS halo writes.
#ifndef SIV_WPD
#define SIV_WPD 6      // weight-fragment prefetch distance in taps (divides 36)
#endif
#ifndef SIV_WEXP
#define SIV_WEXP 0
#endif

namespace siv {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int UPS>
struct WinoCfg {
    static constexpr int BD = 4, BH = 8, BW = 8;
    static constexpr int CK = 8, CKP = 12;
    static constexpr int PW = BW / 2;                     // output pairs per brick row
    static constexpr int M = BD * BH * PW;                // 128 rows
    static constexpr int WM = 2, WN = 2, NW = 4, NTHR = 256;
    static constexpr int NT = 64;
    static constexpr int AM = (M / 32) / WM;              // 2 row tiles per wave, 1 column tile
    static constexpr int HD = BD + 2, HH = BH + 2;
    static constexpr int HP = HD * HH * PW;               // (halo row, pair) items: 240
    static constexpr int PLANE = HP * CKP;                // floats of one Winograd position
    static constexpr int V_FLOATS = 4 * PLANE;
    static constexpr int TAPS = 36;                       // (kd, kh, pos)
    static constexpr int SLICE = NT * 8;
    static constexpr int EPI_STRIDE = 40, EPI_FLOATS = 32 * EPI_STRIDE;
    static constexpr int LDS_FLOATS = V_FLOATS > NW * EPI_FLOATS ? V_FLOATS : NW * EPI_FLOATS;
    static constexpr size_t LDS_BYTES = (size_t)LDS_FLOATS * sizeof(float);
};

__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

template <int UPS>
__global__ __launch_bounds__(256, 2) void k3_fwd_wino(const float* __restrict__ x,
                                                      const float* __restrict__ wp,
                                                      const float* __restrict__ bias,
                                                      float* __restrict__ y, int N, int D, int H, int W,
                                                      int Cin, int Cout, int nbd, int nbh, int nbw,
                                                      float* __restrict__ stat_part) {
    using C = WinoCfg<UPS>;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;

    // each XCD walks one contiguous eighth of the brick list (bricks sharing halo voxels meet in one L2)
    int b = blockIdx.x;
    if ((gridDim.x & 7) == 0) b = (b & 7) * (gridDim.x >> 3) + (b >> 3);
    const int bx = b % nbw; b /= nbw;
    const int by = b % nbh; b /= nbh;
    const int bz = b % nbd;
    const int n = b / nbd;
    const int z0 = bz * C::BD, y0 = by * C::BH, x0 = bx * C::BW;
    const int co0 = blockIdx.y * C::NT;
    const int Di = D / UPS, Hi = H / UPS, Wi = W / UPS;
    const int nchunk = (Cin + C::CK - 1) / C::CK;

    f32x16 acc[4][C::AM];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int a = 0; a < C::AM; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][a][r] = 0.f;

    // LDS offset of this lane's A row (pair) for tap (0,0), position 0
    int aoff[C::AM];
#pragma unroll
    for (int a = 0; a < C::AM; ++a) {
        const int m = (wm * C::AM + a) * 32 + li;
        const int dz = m / (C::BH * C::PW), dy = (m / C::PW) % C::BH, px = m % C::PW;
        aoff[a] = ((dz * C::HH + dy) * C::PW + px) * C::CKP + 4 * lh;
    }
    // this lane's B fragment inside a packed [64 co][8 ci] slice
    const float* wlane = wp + (long)blockIdx.y * nchunk * C::TAPS * C::SLICE + (wn * 32 + li) * 8 + 4 * lh;
    // weight fragments are fetched PD taps ahead of their MFMAs (a ring of PD registers quads; PD
    // divides the 36 taps so that the ring index is a compile-time constant)
    constexpr int PD = SIV_WPD;
    static_assert(C::TAPS % PD == 0, "ring index must not depend on the chunk");
    float4 bring[PD];

    // halo staging: thread = (halo row, half of the row, channel half).  It loads the six raw columns
    // its two output pairs depend on (pairs 2h, 2h+1 use columns 4h..4h+3 and 4h+2..4h+5) and writes
    // their 2 x 4 transformed values: 6 loads for 8 LDS writes, 240 of the 256 threads busy.
    static_assert(C::PW == 4 && C::HD * C::HH * 4 <= C::NTHR, "staging map assumes 4 pairs per row");
    const int s_q = tid & 1, s_half = (tid >> 1) & 1, s_row = tid >> 2;
    const bool s_on = s_row < C::HD * C::HH;
    float4 hreg[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) hreg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    // chunk-invariant part of the six addresses (offsets from x of this thread's channel quad in
    // chunk 0) and their validity, computed once: per chunk only an add and a select remain
    long hoff[6];
    unsigned hmask = 0;
    {
        const int hy = s_row % C::HH, hz = s_row / C::HH;
        const int gz = z0 + hz - 1, gy = y0 + hy - 1;
        const bool rowok = s_on && gz >= 0 && gz < D && gy >= 0 && gy < H;
        const long rowbase = (((long)n * Di + (rowok ? gz : 0) / UPS) * Hi + (rowok ? gy : 0) / UPS) * Wi;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int gx = x0 + 4 * s_half - 1 + j;
            const bool ok = rowok && gx >= 0 && gx < W;
            hoff[j] = ok ? (rowbase + gx / UPS) * Cin + 4 * s_q : 0;
            if (SIV_WEXP & 32) hoff[j] &= 0xFFFF;     // timing only: every halo comes from one 256 KB window
            hmask |= (unsigned)ok << j;
        }
    }
    // The loads are unconditional (out-of-volume columns read voxel 0 and are zeroed when they are
    // consumed): straight-line code lets the compiler wait for exactly these six loads at the top of
    // the next chunk and leave the younger weight-fragment loads in flight.
    auto load_halo = [&](int c0) {
#pragma unroll
        for (int j = 0; j < 6; ++j)
            if (!(SIV_WEXP & 4)) hreg[j] = *reinterpret_cast<const float4*>(x + hoff[j] + c0);
    };
    load_halo(0);
#pragma unroll
    for (int i = 0; i < PD; ++i) bring[i] = *reinterpret_cast<const float4*>(wlane + i * C::SLICE);
    // Nothing can start before the first halo has arrived anyway.  Pinning the loaded values here (the
    // empty asm statements make the compiler wait for each of them) costs nothing and leaves the loop
    // with one known state on entry: inside it the compiler then counts, per wait, the loads that may
    // stay in flight (loads return in order), instead of taking the worse of two orders and draining
    // the weight-fragment pipeline at the top of every chunk.
#define SIV_PIN(V) asm volatile("" : "+v"((V).x), "+v"((V).y), "+v"((V).z), "+v"((V).w))
#pragma unroll
    for (int j = 0; j < 6; ++j) SIV_PIN(hreg[j]);
#pragma unroll
    for (int i = 0; i < PD; ++i) SIV_PIN(bring[i]);
#undef SIV_PIN
    for (int cc = 0; cc < nchunk; ++cc) {
        __syncthreads();  // previous chunk's MFMAs are done with V
#pragma unroll
        for (int j = 0; j < 6; ++j)
            if (!((hmask >> j) & 1)) hreg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s_on && (!(SIV_WEXP & 16) || hreg[0].x == 123.f)) {
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                float* dst = lds + (s_row * C::PW + 2 * s_half + pp) * C::CKP + 4 * s_q;
                *reinterpret_cast<float4*>(dst) = f4sub(hreg[2 * pp], hreg[2 * pp + 2]);
                *reinterpret_cast<float4*>(dst + C::PLANE) = f4add(hreg[2 * pp + 1], hreg[2 * pp + 2]);
                *reinterpret_cast<float4*>(dst + 2 * C::PLANE) = f4sub(hreg[2 * pp + 2], hreg[2 * pp + 1]);
                *reinterpret_cast<float4*>(dst + 3 * C::PLANE) = f4sub(hreg[2 * pp + 1], hreg[2 * pp + 3]);
            }
        }
        __syncthreads();
        // (the last chunk re-requests its own halo rather than branching around the loads: with one
        // instruction stream the compiler can count exactly how many younger loads may stay in flight
        // at each wait, instead of assuming the shorter of two paths)
        load_halo((cc + 1 < nchunk ? cc + 1 : cc) * C::CK);
        const float* wc = wlane + (long)cc * C::TAPS * C::SLICE;
        float4 anext[C::AM];
#pragma unroll
        for (int a = 0; a < C::AM; ++a) anext[a] = *reinterpret_cast<const float4*>(lds + aoff[a]);
#pragma unroll
        for (int t = 0; t < C::TAPS; ++t) {
            const int pos = t & 3;
            float4 av[C::AM];
            const float4 bv = bring[t % PD];
#pragma unroll
            for (int a = 0; a < C::AM; ++a) av[a] = anext[a];
            if (!(SIV_WEXP & 1)) bring[t % PD] = *reinterpret_cast<const float4*>(wc + (t + PD) * C::SLICE);
            if (t + 1 < C::TAPS) {
                const int t1 = t + 1;
                const int tapoff = (t1 & 3) * C::PLANE + (((t1 / 12) * C::HH + (t1 / 4) % 3) * C::PW) * C::CKP;
#pragma unroll
                for (int a = 0; a < C::AM; ++a) {
                    if (SIV_WEXP & 2)
                        asm volatile("" : "+v"(anext[a].x), "+v"(anext[a].y), "+v"(anext[a].z), "+v"(anext[a].w));
                    else
                        anext[a] = *reinterpret_cast<const float4*>(lds + aoff[a] + tapoff);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#define SIV_WSTEP(F)                                                                        \
    _Pragma("unroll") for (int a = 0; a < C::AM; ++a)                                       \
        acc[pos][a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a].F, bv.F, acc[pos][a], 0, 0, 0);
            SIV_WSTEP(x)
            SIV_WSTEP(y)
            SIV_WSTEP(z)
            SIV_WSTEP(w)
#undef SIV_WSTEP
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // output transform + store: per row tile and output of the pair, transpose the 32x32 result
    // through this wave's slice of LDS and write 16-byte channel runs.  When stat_part is given the
    // block also emits, per output channel, the sum and the sum of squares of its (bias-free) outputs:
    // the BatchNorm that follows gets its batch statistics without re-reading the tensor.
    float* tile = lds + wave * C::EPI_FLOATS;
    const int cq = lane & 7;
    const int co = co0 + wn * 32 + 4 * cq;
    float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias && co < Cout) bq = *reinterpret_cast<const float4*>(bias + co);
    const int orig_b = blockIdx.x;
    const bool full = z0 + C::BD <= D && y0 + C::BH <= H && x0 + C::BW <= W;   // no padded voxels
    float s1 = 0.f, s2 = 0.f;     // this lane's column (channel co0 + wn*32 + li), its 16 rows per tile
#pragma unroll
    for (int a = 0; a < C::AM; ++a) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = half == 0 ? (acc[0][a][r] + acc[1][a][r]) + acc[2][a][r]
                                          : (acc[1][a][r] - acc[2][a][r]) - acc[3][a][r];
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                tile[row * C::EPI_STRIDE + li] = v;
                if (stat_part) {
                    bool in = full;
                    if (!full) {
                        const int m = (wm * C::AM + a) * 32 + row;
                        const int gz = z0 + m / (C::BH * C::PW), gy = y0 + (m / C::PW) % C::BH;
                        const int gx = x0 + 2 * (m % C::PW) + half;
                        in = gz < D && gy < H && gx < W;
                    }
                    if (in) {
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
                const int gz = z0 + m / (C::BH * C::PW), gy = y0 + (m / C::PW) % C::BH;
                const int gx = x0 + 2 * (m % C::PW) + half;
                float4 v = *reinterpret_cast<const float4*>(tile + row * C::EPI_STRIDE + 4 * cq);
                v.x += bq.x; v.y += bq.y; v.z += bq.z; v.w += bq.w;
                if (gz < D && gy < H && gx < W && co < Cout && (!(SIV_WEXP & 8) || v.x == 123.f))
                    *reinterpret_cast<float4*>(y + ((((long)n * D + gz) * H + gy) * W + gx) * Cout + co) = v;
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

// Packed, transformed weights: wp[ct][cc][(kd,kh,pos)][co (64)][ci (8)], zero padded.
//   flip == 0: g[kw] = w[kco][kd][kh][kw][kci]              (w is [Kout][27][Kin])
//   flip == 1: g[kw] = w[kci][2-kd][2-kh][2-kw][kco]        (w is [Kin][27][Kout]; input gradient)
__global__ __launch_bounds__(256) void k3_pack_weights_wino(const float* __restrict__ w,
                                                            float* __restrict__ wp, int Kin, int Kout,
                                                            int flip) {
    const int nchunk = (Kin + 7) / 8, ntile = (Kout + 63) / 64;
    const long total = (long)ntile * nchunk * 36 * 64 * 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7);
        long r = i >> 3;
        const int co = (int)(r & 63); r >>= 6;
        const int t = (int)(r % 36); r /= 36;
        const int cc = (int)(r % nchunk);
        const int ct = (int)(r / nchunk);
        const int pos = t & 3, kh = (t >> 2) % 3, kd = t / 12;
        const int kco = ct * 64 + co, kci = cc * 8 + j;
        float v = 0.f;
        if (kco < Kout && kci < Kin) {
            float g[3];
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int tap = (kd * 3 + kh) * 3 + kw;
                g[kw] = flip ? w[((long)kci * 27 + (26 - tap)) * Kout + kco] : w[((long)kco * 27 + tap) * Kin + kci];
            }
            v = pos == 0 ? g[0] : pos == 1 ? 0.5f * ((g[0] + g[1]) + g[2]) : pos == 2 ? 0.5f * ((g[0] - g[1]) + g[2]) : g[2];
        }
        wp[i] = v;
    }
}

bool k3_fwd_wino_supported(int Cin, int Cout) { return Cin % 8 == 0 && Cout % 64 == 0; }
size_t k3_fwd_wino_ws_bytes(int Cin, int Cout) {
    return ((size_t)((Cout + 63) / 64) * ((Cin + 7) / 8) * 36 * 512 + SIV_WPD * 512) * sizeof(float);
}

// stat_part (optional): receives [bricks][2][Cout] per-brick sums / sums of squares of the bias-free
// output; *stat_blocks receives the number of bricks.  It must not overlap the first
// k3_fwd_wino_ws_bytes() bytes of ws.
long k3_fwd_wino_bricks(int N, int D, int H, int W) {
    return (long)N * ((D + 3) / 4) * ((H + 7) / 8) * ((W + 7) / 8);
}
int k3_fwd_wino_launch(const float* x, const float* w, const float* bias, float* y, int N, int D, int H,
                       int W, int Cin, int Cout, int ups, int flip, void* ws, size_t ws_bytes,
                       hipStream_t s, float* stat_part) {
    SIV_CHECK_WS(k3_fwd_wino_ws_bytes(Cin, Cout), ws_bytes, "conv3d_k3(winograd)");
    bool need_pack;
    float* wp = (float*)pack_slot(w, PF_WINO, 0, flip, Cin, Cout, k3_fwd_wino_ws_bytes(Cin, Cout), ws, &need_pack);
    const long total = (long)((Cout + 63) / 64) * ((Cin + 7) / 8) * 36 * 512;
    long g = cdiv(total, 256);
    if (g > 4096) g = 4096;
    if (need_pack) {
        k3_pack_weights_wino<<<(unsigned)g, 256, 0, s>>>(w, wp, Cin, Cout, flip);
        SIV_CHECK_PACK_LAUNCH("conv3d_k3(pack winograd weights)", wp);
    }
    const int nbd = (D + 3) / 4, nbh = (H + 7) / 8, nbw = (W + 7) / 8;
    const long nblk = (long)N * nbd * nbh * nbw;
    if (nblk > 0x7fffffffL) {
        set_error("conv3d_k3(winograd): too many bricks");
        return SIV_EINVAL;
    }
    const dim3 grid((unsigned)nblk, (unsigned)(Cout / 64));
    if (ups == 1)
        k3_fwd_wino<1><<<grid, 256, WinoCfg<1>::LDS_BYTES, s>>>(x, wp, bias, y, N, D, H, W, Cin, Cout, nbd, nbh, nbw, stat_part);
    else
        k3_fwd_wino<2><<<grid, 256, WinoCfg<2>::LDS_BYTES, s>>>(x, wp, bias, y, N, D, H, W, Cin, Cout, nbd, nbh, nbw,
                                                               stat_part);
    SIV_CHECK_LAUNCH("conv3d_k3(winograd)");
    return SIV_OK;
}

}  // namespace siv
