// 3x3x3 convolution, forward / input gradient: Winograd F(2,3) along W as in conv_wino.hip, with every
// fp32 product formed on the bf16 matrix cores from three bf16 pieces per operand.
//
//   x = h + m + l  (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m): 24+ significant bits)
//   a * b ~= ah*bh + ah*bm + am*bh + ah*bl + al*bh + am*bm      (dropped terms < 2^-24 |a b|)
//
// bf16 x bf16 products are exact in fp32 and the matrix core accumulates in fp32, so the result
// carries fp32-grade error (measured 2.4e-7 of sum|terms| against 2.3e-7 for an fmaf chain,
// tools/ubench/mfma_bf16_split.hip) -- and v_mfma_f32_32x32x16_bf16 retires 16 k-values in 32 cycles
// where v_mfma_f32_32x32x2_f32 retires 2 in 64.  One instruction takes two of the six piece products:
// its K = 16 is 8 input channels from lane half 0 and 8 from lane half 1, and the two halves are
// given different piece pairs:
//     instruction 1:  ah*bh | ah*bm      instruction 2:  am*bh | ah*bl      instruction 3:  al*bh | am*bm
// so an 8-channel chunk of a tap costs 3 x 32 cycles per 32x32 tile instead of 4 x 64.
//
// LDS holds the transformed input as bf16 pieces, V[pos][halo row][pair][h(8ci) | m(8ci) | l(8ci)] --
// 48 bytes per row, the footprint of the fp32 kernel; the packed weights hold, per tap, the two B
// operands (bh | bm) and (bh | bl).  Tiling, staging map, epilogue and statistics are those of
// k3_fwd_wino<1>.  Inf inputs turn into NaN (inf - inf in the split), finite data are unaffected.
#include "siv_common.h"
#include "split_bf16.h"

namespace siv {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct B3Cfg {
    static constexpr int BD = 4, BH = 8, BW = 8;
    static constexpr int CK = 8, CKP = 12;                // 12 floats = 48 bytes = 3 pieces x 8 bf16
    static constexpr int PW = BW / 2;
    static constexpr int M = BD * BH * PW;                // 128 rows (output pairs)
    static constexpr int NW = 4, NTHR = 256, NT = 64, AM = 2;
    static constexpr int HD = BD + 2, HH = BH + 2;
    static constexpr int HP = HD * HH * PW;
    static constexpr int PLANE = HP * CKP;
    static constexpr int V_FLOATS = 4 * PLANE;
    static constexpr int TAPS = 36;
    static constexpr int TAP_Q = 256;                     // uint4 per tap: two operand slices of [64 co][2][8 ci]
#ifndef SIV_B3PD
#define SIV_B3PD 3
#endif
    static constexpr int PD = SIV_B3PD;                   // weight operands fetched PD taps ahead
    static constexpr int EPI_STRIDE = 40, EPI_FLOATS = 32 * EPI_STRIDE;
    static constexpr int LDS_FLOATS = V_FLOATS > NW * EPI_FLOATS ? V_FLOATS : NW * EPI_FLOATS;
    static constexpr size_t LDS_BYTES = (size_t)LDS_FLOATS * sizeof(float);
};

__device__ __forceinline__ float4 q4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 q4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

__global__ __launch_bounds__(256, 2) void k3_fwd_wino_b3(const float* __restrict__ x,
                                                         const uint4* __restrict__ wp,
                                                         const float* __restrict__ bias,
                                                         float* __restrict__ y, int N, int D, int H, int W,
                                                         int Cin, int Cout, int nbd, int nbh, int nbw,
                                                         float* __restrict__ stat_part) {
    using C = B3Cfg;
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
    const int z0 = bz * C::BD, y0 = by * C::BH, x0 = bx * C::BW;
    const int co0 = blockIdx.y * C::NT;
    const int nchunk = Cin / C::CK;

    f32x16 acc[4][C::AM];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int a = 0; a < C::AM; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][a][r] = 0.f;

    // LDS offsets (floats) of this lane's three A operands for tap (0,0), position 0: the lane halves
    // read different pieces of the row (h|h, m|h, l|m)
    int aoff1[C::AM], aoff2[C::AM], aoff3[C::AM];
#pragma unroll
    for (int a = 0; a < C::AM; ++a) {
        const int m = (wm * C::AM + a) * 32 + li;
        const int dz = m / (C::BH * C::PW), dy = (m / C::PW) % C::BH, px = m % C::PW;
        const int row = ((dz * C::HH + dy) * C::PW + px) * C::CKP;
        aoff1[a] = row;
        aoff2[a] = row + (lh ? 0 : 4);
        aoff3[a] = row + (lh ? 4 : 8);
    }
    const uint4* wlane = wp + (long)blockIdx.y * nchunk * C::TAPS * C::TAP_Q + (wn * 32 + li) * 2 + lh;
    constexpr int PD = C::PD;
    static_assert(C::TAPS % PD == 0, "ring index must not depend on the chunk");
    uint4 bxr[PD], byr[PD];

    static_assert(C::PW == 4 && C::HD * C::HH * 4 <= C::NTHR, "staging map assumes 4 pairs per row");
    const int s_q = tid & 1, s_half = (tid >> 1) & 1, s_row = tid >> 2;
    const bool s_on = s_row < C::HD * C::HH;
    float4 hreg[6];
    long hoff[6];
    unsigned hmask = 0;
    {
        const int hy = s_row % C::HH, hz = s_row / C::HH;
        const int gz = z0 + hz - 1, gy = y0 + hy - 1;
        const bool rowok = s_on && gz >= 0 && gz < D && gy >= 0 && gy < H;
        const long rowbase = (((long)n * D + (rowok ? gz : 0)) * H + (rowok ? gy : 0)) * W;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int gx = x0 + 4 * s_half - 1 + j;
            const bool ok = rowok && gx >= 0 && gx < W;
            hoff[j] = ok ? (rowbase + gx) * Cin + 4 * s_q : 0;
            hmask |= (unsigned)ok << j;
        }
    }
    auto load_halo = [&](int c0) {
#pragma unroll
        for (int j = 0; j < 6; ++j) hreg[j] = *reinterpret_cast<const float4*>(x + hoff[j] + c0);
    };
    load_halo(0);
#pragma unroll
    for (int i = 0; i < PD; ++i) {
        bxr[i] = wlane[i * C::TAP_Q];
        byr[i] = wlane[i * C::TAP_Q + 128];
    }
#define SIV_PIN(V) asm volatile("" : "+v"((V).x), "+v"((V).y), "+v"((V).z), "+v"((V).w))
#pragma unroll
    for (int j = 0; j < 6; ++j) SIV_PIN(hreg[j]);
#pragma unroll
    for (int i = 0; i < PD; ++i) { SIV_PIN(bxr[i]); SIV_PIN(byr[i]); }
#undef SIV_PIN

    for (int cc = 0; cc < nchunk; ++cc) {
        __syncthreads();  // previous chunk's MFMAs are done with V
#pragma unroll
        for (int j = 0; j < 6; ++j)
            if (!((hmask >> j) & 1)) hreg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s_on) {
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                float* dst = lds + (s_row * C::PW + 2 * s_half + pp) * C::CKP + 2 * s_q;
#pragma unroll
                for (int pos = 0; pos < 4; ++pos) {
                    const float4 t = pos == 0 ? q4sub(hreg[2 * pp], hreg[2 * pp + 2])
                                   : pos == 1 ? q4add(hreg[2 * pp + 1], hreg[2 * pp + 2])
                                   : pos == 2 ? q4sub(hreg[2 * pp + 2], hreg[2 * pp + 1])
                                              : q4sub(hreg[2 * pp + 1], hreg[2 * pp + 3]);
                    uint2 ph, pm, pl;
                    split3(t, ph, pm, pl);
                    float* d = dst + pos * C::PLANE;
                    *reinterpret_cast<uint2*>(d) = ph;
                    *reinterpret_cast<uint2*>(d + 4) = pm;
                    *reinterpret_cast<uint2*>(d + 8) = pl;
                }
            }
        }
        __syncthreads();
        load_halo((cc + 1 < nchunk ? cc + 1 : cc) * C::CK);
        const uint4* wc = wlane + (long)cc * C::TAPS * C::TAP_Q;
        // A operands are read one (tap, row tile) unit ahead of their MFMAs
        bf16x8 n1, n2, n3;
        n1 = *reinterpret_cast<const bf16x8*>(lds + aoff1[0]);
        n2 = *reinterpret_cast<const bf16x8*>(lds + aoff2[0]);
        n3 = *reinterpret_cast<const bf16x8*>(lds + aoff3[0]);
#pragma unroll
        for (int u = 0; u < C::TAPS * C::AM; ++u) {
            const int t = u / C::AM, a = u % C::AM;
            const int pos = t & 3;
            const bf16x8 a1 = n1, a2 = n2, a3 = n3;
            const bf16x8 bx8 = __builtin_bit_cast(bf16x8, bxr[t % PD]);
            const bf16x8 by8 = __builtin_bit_cast(bf16x8, byr[t % PD]);
            if (u + 1 < C::TAPS * C::AM) {
                const int t1 = (u + 1) / C::AM, a1i = (u + 1) % C::AM;
                const int tapoff = (t1 & 3) * C::PLANE + (((t1 / 12) * C::HH + (t1 / 4) % 3) * C::PW) * C::CKP;
                n1 = *reinterpret_cast<const bf16x8*>(lds + aoff1[a1i] + tapoff);
                n2 = *reinterpret_cast<const bf16x8*>(lds + aoff2[a1i] + tapoff);
                n3 = *reinterpret_cast<const bf16x8*>(lds + aoff3[a1i] + tapoff);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[pos][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, bx8, acc[pos][a], 0, 0, 0);   // al*bh | am*bm
            acc[pos][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, by8, acc[pos][a], 0, 0, 0);   // am*bh | ah*bl
            acc[pos][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bx8, acc[pos][a], 0, 0, 0);   // ah*bh | ah*bm
            __builtin_amdgcn_sched_barrier(0);
            if (a == C::AM - 1) {      // this tap's weight operands are consumed: fetch tap t + PD into their slot
                bxr[t % PD] = wc[(t + PD) * C::TAP_Q];
                byr[t % PD] = wc[(t + PD) * C::TAP_Q + 128];
            }
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
                if (gz < D && gy < H && gx < W && co < Cout)
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

// Packed weights: per (co tile, ci chunk, tap (kd,kh,pos)) two operand slices of [64 co][2 lane halves][8 ci]
// bf16: slice 0 = (h | m), slice 1 = (h | l) of the Winograd-transformed weight.
//   flip == 0: g[kw] = w[kco][kd][kh][kw][kci]              (w is [Kout][27][Kin])
//   flip == 1: g[kw] = w[kci][2-kd][2-kh][2-kw][kco]        (w is [Kin][27][Kout]; input gradient)
__global__ __launch_bounds__(256) void k3_pack_weights_wino_b3(const float* __restrict__ w,
                                                               __bf16* __restrict__ wp, int Kin, int Kout,
                                                               int flip) {
    const int nchunk = Kin / 8, ntile = Kout / 64;
    const long total = (long)ntile * nchunk * 36 * 2 * 64 * 2 * 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7);
        long r = i >> 3;
        const int half = (int)(r & 1); r >>= 1;
        const int co = (int)(r & 63); r >>= 6;
        const int which = (int)(r & 1); r >>= 1;
        const int t = (int)(r % 36); r /= 36;
        const int cc = (int)(r % nchunk);
        const int ct = (int)(r / nchunk);
        const int pos = t & 3, kh = (t >> 2) % 3, kd = t / 12;
        const int kco = ct * 64 + co, kci = cc * 8 + j;
        float g[3];
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int tap = (kd * 3 + kh) * 3 + kw;
            g[kw] = flip ? w[((long)kci * 27 + (26 - tap)) * Kout + kco] : w[((long)kco * 27 + tap) * Kin + kci];
        }
        const float v = pos == 0 ? g[0] : pos == 1 ? 0.5f * ((g[0] + g[1]) + g[2]) : pos == 2 ? 0.5f * ((g[0] - g[1]) + g[2]) : g[2];
        __bf16 h, m, l;
        split3_scalar(v, h, m, l);
        wp[i] = half == 0 ? h : (which == 0 ? m : l);
    }
}

bool k3_fwd_wino_b3_supported(int Cin, int Cout) { return Cin % 8 == 0 && Cout % 64 == 0; }
size_t k3_fwd_wino_b3_ws_bytes(int Cin, int Cout) {
    return ((size_t)(Cout / 64) * (Cin / 8) * 36 + B3Cfg::PD) * B3Cfg::TAP_Q * sizeof(uint4);
}

// Same contract as k3_fwd_wino_launch with ups == 1 (stat_part rows: k3_fwd_wino_bricks()).
int k3_fwd_wino_b3_launch(const float* x, const float* w, const float* bias, float* y, int N, int D, int H,
                          int W, int Cin, int Cout, int flip, void* ws, size_t ws_bytes, hipStream_t s,
                          float* stat_part) {
    SIV_CHECK_WS(k3_fwd_wino_b3_ws_bytes(Cin, Cout), ws_bytes, "conv3d_k3(winograd, split bf16)");
    const long total = (long)(Cout / 64) * (Cin / 8) * 36 * 2 * 64 * 2 * 8;
    long g = cdiv(total, 256);
    if (g > 4096) g = 4096;
    bool need_pack;
    void* wpk = pack_slot(w, PF_WINO_B3, 0, flip, Cin, Cout, k3_fwd_wino_b3_ws_bytes(Cin, Cout), ws, &need_pack);
    if (need_pack) {
        k3_pack_weights_wino_b3<<<(unsigned)g, 256, 0, s>>>(w, (__bf16*)wpk, Cin, Cout, flip);
        SIV_CHECK_PACK_LAUNCH("conv3d_k3(pack split-bf16 winograd weights)", wpk);
    }
    const int nbd = (D + 3) / 4, nbh = (H + 7) / 8, nbw = (W + 7) / 8;
    const long nblk = (long)N * nbd * nbh * nbw;
    if (nblk > 0x7fffffffL) {
        set_error("conv3d_k3(winograd, split bf16): too many bricks");
        return SIV_EINVAL;
    }
    k3_fwd_wino_b3<<<dim3((unsigned)nblk, (unsigned)(Cout / 64)), 256, B3Cfg::LDS_BYTES, s>>>(
        x, (const uint4*)wpk, bias, y, N, D, H, W, Cin, Cout, nbd, nbh, nbw, stat_part);
    SIV_CHECK_LAUNCH("conv3d_k3(winograd, split bf16)");
    return SIV_OK;
}

}  // namespace siv
