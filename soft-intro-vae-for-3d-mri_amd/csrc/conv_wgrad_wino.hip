// Weight gradient of the 3x3x3 convolution with the Winograd F(3x3, 2x2) minimal-filtering identity
// applied over (H, W): the 3x3 (kh, kw) gradient of one z-plane pairing is the correlation of a 2x2 tile
// of dy with the 4x4 input patch around it,
//   dW_kd = A^T [ sum over tiles (G dy G^T) .* (B^T x B) ] A
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 -1 0 1],  G = [1 0; .5 .5; .5 -.5; 0 1],  A^T = [1 1 1 0; 0 1 -1 0; 0 1 1 1]
// i.e. 16 multiply-adds per tile, plane pairing and channel pair instead of the 36 of the direct kernel.
// The sum over tiles (the whole batch and volume) is the contraction dimension of the fp32 matrix
// cores; A^T . A is applied once, after the partial sums of all blocks have been added.  Everything
// stays fp32; only the additions of the transforms change the rounding (the factors 1/2 are exact).
//
//   a block  = one row ph of the 4x4 transformed positions x 64 input x 64 output channels:
//              4 positions x 3 plane pairings kd x (64 x 64) = 96 accumulator registers per lane;
//              wave (pw, co half) owns position (ph, pw): 3 kd x 2 ci halves of 32x32
//   its work = a list of 8x8-voxel (H, W) columns of the volume, walked along z: per x-plane z the
//              transformed patch V[z] meets the transformed dy planes U[z+1], U[z], U[z-1] (kd = 0, 1, 2)
//   LDS      = ring of four U planes + two V planes, [position][channel][tile], 20 KB each
//
// Both operands come from LDS, so the global loads of the staging have no other loads behind them:
// they are issued right after their registers were consumed and land a whole step later.  Waves 0..3
// stage x (two patch rows suffice for one ph), waves 4..7 stage dy; the transform is cut into pieces
// that sit between the MFMA groups.
#include "siv_common.h"

namespace siv {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GwCfg {
    static constexpr int RH = 8, RW = 8;                 // output voxels of a column per plane
    static constexpr int T = 16;                         // its 2x2 tiles
    static constexpr int CS = 20;                        // floats per channel row: 16 tiles + 4 (bank spread)
    static constexpr int BUF = 4 * 64 * CS;              // one transformed plane: 4 positions x 64 channels
    static constexpr int NU = 4, NV = 2;
    static constexpr int LDS_FLOATS = (NU + NV) * BUF;
    static constexpr size_t LDS_BYTES = (size_t)LDS_FLOATS * sizeof(float);   // 120 KB: one block per CU
    static constexpr int NTHR = 512;
    static constexpr int NBLOCKS = 256;                  // one per CU; every block owns 1/64 of the columns
};

__device__ __forceinline__ float4 gw_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 gw_sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 gw_half(float4 a) { return make_float4(0.5f * a.x, 0.5f * a.y, 0.5f * a.z, 0.5f * a.w); }

// part[set][channel tile][ph][pw][kd][ci (64)][co (64)]
__global__ __launch_bounds__(512, 2) void k3_wgrad_wino2d(const float* __restrict__ x,
                                                          const float* __restrict__ dy,
                                                          float* __restrict__ part, int N, int D, int H, int W,
                                                          int Cin, int Cout, int nrh, int nrw, int zsplit,
                                                          int nsets) {
    using C = GwCfg;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int pw = wave & 3, coh = wave >> 2;

    // the four ph blocks of a set are neighbours on one XCD: they read the same columns at the same time
    int b = blockIdx.x;
    if ((gridDim.x & 7) == 0) b = (b & 7) * (gridDim.x >> 3) + (b >> 3);
    const int ph = b & 3, set = b >> 2;
    const int ncot = Cout >> 6;
    const int cit = blockIdx.y / ncot, cot = blockIdx.y % ncot;

    f32x16 acc[3][2];          // [kd][ci half]
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[k][a][r] = 0.f;

    const int aoff = (pw * 64 + li) * C::CS + 4 * lh;                 // + ci half * 32 * CS + 8 j
    const int boff = (pw * 64 + coh * 32 + li) * C::CS + 4 * lh;      // + 8 j

    // staging item: (tile, group of four channels); waves 0..3: x, waves 4..7: dy
    const bool isV = wave < 4;
    const int st = tid & 255;
    const int cg = st & 15, tile = st >> 4;
    const int th = tile >> 2, tw = tile & 3;
    const int soff = (4 * cg) * C::CS + tile;                          // + position * 64 * CS + e * CS
    // rows of the 4x4 patch (x) / of the 2x2 tile (dy) that position row ph combines
    const int rA = isV ? (ph == 0 ? 0 : ph == 1 ? 1 : ph == 2 ? 2 : 3) : (ph == 3 ? 1 : 0);
    const int rB = isV ? (ph == 0 ? 2 : ph == 1 ? 2 : 1) : (ph == 0 ? 0 : 1);
    const int Cs = isV ? Cin : Cout;                                   // channels of the staged tensor
    const float* sbase = isV ? x + cit * 64 + 4 * cg : dy + cot * 64 + 4 * cg;
    const int ncol = isV ? 4 : 2;
    const long plane = (long)H * W * Cs;

    float4 pa[4], pb[4];
    const int nitems = N * nrh * nrw * zsplit;
    for (int item = set; item < nitems; item += nsets) {
        int r = item / zsplit;
        const int zp = item - r * zsplit;
        const int rx = r % nrw; r /= nrw;
        const int ry = r % nrh;
        const int n = r / nrh;
        const int zlo = (int)((long)zp * D / zsplit), zhi = (int)((long)(zp + 1) * D / zsplit);
        if (zlo >= zhi) continue;

        // first row / column of this item's rows A and B, validity of its (up to) 2 x 4 voxels
        const int gy0 = ry * C::RH + 2 * th + (isV ? -1 : 0), gx0 = rx * C::RW + 2 * tw + (isV ? -1 : 0);
        unsigned mask = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bool okx = c < ncol && gx0 + c >= 0 && gx0 + c < W;
            mask |= (unsigned)(okx && gy0 + rA >= 0 && gy0 + rA < H) << c;
            mask |= (unsigned)(okx && gy0 + rB >= 0 && gy0 + rB < H) << (4 + c);
        }
        const int offA = ((gy0 + rA) * W + gx0) * Cs, offB = ((gy0 + rB) * W + gx0) * Cs;
        const float* sn = sbase + (long)n * D * plane;

        // loads of x-plane zv (waves 0..3) / dy-plane zu (waves 4..7); out of range: any valid address
        auto issue = [&](int zv, int zu) {
            int z = isV ? zv : zu;
            z = z < 0 ? 0 : z >= D ? D - 1 : z;
            const float* pl = sn + (long)z * plane;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c < 2 || isV) {
                    pa[c] = *reinterpret_cast<const float4*>(pl + (((mask >> c) & 1) ? offA + c * Cs : 0));
                    pb[c] = *reinterpret_cast<const float4*>(pl + (((mask >> (4 + c)) & 1) ? offB + c * Cs : 0));
                }
            }
        };
        // transform along H into pa[] (zeros for voxels outside the volume)
        auto hxf = [&](int zv, int zu) {
            const int z = isV ? zv : zu;
            const unsigned m = (z >= 0 && z < D) ? mask : 0u;
            const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c < 2 || isV) {
                    const float4 a = ((m >> c) & 1) ? pa[c] : zero;
                    const float4 bb = ((m >> (4 + c)) & 1) ? pb[c] : zero;
                    if (isV) pa[c] = ph == 1 ? gw_add(a, bb) : gw_sub(a, bb);
                    else pa[c] = (ph == 0 || ph == 3) ? a : ph == 1 ? gw_half(gw_add(a, bb)) : gw_half(gw_sub(a, bb));
                }
            }
        };
        // transform along W for position (ph, p) and its four channel rows
        auto wpiece = [&](int p, float* vdst, float* udst) {
            float4 v;
            if (isV) v = p == 0 ? gw_sub(pa[0], pa[2]) : p == 1 ? gw_add(pa[1], pa[2]) : p == 2 ? gw_sub(pa[2], pa[1]) : gw_sub(pa[3], pa[1]);
            else v = p == 0 ? pa[0] : p == 1 ? gw_half(gw_add(pa[0], pa[1])) : p == 2 ? gw_half(gw_sub(pa[0], pa[1])) : pa[1];
            float* d = (isV ? vdst : udst) + p * 64 * C::CS + soff;
            d[0] = v.x; d[C::CS] = v.y; d[2 * C::CS] = v.z; d[3 * C::CS] = v.w;
        };
        auto vbuf = [&](int z) { return lds + (C::NU + (z & 1)) * C::BUF; };
        auto ubuf = [&](int z) { return lds + (z & 3) * C::BUF; };

        // fill: V[zlo], U[zlo-1], U[zlo], U[zlo+1]; the loads of the first step's staging in flight
        issue(zlo, zlo - 1);
        hxf(zlo, zlo - 1);
#pragma unroll
        for (int p = 0; p < 4; ++p) wpiece(p, vbuf(zlo), ubuf(zlo - 1));
        if (!isV) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                issue(0, zlo + k);
                hxf(0, zlo + k);
#pragma unroll
                for (int p = 0; p < 4; ++p) wpiece(p, nullptr, ubuf(zlo + k));
            }
        }
        if (zlo + 1 < zhi) issue(zlo + 1, zlo + 2);
        __syncthreads();

        for (int z = zlo; z < zhi; ++z) {
            const float* vb = vbuf(z);
            const float* u[3] = {ubuf(z + 1), ubuf(z), ubuf(z - 1)};
            const bool more = z + 1 < zhi;
            float* vdst = vbuf(z + 1);
            float* udst = ubuf(z + 2);
            float4 af[2][2], bf[2][3];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int a = 0; a < 2; ++a) af[j][a] = *reinterpret_cast<const float4*>(vb + aoff + a * 32 * C::CS + 8 * j);
#pragma unroll
                for (int k = 0; k < 3; ++k) bf[j][k] = *reinterpret_cast<const float4*>(u[k] + boff + 8 * j);
            }
#pragma unroll
            for (int g = 0; g < 6; ++g) {
                const int j = g / 3, k = g % 3;
                __builtin_amdgcn_sched_barrier(0);
#define SIV_GWSTEP(F)                                                                       \
    _Pragma("unroll") for (int a = 0; a < 2; ++a)                                           \
        acc[k][a] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j][a].F, bf[j][k].F, acc[k][a], 0, 0, 0);
                SIV_GWSTEP(x)
                SIV_GWSTEP(y)
                SIV_GWSTEP(z)
                SIV_GWSTEP(w)
#undef SIV_GWSTEP
                __builtin_amdgcn_sched_barrier(0);
                if (more) {
                    if (g == 1) hxf(z + 1, z + 2);
                    if (g >= 2) wpiece(g - 2, vdst, udst);
                }
            }
            if (z + 2 < zhi) issue(z + 2, z + 3);
            __syncthreads();
        }
    }

    // partial sums of this block
    float* pp = part + ((((long)set * gridDim.y + blockIdx.y) * 4 + ph) * 4 + pw) * 3 * 4096;
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                pp[(k * 64 + ci) * 64 + coh * 32 + li] = acc[k][a][r];
            }
}

// dw[co][kd][kh][kw][ci] (+)= sum_{ph,pw} At[kh][ph] At[kw][pw] sum_set part[set][tile][ph][pw][kd][ci][co]
__global__ __launch_bounds__(256) void k3_wgrad_wino2d_finish(const float* __restrict__ part, int nsets,
                                                              int ntile, float* __restrict__ dw, int Cin,
                                                              int Cout, int accumulate) {
    const long total = (long)ntile * 3 * 4096;
    const int ncot = Cout >> 6;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int co = (int)(i & 63), ci = (int)((i >> 6) & 63);
        const int kd = (int)((i >> 12) % 3), tl = (int)((i >> 12) / 3);
        float m[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) m[p] = 0.f;
        for (int s = 0; s < nsets; ++s) {
            const float* pp = part + (((long)s * ntile + tl) * 16) * 3 * 4096 + (kd * 64 + ci) * 64 + co;
#pragma unroll
            for (int p = 0; p < 16; ++p) m[p] += pp[(long)p * 3 * 4096];
        }
        // rows of A^T over ph, then over pw
        float t[3][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            t[0][q] = (m[0 * 4 + q] + m[1 * 4 + q]) + m[2 * 4 + q];
            t[1][q] = m[1 * 4 + q] - m[2 * 4 + q];
            t[2][q] = (m[1 * 4 + q] + m[2 * 4 + q]) + m[3 * 4 + q];
        }
        const int cog = (tl % ncot) * 64 + co, cig = (tl / ncot) * 64 + ci;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const float g0 = (t[kh][0] + t[kh][1]) + t[kh][2];
            const float g1 = t[kh][1] - t[kh][2];
            const float g2 = (t[kh][1] + t[kh][2]) + t[kh][3];
            float* o = dw + ((long)cog * 27 + (kd * 3 + kh) * 3) * Cin + cig;
            if (accumulate) { o[0] += g0; o[Cin] += g1; o[2 * Cin] += g2; }
            else { o[0] = g0; o[Cin] = g1; o[2 * Cin] = g2; }
        }
    }
}

bool k3_wgrad_wino2d_supported(int Cin, int Cout) {
    return Cin % 64 == 0 && Cout % 64 == 0 && (Cin / 64) * (Cout / 64) <= 64;
}

static int gw_nsets(int Cin, int Cout) {
    int n = GwCfg::NBLOCKS / (4 * (Cin / 64) * (Cout / 64));
    return n < 1 ? 1 : n;
}
size_t k3_wgrad_wino2d_ws_bytes(int Cin, int Cout) {
    return (size_t)gw_nsets(Cin, Cout) * (Cin / 64) * (Cout / 64) * 48 * 4096 * sizeof(float);
}

int k3_wgrad_wino2d_launch(const float* x, const float* dy, float* dw, int N, int D, int H, int W, int Cin,
                           int Cout, int accumulate, void* ws, size_t ws_bytes, hipStream_t s) {
    SIV_CHECK_WS(k3_wgrad_wino2d_ws_bytes(Cin, Cout), ws_bytes, "conv3d_k3_wgrad(winograd)");
    SIV_CHECK_ARG((long)H * W * (Cin > Cout ? Cin : Cout) < (1L << 29),
                  "conv3d_k3_wgrad(winograd): plane too large for 32-bit offsets");
    const int ntile = (Cin / 64) * (Cout / 64);
    const int nsets = gw_nsets(Cin, Cout);
    const int nrh = (H + 7) / 8, nrw = (W + 7) / 8;
    // cut the columns along z until every set has at least a dozen items (load balance), while the
    // pieces stay long enough to amortise their pipeline fill
    const long ncol = (long)N * nrh * nrw;
    int zsplit = 1;
    while (ncol * zsplit < 12L * nsets && D / (zsplit * 2) >= 8) zsplit *= 2;
    if (ncol * zsplit > 0x7fffffffL) {
        set_error("conv3d_k3_wgrad(winograd): too many columns");
        return SIV_EINVAL;
    }
    if (hipFuncSetAttribute((const void*)k3_wgrad_wino2d, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)GwCfg::LDS_BYTES) != hipSuccess) {
        set_error("conv3d_k3_wgrad(winograd): cannot reserve %zu bytes of LDS", GwCfg::LDS_BYTES);
        return SIV_EHIP;
    }
    k3_wgrad_wino2d<<<dim3((unsigned)(nsets * 4), (unsigned)ntile), GwCfg::NTHR, GwCfg::LDS_BYTES, s>>>(
        x, dy, (float*)ws, N, D, H, W, Cin, Cout, nrh, nrw, zsplit, nsets);
    SIV_CHECK_LAUNCH("conv3d_k3_wgrad(winograd)");
    const long total = (long)ntile * 3 * 4096;
    k3_wgrad_wino2d_finish<<<(unsigned)cdiv(total, 256), 256, 0, s>>>((const float*)ws, nsets, ntile, dw, Cin,
                                                                        Cout, accumulate);
    SIV_CHECK_LAUNCH("conv3d_k3_wgrad(winograd, finish)");
    return SIV_OK;
}

}  // namespace siv
