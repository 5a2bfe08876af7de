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
// Both operands come from LDS, so the global loads of the staging have no other loads behind them and
// every wait on them is exact.  Waves 0..3 stage x (two patch rows suffice for one ph), waves 4..7 stage
// dy; each role runs its own copy of the multiply loop, a single basic block in which the transform is
// cut into units that follow single pairs of matrix instructions.
#include "siv_common.h"

// SIV_GWEXP: timing-only ablation mask (never set in the library build): 1 = no global loads of the
// staging, 2 = no transform / LDS writes, 4 = no barrier per step, 8 = no LDS fragment reads
#ifndef SIV_GWEXP
#define SIV_GWEXP 0
#endif

namespace siv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

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

struct GwGeom {
    int N, D, H, W, nrh, nrw, zsplit, nsets;
    int tpr;        // tiles per row of a column: 4 (8 x 8 voxels, 16 tiles) or 5 (6 x 10 voxels, 15 tiles)
};

// The work of one wave over all items of its block.  ISV: the wave stages x (4x4 patch, two rows of it
// for position row ph), else dy (2x2 tile).  `src` is the tensor, `chan` the first of the lane's four
// channels, `Cs` the channel count of that tensor.  The row combination is h = a + cB b with
//   x:  cB = +1 for ph = 1, else -1; rows (A, B) = (0,2) (1,2) (2,1) (3,1)
//   dy: cB = 0, +1, -1, 0;           rows (A, B) = (0,0) (0,1) (0,1) (1,1)
// The factors 1/2 of G (positions 1 and 2 along either axis) are left out of the dy transform and applied
// to the finished sums (exact: powers of two).  A sample is read through a buffer descriptor whose per-lane
// offset carries voxel AND plane, so a voxel or a whole plane outside the volume reads zeros: no masks, no
// 64-bit address arithmetic, and the transforms are packed two-channel operations -- every vector
// instruction a wave issues here takes time on the matrix pipe.
template <bool ISV>
__device__ __forceinline__ void gw_run(const float* __restrict__ src, const int chan, const int Cs,
                                       f32x16 (&acc)[3][2], float* lds, const int ph, const int set,
                                       const GwGeom g, const int aoff, const int boff, const int cg,
                                       const int tile) {
    using C = GwCfg;
    constexpr int NC = ISV ? 4 : 2;
    const int th = tile / g.tpr, tw = tile - th * g.tpr;
    const bool tile_ok = th < 16 / g.tpr;                                 // slot 15 is empty with 5 tiles per row
    const int colh = 2 * (16 / g.tpr), colw = 2 * g.tpr;                  // voxels of a column: 8 x 8 or 6 x 10
    const int soff = (4 * cg) * C::CS + tile;                          // + position * 64 * CS + e * CS
    const int rA = ISV ? ph : (ph == 3 ? 1 : 0);
    const int rB = ISV ? (ph <= 1 ? 2 : 1) : (ph == 0 ? 0 : 1);
    const float kB = ISV ? (ph == 1 ? 1.f : -1.f) : (ph == 1 ? 1.f : ph == 2 ? -1.f : 0.f);
    const f32x2_t kB2 = {kB, kB};
    const unsigned plane4 = (unsigned)(g.H * g.W * Cs) * 4u;              // bytes of a plane
    constexpr unsigned OUTSIDE = 0x80000000u;

    f32x2x2 pa[NC], pb[NC];
    unsigned oa[NC], ob[NC];
    __amdgpu_buffer_rsrc_t rs = buf_rsrc(src, 0);

    // plane z of this item's sample; z may lie outside [0, D): the offset is then outside the descriptor
    auto issue_col = [&](int c, int z) {
        if (SIV_GWEXP & 1) return;
        const unsigned zo = (unsigned)z * plane4;
        pa[c] = buf_load16p(rs, oa[c] + zo, 0);
        pb[c] = buf_load16p(rs, ob[c] + zo, 0);
    };
    // rows A and B of column c combined in place
    auto hxf_col = [&](int c) {
        pa[c].lo = pk_fma(kB2, pb[c].lo, pa[c].lo);
        pa[c].hi = pk_fma(kB2, pb[c].hi, pa[c].hi);
    };
    // transform along W for position (ph, p), then its four channel rows
    auto wpiece = [&](int p, float* dst) {
        f32x2x2 v;
        if (ISV) {
            if (p == 0) v = {pk_sub(pa[0].lo, pa[NC - 2].lo), pk_sub(pa[0].hi, pa[NC - 2].hi)};
            else if (p == 1) v = {pk_add(pa[1].lo, pa[NC - 2].lo), pk_add(pa[1].hi, pa[NC - 2].hi)};
            else if (p == 2) v = {pk_sub(pa[NC - 2].lo, pa[1].lo), pk_sub(pa[NC - 2].hi, pa[1].hi)};
            else v = {pk_sub(pa[NC - 1].lo, pa[1].lo), pk_sub(pa[NC - 1].hi, pa[1].hi)};
        } else {
            if (p == 0) v = pa[0];
            else if (p == 1) v = {pk_add(pa[0].lo, pa[1].lo), pk_add(pa[0].hi, pa[1].hi)};
            else if (p == 2) v = {pk_sub(pa[0].lo, pa[1].lo), pk_sub(pa[0].hi, pa[1].hi)};
            else v = pa[1];
        }
        float* d = dst + p * 64 * C::CS + soff;
        d[0] = v.lo.x; d[C::CS] = v.lo.y; d[2 * C::CS] = v.hi.x; d[3 * C::CS] = v.hi.y;
    };
    auto vbuf = [&](int z) { return lds + (C::NU + (z & 1)) * C::BUF; };
    auto ubuf = [&](int z) { return lds + (z & 3) * C::BUF; };
    // immediate staging of one plane (pipeline fill)
    auto stage_now = [&](int z) {
#pragma unroll
        for (int c = 0; c < NC; ++c) issue_col(c, z);
#pragma unroll
        for (int c = 0; c < NC; ++c) hxf_col(c);
        float* dst = ISV ? vbuf(z) : ubuf(z);
#pragma unroll
        for (int p = 0; p < 4; ++p) wpiece(p, dst);
    };

    const int nitems = g.N * g.nrh * g.nrw * g.zsplit;
    for (int item = set; item < nitems; item += g.nsets) {
        int r = item / g.zsplit;
        const int zp = item - r * g.zsplit;
        const int rx = r % g.nrw; r /= g.nrw;
        const int ry = r % g.nrh;
        const int n = r / g.nrh;
        const int zlo = (int)((long)zp * g.D / g.zsplit), zhi = (int)((long)(zp + 1) * g.D / g.zsplit);
        if (zlo >= zhi) continue;

        // byte offsets of this item's 2 x NC voxels inside a plane of its sample
        const int gy0 = ry * colh + 2 * th + (ISV ? -1 : 0), gx0 = rx * colw + 2 * tw + (ISV ? -1 : 0);
        const bool okA = tile_ok && gy0 + rA >= 0 && gy0 + rA < g.H, okB = tile_ok && gy0 + rB >= 0 && gy0 + rB < g.H;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const bool okx = gx0 + c >= 0 && gx0 + c < g.W;
            oa[c] = (okx && okA) ? (unsigned)((((gy0 + rA) * g.W + gx0 + c) * Cs + chan) * 4) : OUTSIDE;
            ob[c] = (okx && okB) ? (unsigned)((((gy0 + rB) * g.W + gx0 + c) * Cs + chan) * 4) : OUTSIDE;
        }
        rs = buf_rsrc(src + (long)n * g.D * g.H * g.W * Cs, (long)g.D * plane4);

        // fill: V[zlo], U[zlo-1], U[zlo], U[zlo+1]; then the loads of the first step's staging
        if (ISV) {
            stage_now(zlo);
        } else {
            stage_now(zlo - 1);
            stage_now(zlo);
            stage_now(zlo + 1);
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) issue_col(c, ISV ? zlo + 1 : zlo + 2);
        __syncthreads();

        for (int z = zlo; z < zhi; ++z) {
            const float* vb = vbuf(z);
            const float* u[3] = {ubuf(z + 1), ubuf(z), ubuf(z - 1)};
            // this step stages x-plane z + 1 / dy-plane z + 2 (loaded during the previous step) and
            // fetches the planes after them; past the end of the item both are harmless extra work
            const int zst = ISV ? z + 1 : z + 2;
            float* dst = ISV ? vbuf(zst) : ubuf(zst);
            f32x4 af[2][2], bf[2][3];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    if (SIV_GWEXP & 8) af[j][a] = f32x4{(float)z, 1.f, 2.f, 3.f};
                    else af[j][a] = *reinterpret_cast<const f32x4*>(vb + aoff + a * 32 * C::CS + 8 * j);
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    if (SIV_GWEXP & 8) bf[j][k] = f32x4{(float)z, 1.f, 2.f, 3.f};
                    else bf[j][k] = *reinterpret_cast<const f32x4*>(u[k] + boff + 8 * j);
                }
            }
            // 24 MFMA pairs; the staging is cut into units that follow single pairs, so that the vector
            // and LDS instructions of a wave run in the shadow of its own matrix instructions
#pragma unroll
            for (int m = 0; m < 24; ++m) {
                const int gq = m >> 2, e = m & 3;
                const int j = gq / 3, k = gq % 3;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    acc[k][a] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j][a][e], bf[j][k][e], acc[k][a], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (!(SIV_GWEXP & 2)) {
                    if (m >= 2 && m < 2 + NC) hxf_col(m - 2);
                    if (m == 8 || m == 11 || m == 14 || m == 17) wpiece((m - 8) / 3, dst);
                }
                // The loads of the next planes go last, into the registers the transform has just
                // released.  (Issued earlier, into registers of their own, they cost 5-15 % more: a
                // wave that issues 1 KB loads between its matrix instructions holds those back, while
                // at the end of the step the issue overlaps the wait at the barrier.)
                if (m >= 20 && m - 20 < NC) issue_col(m - 20, zst + 1);
            }
            if (!(SIV_GWEXP & 4)) __syncthreads();
        }
    }
}

// part[set][channel tile][ph][pw][kd][ci (64)][co (64)]
__global__ __launch_bounds__(512, 2) void k3_wgrad_wino2d(const float* __restrict__ x,
                                                          const float* __restrict__ dy,
                                                          float* __restrict__ part, int N, int D, int H, int W,
                                                          int Cin, int Cout, int nrh, int nrw, int zsplit,
                                                          int nsets, int tpr) {
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
    const int st = tid & 255;
    const int cg = st & 15, tile = st >> 4;
    const GwGeom g = {N, D, H, W, nrh, nrw, zsplit, nsets, tpr};
    if (wave < 4) gw_run<true>(x, cit * 64 + 4 * cg, Cin, acc, lds, ph, set, g, aoff, boff, cg, tile);
    else gw_run<false>(dy, cot * 64 + 4 * cg, Cout, acc, lds, ph, set, g, aoff, boff, cg, tile);

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

// part[0][e] = sum_s part[s][e] in fixed order (every thread owns its elements: safe in place)
__global__ __launch_bounds__(256) void gw_sum_sets(float* __restrict__ part, int nsets, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        // eight sets in flight per thread (independent loads, added in set order): the loop is bound by load
        // latency, and on the side stream it competes with the main stream's kernels for every CU it gets
        float s = 0.f;
        int k = 0;
        for (; k + 8 <= nsets; k += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(long)(k + u) * n + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < nsets; ++k) s += part[(long)k * n + i];
        part[i] = s;
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
        {   // the sets have been summed into part[0] (gw_sum_sets)
            const float* pp = part + ((long)tl * 16) * 3 * 4096 + (kd * 64 + ci) * 64 + co;
            // (the factors 1/2 of the dy transform, positions 1 and 2 along H and along W, are applied here)
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const float sc = (((p >> 2) == 1 || (p >> 2) == 2) ? 0.5f : 1.f) * (((p & 3) == 1 || (p & 3) == 2) ? 0.5f : 1.f);
                m[p] = sc * pp[(long)p * 3 * 4096];
            }
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

// ---------------------------------------------------------------------------------------------
// The same weight-gradient-stationary kernel for a convolution of a nearest x2 upsampled input
// (x is the LOW-resolution tensor).  Along H and W the 4x4 patch of an upsampled plane around a 2x2
// tile is (a, b, b, c) per axis, so B^T x = (a - b, 2 b, 0, c - b): one tile = one low-resolution
// voxel and its 3x3 neighbourhood, and only the 3 x 3 positions {0, 1, 3}^2 are non-zero.  Along z
// the planes pair up: with s the low-resolution plane,
//   P00 = sum dy[2s] x[s-1],  P01 = sum dy[2s] x[s],  P10 = sum dy[2s+1] x[s],  P11 = sum dy[2s+1] x[s+1]
//   dW[kd=0] = P00 + P10,  dW[1] = P01 + P10,  dW[2] = P01 + P11
// i.e. per x-plane s the transformed patch V[s] meets U[2s+2], U[2s], U[2s+1], U[2s-1]: 9 positions x 4
// pairings = 36 multiply-adds per low-resolution voxel and channel pair for 8 outputs (4.5 per output;
// the collapsed-tap kernel issues 8, the direct one 27).
//   a block = one position row (ph in {0, 1, 3}) x 64 ci x 64 co: 3 positions x 4 pairings x 4 tiles;
//             wave (pairing, co half) owns 3 positions x 2 ci halves = 96 accumulator registers
//   LDS     = ring of six U planes + two V planes of 3 positions, 15 KB each
// ---------------------------------------------------------------------------------------------
struct GuCfg {
    static constexpr int CS = 20;
    static constexpr int BUF = 3 * 64 * CS;
    static constexpr int NU = 6, NV = 2;
    static constexpr int LDS_FLOATS = (NU + NV) * BUF;
    static constexpr size_t LDS_BYTES = (size_t)LDS_FLOATS * sizeof(float);   // 120 KB
    static constexpr int NTHR = 512;
    static constexpr int NBLOCKS = 255;                  // 3 position rows x 85 sets
};

// ISV: the wave stages x (low resolution; NP = 1 plane per step), else dy (NP = 2 planes per step).
// D, H, W in `g` are the full-resolution extents.  Loads, row combination h = a + cB b and the deferred
// factors 1/2 of the dy transform (position 1 along either axis) as in gw_run.
template <bool ISV>
__device__ __forceinline__ void gu_run(const float* __restrict__ src, const int chan, const int Cs,
                                       f32x16 (&acc)[3][2], float* lds, const int phi, const int set,
                                       const GwGeom g, const int aoff, const int boff, const int zt,
                                       const int cg, const int tile) {
    using C = GuCfg;
    constexpr int NC = ISV ? 3 : 2, NP = ISV ? 1 : 2;
    const int th = tile >> 2, tw = tile & 3;
    const int soff = (4 * cg) * C::CS + tile;
    const int Dl = g.D >> 1, Hs = ISV ? g.H >> 1 : g.H, Ws = ISV ? g.W >> 1 : g.W, Ds = ISV ? Dl : g.D;
    // rows combined for position row phi (positions 0, 1, 3):
    //   x:  (jy-1, jy; -1)  (jy, jy; +1)  (jy+1, jy; -1)      dy: (0, 0; 0) (0, 1; +1) (1, 1; 0)
    const int rA = ISV ? phi - 1 : (phi == 2 ? 1 : 0);
    const int rB = ISV ? 0 : (phi == 0 ? 0 : 1);
    const float kB = ISV ? (phi == 1 ? 1.f : -1.f) : (phi == 1 ? 1.f : 0.f);
    const f32x2_t kB2 = {kB, kB};
    const unsigned plane4 = (unsigned)(Hs * Ws * Cs) * 4u;
    constexpr unsigned OUTSIDE = 0x80000000u;

    f32x2x2 pa[NP][NC], pb[NP][NC];
    unsigned oa[NC], ob[NC];
    __amdgpu_buffer_rsrc_t rs = buf_rsrc(src, 0);

    auto issue_col = [&](int q, int c, int z) {
        const unsigned zo = (unsigned)z * plane4;
        pa[q][c] = buf_load16p(rs, oa[c] + zo, 0);
        pb[q][c] = buf_load16p(rs, ob[c] + zo, 0);
    };
    auto hxf_col = [&](int q, int c) {
        pa[q][c].lo = pk_fma(kB2, pb[q][c].lo, pa[q][c].lo);
        pa[q][c].hi = pk_fma(kB2, pb[q][c].hi, pa[q][c].hi);
    };
    // position (phi, p), p = 0, 1, 2 standing for 0, 1, 3
    auto wpiece = [&](int q, int p, float* dst) {
        f32x2x2 v;
        if (ISV) {
            if (p == 0) v = {pk_sub(pa[q][0].lo, pa[q][1].lo), pk_sub(pa[q][0].hi, pa[q][1].hi)};
            else if (p == 1) v = {pk_add(pa[q][1].lo, pa[q][1].lo), pk_add(pa[q][1].hi, pa[q][1].hi)};
            else v = {pk_sub(pa[q][NC - 1].lo, pa[q][1].lo), pk_sub(pa[q][NC - 1].hi, pa[q][1].hi)};
        } else {
            if (p == 0) v = pa[q][0];
            else if (p == 1) v = {pk_add(pa[q][0].lo, pa[q][1].lo), pk_add(pa[q][0].hi, pa[q][1].hi)};
            else v = pa[q][1];
        }
        float* d = dst + p * 64 * C::CS + soff;
        d[0] = v.lo.x; d[C::CS] = v.lo.y; d[2 * C::CS] = v.hi.x; d[3 * C::CS] = v.hi.y;
    };
    auto vbuf = [&](int z) { return lds + (C::NU + (z & 1)) * C::BUF; };
    auto ubuf = [&](int z) { return lds + ((z + 6) % 6) * C::BUF; };        // z >= -1
    auto stage_now = [&](int z) {                      // planes z .. z + NP - 1
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int c = 0; c < NC; ++c) issue_col(q, c, z + q);
#pragma unroll
        for (int q = 0; q < NP; ++q) {
#pragma unroll
            for (int c = 0; c < NC; ++c) hxf_col(q, c);
            float* dst = ISV ? vbuf(z + q) : ubuf(z + q);
#pragma unroll
            for (int p = 0; p < 3; ++p) wpiece(q, p, dst);
        }
    };

    const int nitems = g.N * g.nrh * g.nrw * g.zsplit;
    for (int item = set; item < nitems; item += g.nsets) {
        int r = item / g.zsplit;
        const int zp = item - r * g.zsplit;
        const int rx = r % g.nrw; r /= g.nrw;
        const int ry = r % g.nrh;
        const int n = r / g.nrh;
        const int slo = (int)((long)zp * Dl / g.zsplit), shi = (int)((long)(zp + 1) * Dl / g.zsplit);
        if (slo >= shi) continue;

        // x: low-resolution voxel (ry*4 + th, rx*4 + tw) and its neighbours; dy: its 2x2 tile
        const int gy0 = ISV ? ry * 4 + th : 2 * (ry * 4 + th), gx0 = ISV ? rx * 4 + tw - 1 : 2 * (rx * 4 + tw);
        const bool okA = gy0 + rA >= 0 && gy0 + rA < Hs, okB = gy0 + rB >= 0 && gy0 + rB < Hs;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const bool okx = gx0 + c >= 0 && gx0 + c < Ws;
            oa[c] = (okx && okA) ? (unsigned)((((gy0 + rA) * Ws + gx0 + c) * Cs + chan) * 4) : OUTSIDE;
            ob[c] = (okx && okB) ? (unsigned)((((gy0 + rB) * Ws + gx0 + c) * Cs + chan) * 4) : OUTSIDE;
        }
        rs = buf_rsrc(src + (long)n * Ds * Hs * Ws * Cs, (long)Ds * plane4);

        // fill: V[slo], U[2 slo - 1 .. 2 slo + 2]; then the loads of the first step's staging
        if (ISV) {
            stage_now(slo);
        } else {
            stage_now(2 * slo - 1);
            stage_now(2 * slo + 1);
        }
        const int zfirst = ISV ? slo + 1 : 2 * slo + 3;
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int c = 0; c < NC; ++c) issue_col(q, c, zfirst + q);
        __syncthreads();

        for (int z = slo; z < shi; ++z) {
            const float* vb = vbuf(z);
            const float* ub = ubuf(2 * z + (zt == 0 ? 2 : zt == 1 ? 0 : zt == 2 ? 1 : -1));
            // this step stages x-plane z + 1 / dy-planes 2 z + 3, 2 z + 4 and fetches the ones after them
            const int zst = ISV ? z + 1 : 2 * z + 3;
            float* dst0 = ISV ? vbuf(zst) : ubuf(zst);
            float* dst1 = ISV ? dst0 : ubuf(zst + 1);
            f32x4 af[2][3][2], bf[2][3];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
#pragma unroll
                    for (int a = 0; a < 2; ++a)
                        af[j][p][a] = *reinterpret_cast<const f32x4*>(vb + aoff + (p * 64 + a * 32) * C::CS + 8 * j);
                    bf[j][p] = *reinterpret_cast<const f32x4*>(ub + boff + p * 64 * C::CS + 8 * j);
                }
#pragma unroll
            for (int m = 0; m < 24; ++m) {
                const int gq = m >> 2, e = m & 3;
                const int j = gq / 3, p = gq % 3;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    acc[p][a] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j][p][a][e], bf[j][p][e], acc[p][a], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (ISV) {
                    if (m >= 2 && m < 5) hxf_col(0, m - 2);
                    if (m == 8 || m == 11 || m == 14) wpiece(0, (m - 8) / 3, dst0);
                    if (m >= 20 && m < 23) issue_col(0, m - 20, zst + 1);
                } else {
                    if (m >= 2 && m < 6) hxf_col((m - 2) >> 1, (m - 2) & 1);
                    if (m >= 7 && m < 19 && ((m - 7) & 1) == 0) wpiece((m - 7) / 6, ((m - 7) >> 1) % 3, (m - 7) / 6 ? dst1 : dst0);
                    if (m >= 20) issue_col((m - 20) >> 1, (m - 20) & 1, zst + 2 + ((m - 20) >> 1));
                }
            }
            __syncthreads();
        }
    }
}

// part[set][channel tile][phi (3)][p (3)][pairing (4)][ci (64)][co (64)]
__global__ __launch_bounds__(512, 2) void k2_up_wgrad_wino(const float* __restrict__ x,
                                                           const float* __restrict__ dy,
                                                           float* __restrict__ part, int N, int D, int H, int W,
                                                           int Cin, int Cout, int nrh, int nrw, int zsplit,
                                                           int nsets) {
    using C = GuCfg;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int zt = wave & 3, coh = wave >> 2;
    const int phi = blockIdx.x % 3, set = blockIdx.x / 3;
    const int ncot = Cout >> 6;
    const int cit = blockIdx.y / ncot, cot = blockIdx.y % ncot;

    f32x16 acc[3][2];          // [p][ci half]
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[k][a][r] = 0.f;

    const int aoff = li * C::CS + 4 * lh;
    const int boff = (coh * 32 + li) * C::CS + 4 * lh;
    const int st = tid & 255;
    const int cg = st & 15, tile = st >> 4;
    const GwGeom g = {N, D, H, W, nrh, nrw, zsplit, nsets, 4};
    if (wave < 4) gu_run<true>(x, cit * 64 + 4 * cg, Cin, acc, lds, phi, set, g, aoff, boff, zt, cg, tile);
    else gu_run<false>(dy, cot * 64 + 4 * cg, Cout, acc, lds, phi, set, g, aoff, boff, zt, cg, tile);

    float* pp = part + (((long)set * gridDim.y + blockIdx.y) * 3 + phi) * 12 * 4096;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                pp[((p * 4 + zt) * 64 + ci) * 64 + coh * 32 + li] = acc[p][a][r];
            }
}

__global__ __launch_bounds__(256) void k2_up_wgrad_wino_finish(const float* __restrict__ part, int nsets,
                                                               int ntile, float* __restrict__ dw, int Cin,
                                                               int Cout, int accumulate) {
    const long total = (long)ntile * 4096;
    const int ncot = Cout >> 6;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int co = (int)(i & 63), ci = (int)((i >> 6) & 63), tl = (int)(i >> 12);
        float m[36];                       // [phi][p][pairing]
#pragma unroll
        for (int q = 0; q < 36; ++q) m[q] = 0.f;
        {   // the sets have been summed into part[0] (gw_sum_sets)
            const float* pp = part + (long)tl * 36 * 4096 + ci * 64 + co;
            // (the factors 1/2 of the dy transform, position 1 along H and along W, are applied here)
#pragma unroll
            for (int q = 0; q < 36; ++q)
                m[q] = ((q / 12 == 1) ? 0.5f : 1.f) * (((q / 4) % 3 == 1) ? 0.5f : 1.f) * pp[(long)q * 4096];
        }
        const int cog = (tl % ncot) * 64 + co, cig = (tl / ncot) * 64 + ci;
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
            // pairings of plane offset kd: 0 <- P00 + P10, 1 <- P01 + P10, 2 <- P01 + P11
            const int z0 = kd == 0 ? 0 : 1, z1 = kd == 2 ? 3 : 2;
            float t[3][3];                 // [phi][p]
#pragma unroll
            for (int q = 0; q < 9; ++q) t[q / 3][q % 3] = m[q * 4 + z0] + m[q * 4 + z1];
            // positions (0, 1, 3) -> taps: 0 <- T0 + T1, 1 <- T1, 2 <- T3 + T1, along H then W
            float u[3][3];                 // [kh][p]
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                u[0][q] = t[0][q] + t[1][q];
                u[1][q] = t[1][q];
                u[2][q] = t[2][q] + t[1][q];
            }
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const float g0 = u[kh][0] + u[kh][1], g1 = u[kh][1], g2 = u[kh][2] + u[kh][1];
                float* o = dw + ((long)cog * 27 + (kd * 3 + kh) * 3) * Cin + cig;
                if (accumulate) { o[0] += g0; o[Cin] += g1; o[2 * Cin] += g2; }
                else { o[0] = g0; o[Cin] = g1; o[2 * Cin] = g2; }
            }
        }
    }
}

static int gu_nsets(int Cin, int Cout) {
    int n = GuCfg::NBLOCKS / (3 * (Cin / 64) * (Cout / 64));
    return n < 1 ? 1 : n;
}
size_t k2_up_wgrad_wino_ws_bytes(int Cin, int Cout) {
    return (size_t)gu_nsets(Cin, Cout) * (Cin / 64) * (Cout / 64) * 36 * 4096 * sizeof(float);
}

// x: [N][D/2][H/2][W/2][Cin], dy: [N][D][H][W][Cout] (D, H, W even), Cin % 64 == 0, Cout % 64 == 0
int k2_up_wgrad_wino_launch(const float* x, const float* dy, float* dw, int N, int D, int H, int W, int Cin,
                            int Cout, int accumulate, void* ws, size_t ws_bytes, hipStream_t s) {
    SIV_CHECK_WS(k2_up_wgrad_wino_ws_bytes(Cin, Cout), ws_bytes, "conv3d_k3_wgrad(upsampled, winograd)");
    SIV_CHECK_ARG((long)(D + 3) * H * W * (Cin > Cout ? Cin : Cout) < (1L << 29),
                  "conv3d_k3_wgrad(upsampled, winograd): sample too large for 32-bit byte offsets");
    const int ntile = (Cin / 64) * (Cout / 64);
    const int nsets = gu_nsets(Cin, Cout);
    const int nrh = (H + 7) / 8, nrw = (W + 7) / 8;
    const long ncol = (long)N * nrh * nrw;
    int zsplit = 1;
    while (ncol * zsplit < 12L * nsets && (D / 2) / (zsplit * 2) >= 5) zsplit *= 2;
    if (ncol * zsplit > 0x7fffffffL) {
        set_error("conv3d_k3_wgrad(upsampled, winograd): too many columns");
        return SIV_EINVAL;
    }
    if (hipFuncSetAttribute((const void*)k2_up_wgrad_wino, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)GuCfg::LDS_BYTES) != hipSuccess) {
        set_error("conv3d_k3_wgrad(upsampled, winograd): cannot reserve %zu bytes of LDS", GuCfg::LDS_BYTES);
        return SIV_EHIP;
    }
    k2_up_wgrad_wino<<<dim3((unsigned)(nsets * 3), (unsigned)ntile), GuCfg::NTHR, GuCfg::LDS_BYTES, s>>>(
        x, dy, (float*)ws, N, D, H, W, Cin, Cout, nrh, nrw, zsplit, nsets);
    SIV_CHECK_LAUNCH("conv3d_k3_wgrad(upsampled, winograd)");
    const long nel = (long)ntile * 36 * 4096;
    gw_sum_sets<<<(unsigned)cdiv(nel, 256), 256, 0, s>>>((float*)ws, nsets, nel);
    SIV_CHECK_LAUNCH("conv3d_k3_wgrad(upsampled, winograd, sum)");
    const long total = (long)ntile * 4096;
    k2_up_wgrad_wino_finish<<<(unsigned)cdiv(total, 256), 256, 0, s>>>((const float*)ws, nsets, ntile, dw, Cin,
                                                                         Cout, accumulate);
    SIV_CHECK_LAUNCH("conv3d_k3_wgrad(upsampled, winograd, finish)");
    return SIV_OK;
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
    SIV_CHECK_ARG((long)(D + 3) * H * W * (Cin > Cout ? Cin : Cout) < (1L << 29),
                  "conv3d_k3_wgrad(winograd): sample too large for 32-bit byte offsets");
    const int ntile = (Cin / 64) * (Cout / 64);
    const int nsets = gw_nsets(Cin, Cout);
    // 6 x 10-voxel columns (15 tiles) where they cover the plane with less padding than 8 x 8 ones
    // (10-, 20-wide levels of an 80 x 96 x 80 volume)
    const long pad8 = cdiv(H, 8) * 8 * cdiv(W, 8) * 8, pad10 = cdiv(H, 6) * 6 * cdiv(W, 10) * 10;
    const int tpr = pad10 * 16 < pad8 * 15 ? 5 : 4;
    const int nrh = tpr == 5 ? (int)cdiv(H, 6) : (H + 7) / 8, nrw = tpr == 5 ? (int)cdiv(W, 10) : (W + 7) / 8;
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
        x, dy, (float*)ws, N, D, H, W, Cin, Cout, nrh, nrw, zsplit, nsets, tpr);
    SIV_CHECK_LAUNCH("conv3d_k3_wgrad(winograd)");
    const long nel = (long)ntile * 48 * 4096;
    gw_sum_sets<<<(unsigned)cdiv(nel, 256), 256, 0, s>>>((float*)ws, nsets, nel);
    SIV_CHECK_LAUNCH("conv3d_k3_wgrad(winograd, sum)");
    const long total = (long)ntile * 3 * 4096;
    k3_wgrad_wino2d_finish<<<(unsigned)cdiv(total, 256), 256, 0, s>>>((const float*)ws, nsets, ntile, dw, Cin,
                                                                        Cout, accumulate);
    SIV_CHECK_LAUNCH("conv3d_k3_wgrad(winograd, finish)");
    return SIV_OK;
}

}  // namespace siv
