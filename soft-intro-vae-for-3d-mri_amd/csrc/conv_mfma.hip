// 3x3x3 convolution for wide channel counts on the fp32 matrix cores of gfx950.
//
// The wide layers are compute bound (arithmetic intensity > 400 FLOP/B against a ridge of
// ~20), so they are written as an implicit GEMM on v_mfma_f32_32x32x2_f32: exact fp32
// (a k-ordered fmaf chain), issued at the full fp32 rate of the chip.  No im2col buffer
// exists: the input halo brick and the weight slice are staged in LDS and the 27 taps are
// read from the halo at shifted offsets.
//
//   forward / dgrad : M = voxels of a spatial brick, N = 64 output channels, K = 27 * Cin
//   wgrad           : M = 32 co, N = 32 ci per wave and tap, K = voxels (split over blocks)
//
// MFMA operand maps (32x32x2 f32): lane l holds A[i = l&31][k = l>>5] and
// B[k = l>>5][j = l&31]; accumulator register r of lane l is D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31].
#include "siv_common.h"

// SIV_EXP is a timing-only ablation mask used by tools/ubench (never set in the library build):
// 1 = no weight-fragment loads, 2 = no LDS fragment reads, 4 = no halo staging,
// 8 = no output stores, 16 = no block barriers, 32 = no LDS halo writes, 128 = scalar epilogue, 256 = no XCD-aware brick order.
#ifndef SIV_EXP
#define SIV_EXP 0
#endif

namespace siv {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------
// forward (and dgrad through transformed weights)
// ---------------------------------------------------------------------------------
// Weights are first re-packed (a 7 MB copy at most) into the order the kernel consumes them:
//   wp[co_tile][ci_chunk][tap][co (64)][ci (8)]
// so that the B fragments of one (chunk, tap) are 2 KB of contiguous memory that every wave
// reads with one coalesced 16-byte load per lane, straight into registers and one tap ahead
// of its use.  Only the input halo goes through LDS: one barrier pair per 8-channel chunk
// (27 taps = 432 MFMAs per wave) instead of one per weight slice.
template <int BD, int BH, int BW, int WM, int WN, int NT, int UPS>
struct FwdCfg {
    static constexpr int CK = 8;               // input channels per chunk
    static constexpr int M = BD * BH * BW;
    static constexpr int MS = (M + 31) / 32;   // 32-row sub-tiles in the brick (rows >= M are padding)
    static constexpr int NW = WM * WN;         // waves per block
    static constexpr int NTHR = 64 * NW;
    static constexpr int NS = NT / 32;         // NT (32 or 64) output channels per block
    static constexpr int SLICE = NT * 8;       // floats of one packed (chunk, tap) weight slice
    static constexpr int AM = MS / WM;         // accumulators per wave along M
    static constexpr int AN = NS / WN;
    static constexpr int HD = BD + 2, HH = BH + 2, HW = BW + 2;
    static constexpr int HV = HD * HH * HW;
    static constexpr int CKP = CK + 4;         // padded channel stride (keeps 16-B alignment)
    static constexpr int HAL_FLOATS = HV * CKP;
    static constexpr int EPI_STRIDE = 40;      // row stride of the epilogue's transpose tile: the two
                                               // lane halves (4 rows apart) land 32 banks apart
    static constexpr int EPI_FLOATS = 32 * EPI_STRIDE;
    static constexpr int LDS_FLOATS = HAL_FLOATS > NW * EPI_FLOATS ? HAL_FLOATS : NW * EPI_FLOATS;
    static constexpr size_t LDS_BYTES = (size_t)LDS_FLOATS * sizeof(float);
    static_assert(MS % WM == 0 && NS % WN == 0, "bad tiling");
};

template <int BD, int BH, int BW, int WM, int WN, int NT, int UPS>
__global__ __launch_bounds__(64 * WM * WN) void k3_fwd_mfma(const float* __restrict__ x,
                                                   const float* __restrict__ wp,
                                                   const float* __restrict__ bias,
                                                   float* __restrict__ y, int N, int D, int H, int W,
                                                   int Cin, int Cout, int nbd, int nbh, int nbw) {
    using C = FwdCfg<BD, BH, BW, WM, WN, NT, UPS>;
    extern __shared__ __attribute__((aligned(16))) float hal[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave % WM, wn = wave / WM;

    // Workgroups are handed to the 8 XCDs round-robin.  Remap the brick index so that each XCD (and
    // its private L2) works through one contiguous eighth of the brick list: bricks that share halo
    // voxels then run on the same L2 at about the same time.
    int b = blockIdx.x;
    if (!(SIV_EXP & 256) && (gridDim.x & 7) == 0) b = (b & 7) * (gridDim.x >> 3) + (b >> 3);
    const int bx = b % nbw; b /= nbw;
    const int by = b % nbh; b /= nbh;
    const int bz = b % nbd;
    const int n = b / nbd;
    const int z0 = bz * BD, y0 = by * BH, x0 = bx * BW;
    const int co0 = blockIdx.y * NT;
    const int Di = D / UPS, Hi = H / UPS, Wi = W / UPS;
    const int nchunk = (Cin + C::CK - 1) / C::CK;

    f32x16 acc[C::AM][C::AN];
#pragma unroll
    for (int a = 0; a < C::AM; ++a)
#pragma unroll
        for (int bb = 0; bb < C::AN; ++bb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][bb][r] = 0.f;

    // per-lane LDS offsets of the A rows (halo voxel of this lane's output voxel, tap 0)
    int aoff[C::AM];
#pragma unroll
    for (int a = 0; a < C::AM; ++a) {
        int m = (wm * C::AM + a) * 32 + li;
        if (m >= C::M) m = 0;                  // padding row: reads a valid voxel, never stored
        const int dz = m / (BH * BW), dy = (m / BW) % BH, dx = m % BW;
        aoff[a] = ((dz * C::HH + dy) * C::HW + dx) * C::CKP + 4 * lh;
    }
    // this lane's B fragment inside a packed [64][8] tap slice: co = (wn*AN + bb)*32 + li, ci 4*lh..
    const float* wlane = wp + (long)blockIdx.y * nchunk * 27 * C::SLICE + (wn * C::AN * 32 + li) * 8 + 4 * lh;

    // weight fragments are fetched two taps ahead of their MFMAs: global loads return in order,
    // so a fragment load issued behind the (HBM-latency) halo prefetch of the next chunk needs
    // that much slack not to stall the matrix pipe
    float4 bn1[C::AN], bn2[C::AN];
#pragma unroll
    for (int bb = 0; bb < C::AN; ++bb) {
        bn1[bb] = *reinterpret_cast<const float4*>(wlane + bb * 256);
        bn2[bb] = *reinterpret_cast<const float4*>(wlane + C::SLICE + bb * 256);
    }

    // The halo of chunk cc+1 is fetched into registers while chunk cc is being multiplied, and
    // only written to LDS between the two barriers: global-memory latency stays off the
    // critical path of the block.
    constexpr int NH = (C::HV * 2 + C::NTHR - 1) / C::NTHR;
    float4 hreg[NH];
    // chunk-invariant part of the addresses and their validity, computed once per block
    long hoff[NH];
    unsigned hmask = 0;
#pragma unroll
    for (int i = 0; i < NH; ++i) {
        const int idx = tid + C::NTHR * i;
        const int hv = idx >> 1, q = idx & 1;
        const int hx = hv % C::HW, hy = (hv / C::HW) % C::HH, hz = hv / (C::HW * C::HH);
        const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
        const bool ok = hv < C::HV && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W;
        hoff[i] = ok ? ((((long)n * Di + gz / UPS) * Hi + gy / UPS) * Wi + gx / UPS) * Cin + 4 * q : 0;
        hmask |= (unsigned)ok << i;
    }
    auto load_halo = [&](int c0) {
        const bool chok = c0 + 4 * (tid & 1) < Cin;
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!(SIV_EXP & 4) && chok && ((hmask >> i) & 1)) v = *reinterpret_cast<const float4*>(x + hoff[i] + c0);
            hreg[i] = v;
        }
    };
    load_halo(0);
    for (int cc = 0; cc < nchunk; ++cc) {
        if (!(SIV_EXP & 16)) __syncthreads();  // previous chunk's MFMAs are done with the halo
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int idx = tid + C::NTHR * i;
            if ((idx >> 1) < C::HV && (!(SIV_EXP & 32) || hreg[i].x == 123.f))
                *reinterpret_cast<float4*>(hal + (idx >> 1) * C::CKP + 4 * (idx & 1)) = hreg[i];
        }
        if (!(SIV_EXP & 16)) __syncthreads();
        if (cc + 1 < nchunk) load_halo((cc + 1) * C::CK);
        const float* wc = wlane + (long)cc * 27 * C::SLICE;
        // A fragments are read one tap ahead of their MFMAs as well, so that neither the LDS
        // nor the weight loads are waited for right after they are issued
        float4 anext[C::AM];
#pragma unroll
        for (int a = 0; a < C::AM; ++a) {
            if (SIV_EXP & 2) anext[a] = make_float4(0.f, 1.f, 2.f, 3.f);
            else anext[a] = *reinterpret_cast<const float4*>(hal + aoff[a]);
        }
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            float4 bv[C::AN], av[C::AM];
#pragma unroll
            for (int bb = 0; bb < C::AN; ++bb) {
                bv[bb] = bn1[bb];
                bn1[bb] = bn2[bb];
            }
#pragma unroll
            for (int a = 0; a < C::AM; ++a) av[a] = anext[a];
            // prefetch the fragments of tap t+2 (running into the next chunk at the end of this
            // one; the packed buffer carries two slices of padding so that this never leaves it)
#pragma unroll
            for (int bb = 0; bb < C::AN; ++bb)
                if (!(SIV_EXP & 1)) bn2[bb] = *reinterpret_cast<const float4*>(wc + (t + 2) * C::SLICE + bb * 256);
            if (t + 1 < 27) {
                const int t1 = t + 1;
                const int tapoff = (((t1 / 9) * C::HH + (t1 / 3) % 3) * C::HW + t1 % 3) * C::CKP;
#pragma unroll
                for (int a = 0; a < C::AM; ++a) {
                    if (SIV_EXP & 2) {
                        asm volatile("" : "+v"(anext[a].x), "+v"(anext[a].y), "+v"(anext[a].z), "+v"(anext[a].w));
                    } else {
                        anext[a] = *reinterpret_cast<const float4*>(hal + aoff[a] + tapoff);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // rotate through the accumulators so that consecutive MFMAs are independent
#define SIV_MFMA_STEP(F)                                                                          \
    _Pragma("unroll") for (int a = 0; a < C::AM; ++a)                                             \
    _Pragma("unroll") for (int bb = 0; bb < C::AN; ++bb)                                          \
        acc[a][bb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a].F, bv[bb].F, acc[a][bb], 0, 0, 0);
            SIV_MFMA_STEP(x)
            SIV_MFMA_STEP(y)
            SIV_MFMA_STEP(z)
            SIV_MFMA_STEP(w)
#undef SIV_MFMA_STEP
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // epilogue.  A lane holds one column (output channel) of 16 rows per accumulator; writing that
    // directly costs 16 scattered 4-byte stores per accumulator.  Instead every wave transposes
    // one 32x32 accumulator at a time through its own slice of the (now idle) halo LDS and
    // writes whole 128-byte channel runs with 16-byte stores.
    if ((Cout & 3) == 0 && !(SIV_EXP & 128)) {
        float* tile = hal + wave * C::EPI_FLOATS;
#pragma unroll
        for (int a = 0; a < C::AM; ++a) {
#pragma unroll
            for (int bb = 0; bb < C::AN; ++bb) {
                __syncthreads();   // halo reads (first pass) / previous tile's reads are done
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    tile[((r & 3) + 8 * (r >> 2) + 4 * lh) * C::EPI_STRIDE + li] = acc[a][bb][r];
                __syncthreads();
                const int cq = lane & 7;
                const int co = co0 + (wn * C::AN + bb) * 32 + 4 * cq;
                float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
                if (bias && co < Cout) bq = *reinterpret_cast<const float4*>(bias + co);
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    const int row = ps * 8 + (lane >> 3);
                    const int m = (wm * C::AM + a) * 32 + row;
                    const int gz = z0 + m / (BH * BW), gy = y0 + (m / BW) % BH, gx = x0 + m % BW;
                    float4 v = *reinterpret_cast<const float4*>(tile + row * C::EPI_STRIDE + 4 * cq);
                    v.x += bq.x; v.y += bq.y; v.z += bq.z; v.w += bq.w;
                    if (m < C::M && gz < D && gy < H && gx < W && co < Cout && (!(SIV_EXP & 8) || v.x == 123.f))
                        *reinterpret_cast<float4*>(y + ((((long)n * D + gz) * H + gy) * W + gx) * Cout + co) = v;
                }
            }
        }
        return;
    }
    // any other channel count: lane holds column co, rows (r&3)+8*(r>>2)+4*lh of each sub-tile
#pragma unroll
    for (int bb = 0; bb < C::AN; ++bb) {
        const int co = co0 + (wn * C::AN + bb) * 32 + li;
        if (co >= Cout) continue;
        const float bv = bias ? bias[co] : 0.f;
#pragma unroll
        for (int a = 0; a < C::AM; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (wm * C::AM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int gz = z0 + m / (BH * BW), gy = y0 + (m / BW) % BH, gx = x0 + m % BW;
                if (m < C::M && gz < D && gy < H && gx < W && (!(SIV_EXP & 8) || acc[a][bb][r] == 123.f))
                    y[((((long)n * D + gz) * H + gy) * W + gx) * Cout + co] = acc[a][bb][r] + bv;
            }
        }
    }
}

// Pack weights for k3_fwd_mfma: wp[ct][cc][tap][co (NT)][ci (8)], zero padded to whole tiles/chunks.
//   flip == 0 (forward):  kernel (co, ci, tap) reads w[co][tap][ci]              (w is [Kout][27][Kin])
//   flip == 1 (dgrad):    kernel (co, ci, tap) reads w[ci][26 - tap][co]         (w is [Kin][27][Kout])
// Kin / Kout are the kernel's input / output channel counts.
__global__ __launch_bounds__(256) void k3_pack_weights(const float* __restrict__ w,
                                                       float* __restrict__ wp, int Kin, int Kout,
                                                       int NT, int flip) {
    const int nchunk = (Kin + 7) / 8, ntile = (Kout + NT - 1) / NT;
    const long total = (long)ntile * nchunk * 27 * NT * 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7);
        long r = i >> 3;
        const int co = (int)(r % NT); r /= NT;
        const int tap = (int)(r % 27); r /= 27;
        const int cc = (int)(r % nchunk);
        const int ct = (int)(r / nchunk);
        const int kco = ct * NT + co, kci = cc * 8 + j;
        float v = 0.f;
        if (kco < Kout && kci < Kin)
            v = flip ? w[((long)kci * 27 + (26 - tap)) * Kout + kco] : w[((long)kco * 27 + tap) * Kin + kci];
        wp[i] = v;
    }
}

template <int BD, int BH, int BW, int WM, int WN, int NT, int UPS>
static int launch_fwd(const float* x, const float* wp, const float* bias, float* y, int N, int D,
                      int H, int W, int Cin, int Cout, hipStream_t s) {
    using C = FwdCfg<BD, BH, BW, WM, WN, NT, UPS>;
    const int nbd = (D + BD - 1) / BD, nbh = (H + BH - 1) / BH, nbw = (W + BW - 1) / BW;
    const long nblk = (long)N * nbd * nbh * nbw;
    if (nblk > 0x7fffffffL) {
        set_error("conv3d_k3(mfma): too many bricks");
        return SIV_EINVAL;
    }
    k3_fwd_mfma<BD, BH, BW, WM, WN, NT, UPS>
        <<<dim3((unsigned)nblk, (unsigned)((Cout + NT - 1) / NT)), C::NTHR, C::LDS_BYTES, s>>>(
            x, wp, bias, y, N, D, H, W, Cin, Cout, nbd, nbh, nbw);
    SIV_CHECK_LAUNCH("conv3d_k3(mfma)");
    return SIV_OK;
}

// Channel counts the matrix-core path takes: 16-byte channel groups on the input side, anything
// on the output side (tiles of 32 or 64 output channels, zero padded).
bool k3_fwd_mfma_supported(int Cin, int Cout) { return Cin % 4 == 0 && Cin >= 4 && Cout >= 8; }

// Brick / wave tiling per problem shape.
//   kBig    4x8x8 bricks, 4 waves x (2x2) accumulators: the full-resolution and 40x48x40 layers
//           (8x8x8 bricks with 4x2 accumulators per wave halve the weight-fragment loads per MFMA
//           but drop to 2 waves per SIMD: measured +1 %, not kept)
//   kSmall  2x4x8 bricks: volumes the big brick would mostly pad
//   kW10b   2x4x10 bricks (80 rows padded to 96), 3 waves x (1x1), 32-channel tiles: volumes 10
//           wide (10x12x10).  (5-wave 2x8x10 and 4x8x10 variants for 20x24x20 measured 20-25 % slower
//           than kBig with its padded third brick column -- five waves do not spread evenly over
//           four SIMDs -- and are not instantiated.)
//   kN32    4x8x8 bricks with 32-channel tiles: output widths that are not multiples of 64
// (Split-K over the input-channel chunks -- 2 or 4 blocks per output tile, partial slabs summed in a
// second pass -- was measured on the 10x12x10 layers, 240 blocks of one wave per SIMD: 0.378 ms without,
// 0.379 ms with 2 splits, 0.392 ms with 4.  Those layers stream each block's 3.5 MB weight tile from
// L2 / Infinity Cache for only 80 voxels of reuse; more resident waves do not change that.  Not kept.)
enum FwdKind { kBig, kSmall, kW10b, kN32 };
static bool use_small_brick(int D, int H, int W) {
    auto waste = [&](int bd, int bh, int bw) {
        const double full = (double)((D + bd - 1) / bd * bd) * ((H + bh - 1) / bh * bh) *
                            ((W + bw - 1) / bw * bw);
        return full / ((double)D * H * W);
    };
    return waste(2, 4, 8) * 1.15 < waste(4, 8, 8);
}
static FwdKind fwd_kind(int D, int H, int W, int Cout) {
    if (Cout % 64 != 0) return kN32;
    if (W == 10) return kW10b;
    return use_small_brick(D, H, W) ? kSmall : kBig;
}
static int fwd_tile(FwdKind k, int Cout) {
    if (k == kW10b) return Cout % 128 == 0 ? 128 : 32;
    return k == kN32 ? 32 : 64;
}
size_t k3_fwd_mfma_ws_bytes(int Cin, int Cout) {   // upper bound over both tile widths
    return ((size_t)((Cout + 31) / 32) * ((Cin + 7) / 8) * 27 * 32 * 8 + 1024) * sizeof(float);
}

// `w` is the convolution's weight tensor; flip selects forward (0) or input-gradient (1) use.
// Cin / Cout are the channel counts of the tensor the kernel reads / writes.
int k3_fwd_mfma_launch(const float* x, const float* w, const float* bias, float* y, int N, int D,
                       int H, int W, int Cin, int Cout, int ups, int flip, void* ws, size_t ws_bytes,
                       hipStream_t s) {
    SIV_CHECK_WS(k3_fwd_mfma_ws_bytes(Cin, Cout), ws_bytes, "conv3d_k3(mfma)");
    const FwdKind kind = fwd_kind(D, H, W, Cout);
    const int NT = fwd_tile(kind, Cout);
    bool need_pack;
    float* wp = (float*)pack_slot(w, PF_MFMA, NT, flip, Cin, Cout, k3_fwd_mfma_ws_bytes(Cin, Cout), ws, &need_pack);
    const long total = (long)((Cout + NT - 1) / NT) * ((Cin + 7) / 8) * 27 * NT * 8;
    long g = cdiv(total, 256);
    if (g > 4096) g = 4096;
    if (need_pack) {
        k3_pack_weights<<<(unsigned)g, 256, 0, s>>>(w, wp, Cin, Cout, NT, flip);
        SIV_CHECK_PACK_LAUNCH("conv3d_k3(pack weights)", wp);
    }
#define SIV_FWD(BD, BH, BW, WM, WN, NTT)                                                              \
    (ups == 1 ? launch_fwd<BD, BH, BW, WM, WN, NTT, 1>(x, wp, bias, y, N, D, H, W, Cin, Cout, s)       \
              : launch_fwd<BD, BH, BW, WM, WN, NTT, 2>(x, wp, bias, y, N, D, H, W, Cin, Cout, s))
    switch (kind) {
        case kN32: return SIV_FWD(4, 8, 8, 4, 1, 32);
        case kW10b: return NT == 128 ? SIV_FWD(2, 4, 10, 1, 4, 128) : SIV_FWD(2, 4, 10, 3, 1, 32);
        case kSmall: return SIV_FWD(2, 4, 8, 2, 2, 64);
        default: return SIV_FWD(4, 8, 8, 4, 1, 64);
    }
#undef SIV_FWD
}

// ---------------------------------------------------------------------------------
// wgrad
// ---------------------------------------------------------------------------------
// Block = (range of 2x4xBW-voxel bricks) x (one kd plane of taps) x (64 co x 64 ci tile).
// Wave w owns co half (w&1), ci half (w>>1) and the 9 taps (kh,kw) of the plane.
// MODE 1, 2 -- narrow layers (Cin <= 32 and Cout <= 32: the 16- and 32-channel full-resolution layers of the
// FC-latent model): all channels fit one 32 x 32 sub-tile, which would leave three of the four waves idle, so
// the waves split the brick's VOXELS instead (two of its eight rows each, a partial slab per wave); MODE 2
// (Cin <= 16): the 32 columns of a tile hold two taps x 16 input channels, five matrix instructions per k-step
// instead of nine.
// BW = 8, or 10 for volumes whose width is a multiple of 10 but not of 8 (20, 10: the two
// deepest levels of an 80x96x80 input) -- voxels are the K dimension, so padding is pure waste.
template <int UPS, int BW_>
struct WgCfg {
    static constexpr int BD = 2, BH = 4, BW = BW_;
    static constexpr int M = BD * BH * BW;       // 64 (80) voxels = 32 (40) k-steps
    static constexpr int HH = BH + 2, HW = BW + 2;
    static constexpr int HV = BD * HH * HW;      // 120 (144) halo voxels in one kd plane
    static constexpr int CP = 64;
    static constexpr size_t LDS_BYTES = (size_t)(M * CP + HV * CP) * sizeof(float);
};

// Global -> register stage of one brick (issued one brick ahead of its use), then register -> LDS.
template <int UPS, int BW_>
struct WgStage {
    using C = WgCfg<UPS, BW_>;
    static constexpr int NDY = C::M * 16 / 256;                 // 4 (5) float4 of dy per thread
    static constexpr int NX = (C::HV * 16 + 255) / 256;         // 8 (9) float4 of x per thread (last partial)
    float4 dyr[NDY];
    float4 xr[NX];

    __device__ __forceinline__ void load(const float* __restrict__ x, const float* __restrict__ dy,
                                         long br, int kd, int co0, int ci0, int N, int D, int H, int W,
                                         int Cin, int Cout, int nbd, int nbh, int nbw) {
        const int tid = threadIdx.x;
        const int Di = D / UPS, Hi = H / UPS, Wi = W / UPS;
        long b = br;
        const int bx = (int)(b % nbw); b /= nbw;
        const int by = (int)(b % nbh); b /= nbh;
        const int bz = (int)(b % nbd);
        const int n = (int)(b / nbd);
        const int z0 = bz * C::BD, y0 = by * C::BH, x0 = bx * C::BW;
#pragma unroll
        for (int i = 0; i < NDY; ++i) {
            const int idx = tid + 256 * i;
            const int m = idx >> 4, q = idx & 15;
            const int gz = z0 + m / (C::BH * C::BW), gy = y0 + (m / C::BW) % C::BH, gx = x0 + m % C::BW;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gz < D && gy < H && gx < W && co0 + 4 * q < Cout)
                v = *reinterpret_cast<const float4*>(
                    dy + ((((long)n * D + gz) * H + gy) * W + gx) * Cout + co0 + 4 * q);
            dyr[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int idx = tid + 256 * i;
            const int hv = idx >> 4, q = idx & 15;
            const int hx = hv % C::HW, hy = (hv / C::HW) % C::HH, hz = hv / (C::HW * C::HH);
            const int gz = z0 + hz + kd - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (hv < C::HV && ci0 + 4 * q < Cin && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 &&
                gx < W)
                v = *reinterpret_cast<const float4*>(
                    x + ((((long)n * Di + gz / UPS) * Hi + gy / UPS) * Wi + gx / UPS) * Cin + ci0 + 4 * q);
            xr[i] = v;
        }
    }
    __device__ __forceinline__ void store(float* dyl, float* xl) const {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < NDY; ++i) {
            const int idx = tid + 256 * i;
            *reinterpret_cast<float4*>(dyl + (idx >> 4) * C::CP + 4 * (idx & 15)) = dyr[i];
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int idx = tid + 256 * i;
            if ((idx >> 4) < C::HV)
                *reinterpret_cast<float4*>(xl + (idx >> 4) * C::CP + 4 * (idx & 15)) = xr[i];
        }
    }
};

template <int UPS, int BW_, int MODE>
__global__ __launch_bounds__(256, 2) void k3_wgrad_mfma(const float* __restrict__ x,
                                                        const float* __restrict__ dy,
                                                        float* __restrict__ part, int N, int D, int H,
                                                        int W, int Cin, int Cout, int nbd, int nbh,
                                                        int nbw, int bricks_per_block) {
    using C = WgCfg<UPS, BW_>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* dyl = lds;                 // [64 voxels][64 co]
    float* xl = lds + C::M * C::CP;   // [120 halo voxels][64 ci]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int cs = MODE ? 0 : wave & 1, is = MODE ? 0 : wave >> 1;
    const int kd = blockIdx.y;
    const int ntci = (Cin + 63) / 64;
    const int co0 = (blockIdx.z / ntci) * 64, ci0 = (blockIdx.z % ntci) * 64;
    // a wave whose 32x32 (co, ci) sub-tile lies outside the channel ranges only helps staging
    const bool active = co0 + cs * 32 < Cout && ci0 + is * 32 < Cin;
    const long nbricks = (long)N * nbd * nbh * nbw;
    const long b0 = (long)blockIdx.x * bricks_per_block;
    long b1 = b0 + bricks_per_block;
    if (b1 > nbricks) b1 = nbricks;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    WgStage<UPS, BW_> st;
    if (b0 < b1) st.load(x, dy, b0, kd, co0, ci0, N, D, H, W, Cin, Cout, nbd, nbh, nbw);
    for (long br = b0; br < b1; ++br) {
        __syncthreads();              // the previous brick's MFMAs are done with LDS
        st.store(dyl, xl);
        __syncthreads();
        if (br + 1 < b1)              // next brick's global loads fly during this brick's MFMAs
            st.load(x, dy, br + 1, kd, co0, ci0, N, D, H, W, Cin, Cout, nbd, nbh, nbw);
        const float* ap = dyl + lh * C::CP + cs * 32 + li;
        const float* bp = xl + lh * C::CP + is * 32 + li;
        if constexpr (MODE != 0) {
            // this wave's rows 2 wave, 2 wave + 1 of the brick's BD x BH rows: BW / 2 k-steps each
            constexpr int HB = C::BW / 2;
            static_assert(C::BD * C::BH == 8, "two brick rows per wave");
            const int row = 2 * wave;
            const float* aw = ap + 2 * (row * HB) * C::CP;
            const int wbase = ((row / C::BH) * C::HH + row % C::BH) * C::HW;
            if constexpr (MODE == 2) {
                // column li = (tap 2p + (li >> 4), input channel li & 15) of tile p; the odd half of tile 4 has no tap
                const float* bq[5];
#pragma unroll
                for (int p2 = 0; p2 < 5; ++p2) {
                    const int t = 2 * p2 + (li >> 4) < 9 ? 2 * p2 + (li >> 4) : 8;
                    bq[p2] = xl + lh * C::CP + (li & 15) + (wbase + (t / 3) * C::HW + (t % 3)) * C::CP;
                }
#pragma unroll
                for (int j = 0; j < 2 * HB; ++j) {
                    const int off = ((j / HB) * C::HW + 2 * (j % HB)) * C::CP;
                    const float a = aw[2 * j * C::CP];
#pragma unroll
                    for (int p2 = 0; p2 < 5; ++p2)
                        acc[p2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq[p2][off], acc[p2], 0, 0, 0);
                }
            } else {
                const float* bw = bp + wbase * C::CP;
#pragma unroll 1
                for (int jr = 0; jr < 2; ++jr) {                  // (one brick row at a time: register budget)
#pragma unroll
                    for (int jc = 0; jc < HB; ++jc) {
                        const float a = aw[2 * (jr * HB + jc) * C::CP];
#pragma unroll
                        for (int t = 0; t < 9; ++t)
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                a, bw[(jr * C::HW + 2 * jc + (t / 3) * C::HW + (t % 3)) * C::CP], acc[t], 0, 0, 0);
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
            for (int t = 0; t < 9; ++t) {
                const float bvv = bp[(hbase + (t / 3) * C::HW + (t % 3)) * C::CP];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bvv, acc[t], 0, 0, 0);
            }
        }
    }

    // partial slab: part[blockIdx.x][co][tap][ci]; narrow layers: one slab per wave, part[blockIdx.x * 4 + wave]
    float* pp = part + (MODE ? (long)blockIdx.x * 4 + wave : (long)blockIdx.x) * Cout * 27 * Cin;
    if constexpr (MODE == 2) {
#pragma unroll
        for (int p2 = 0; p2 < 5; ++p2) {
            const int t = 2 * p2 + (li >> 4), ci = li & 15;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (co < Cout && t < 9 && ci < Cin) pp[((long)co * 27 + kd * 9 + t) * Cin + ci] = acc[p2][r];
            }
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + cs * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int ci = ci0 + is * 32 + li;
            if (co < Cout && ci < Cin) pp[((long)co * 27 + kd * 9 + t) * Cin + ci] = acc[t][r];
        }
    }
}

// (A transposed-Winograd F(2,3) variant of this kernel -- 4 position sums per filter row instead of 3
// taps, 12 accumulators = 192 registers per wave -- was built and measured on 64->64@80x96x80, where
// this kernel takes 8.5 ms: at 2 waves per SIMD it spills and loses its register-prefetched staging
// (11.1 ms); at 1 wave per SIMD, with double-buffered LDS and software-pipelined operand reads, it
// reaches 8.2-8.3 ms: a single wave cannot hide the ~600 VALU instructions of staging index arithmetic
// per brick, and keeping that arithmetic in brick-invariant registers pushed the compiler into
// scratch.  Not kept.)

// dw[e] = (accumulate ? dw[e] : 0) + sum_s part[s][e], fixed order
__global__ __launch_bounds__(256) void slab_reduce(const float* __restrict__ part, int nslab, long n,
                                                   float* __restrict__ out, int accumulate) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long)gridDim.x * blockDim.x) {
        // sixteen slabs in flight (a thread's loads are independent; added in slab order as before): with
        // hundreds of slabs of a small tensor the loop is bound by load latency, not by bandwidth
        float s = 0.f;
        int k = 0;
        for (; k + 16 <= nslab; k += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = part[(long)(k + u) * n + i];
#pragma unroll
            for (int u = 0; u < 16; ++u) s += v[u];
        }
        for (; k < nslab; ++k) s += part[(long)k * n + i];
        out[i] = accumulate ? out[i] + s : s;
    }
}

bool k3_wgrad_mfma_supported(int Cin, int Cout) {
    return Cin % 4 == 0 && Cout % 4 == 0 && Cin >= 4 && Cout >= 8;
}

int k3_wgrad_mfma_launch(const float* x, const float* dy, float* dw, int N, int D, int H, int W,
                         int Cin, int Cout, int ups, int accumulate, void* ws, size_t ws_bytes,
                         hipStream_t s) {
    const bool wide = (W % 10 == 0) && (W % 8 != 0);
    const int nbd = (D + 1) / 2, nbh = (H + 3) / 4, nbw = wide ? W / 10 : (W + 7) / 8;
    const long nbricks = (long)N * nbd * nbh * nbw;
    const int tiles = ((Cout + 63) / 64) * ((Cin + 63) / 64);
    // enough blocks to fill the chip twice over, bounded by the workspace
    const long wsize = (long)Cout * 27 * Cin;
    // 3*tiles*nsplit blocks at 2 resident blocks per CU: stay at or below a whole number of
    // rounds (1024 slots = 2 rounds) so that no round runs nearly empty
    long nsplit = 1024 / (3L * tiles);
    if (nsplit < 1) nsplit = 1;
    if (nsplit > nbricks) nsplit = nbricks;
    const int mode = (Cin <= 32 && Cout <= 32) ? (Cin <= 16 ? 2 : 1) : 0;
    const int wave_slabs = mode ? 4 : 1;                            // narrow layers: a partial slab per wave
    const long max_by_ws = (long)(ws_bytes / (wsize * sizeof(float))) / wave_slabs;
    if (nsplit > max_by_ws) nsplit = max_by_ws;
    if (nsplit < 1) {
        set_error("conv3d_k3_wgrad: workspace too small (need %zu, have %zu)",
                  (size_t)wsize * sizeof(float), ws_bytes);
        return SIV_EWORKSPACE;
    }
    const int bpb = (int)cdiv(nbricks, nsplit);
    nsplit = cdiv(nbricks, bpb);
    dim3 grid((unsigned)nsplit, 3, (unsigned)tiles);
#define SIV_WGM(U, BWW, MD)                                                                         \
    k3_wgrad_mfma<U, BWW, MD><<<grid, 256, WgCfg<U, BWW>::LDS_BYTES, s>>>(x, dy, (float*)ws, N, D, H, W, Cin, \
                                                                          Cout, nbd, nbh, nbw, bpb)
#define SIV_WG(U, BWW)                                                                              \
    do {                                                                                            \
        if (mode == 2) SIV_WGM(U, BWW, 2); else if (mode == 1) SIV_WGM(U, BWW, 1); else SIV_WGM(U, BWW, 0); \
    } while (0)
    if (ups == 1) {
        if (wide) SIV_WG(1, 10); else SIV_WG(1, 8);
    } else {
        if (wide) SIV_WG(2, 10); else SIV_WG(2, 8);
    }
#undef SIV_WG
#undef SIV_WGM
    SIV_CHECK_LAUNCH("conv3d_k3_wgrad(mfma)");
    long g = cdiv(wsize, 256);
    if (g > 4096) g = 4096;
    slab_reduce<<<(unsigned)g, 256, 0, s>>>((const float*)ws, (int)nsplit * wave_slabs, wsize, dw, accumulate);
    SIV_CHECK_LAUNCH("conv3d_k3_wgrad(reduce)");
    return SIV_OK;
}

}  // namespace siv
