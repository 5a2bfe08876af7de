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

namespace siv {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------
// forward (and dgrad through transformed weights)
// ---------------------------------------------------------------------------------
template <int BD, int BH, int BW, int WM, int WN, int CK, int UPS>
struct FwdCfg {
    static constexpr int M = BD * BH * BW;
    static constexpr int MS = M / 32;          // 32-row sub-tiles in the brick
    static constexpr int NS = 2;               // 64 output channels per block
    static constexpr int AM = MS / WM;         // accumulators per wave along M
    static constexpr int AN = NS / WN;
    static constexpr int HD = BD + 2, HH = BH + 2, HW = BW + 2;
    static constexpr int HV = HD * HH * HW;
    static constexpr int CKP = CK + 4;         // padded channel stride (keeps 16-B alignment)
    static constexpr int HAL_FLOATS = HV * CKP;
    static constexpr int WL_FLOATS = 9 * 64 * CKP;
    static constexpr size_t LDS_BYTES = (size_t)(HAL_FLOATS + WL_FLOATS) * sizeof(float);
    static_assert(M % 32 == 0 && MS % WM == 0 && NS % WN == 0 && WM * WN == 4, "bad tiling");
    static_assert(CK % 8 == 0, "CK must be a multiple of 8");
};

template <int BD, int BH, int BW, int WM, int WN, int CK, int UPS>
__global__ __launch_bounds__(256) void k3_fwd_mfma(const float* __restrict__ x,
                                                   const float* __restrict__ w,
                                                   const float* __restrict__ bias,
                                                   float* __restrict__ y, int N, int D, int H, int W,
                                                   int Cin, int Cout, int nbd, int nbh, int nbw) {
    using C = FwdCfg<BD, BH, BW, WM, WN, CK, UPS>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* hal = lds;
    float* wl = lds + C::HAL_FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave % WM, wn = wave / WM;

    int b = blockIdx.x;
    const int bx = b % nbw; b /= nbw;
    const int by = b % nbh; b /= nbh;
    const int bz = b % nbd;
    const int n = b / nbd;
    const int z0 = bz * BD, y0 = by * BH, x0 = bx * BW;
    const int co0 = blockIdx.y * 64;
    const int Di = D / UPS, Hi = H / UPS, Wi = W / UPS;

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
        const int m = (wm * C::AM + a) * 32 + li;
        const int dz = m / (BH * BW), dy = (m / BW) % BH, dx = m % BW;
        aoff[a] = ((dz * C::HH + dy) * C::HW + dx) * C::CKP + 4 * lh;
    }
    int boff[C::AN];
#pragma unroll
    for (int bb = 0; bb < C::AN; ++bb) boff[bb] = ((wn * C::AN + bb) * 32 + li) * C::CKP + 4 * lh;

    constexpr int Q = CK / 4;  // float4 per voxel per chunk
    for (int c0 = 0; c0 < Cin; c0 += CK) {
        __syncthreads();  // previous chunk's MFMAs are done with hal/wl
        for (int idx = tid; idx < C::HV * Q; idx += 256) {
            const int hv = idx / Q, q = idx % Q;
            const int hx = hv % C::HW, hy = (hv / C::HW) % C::HH, hz = hv / (C::HW * C::HH);
            const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W) {
                const long off =
                    ((((long)n * Di + gz / UPS) * Hi + gy / UPS) * Wi + gx / UPS) * Cin + c0 + 4 * q;
                v = *reinterpret_cast<const float4*>(x + off);
            }
            *reinterpret_cast<float4*>(hal + hv * C::CKP + 4 * q) = v;
        }
        for (int kd = 0; kd < 3; ++kd) {
            if (kd) __syncthreads();  // previous kd's MFMAs are done with wl
            for (int idx = tid; idx < 9 * 64 * Q; idx += 256) {
                const int q = idx % Q, co = (idx / Q) % 64, t = idx / (Q * 64);
                const long off = ((long)(co0 + co) * 27 + kd * 9 + t) * Cin + c0 + 4 * q;
                *reinterpret_cast<float4*>(wl + (t * 64 + co) * C::CKP + 4 * q) =
                    *reinterpret_cast<const float4*>(w + off);
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int kh = t / 3, kw = t % 3;
                const int tapoff = ((kd * C::HH + kh) * C::HW + kw) * C::CKP;
#pragma unroll
                for (int g = 0; g < CK / 8; ++g) {
                    float4 av[C::AM], bv[C::AN];
#pragma unroll
                    for (int a = 0; a < C::AM; ++a)
                        av[a] = *reinterpret_cast<const float4*>(hal + aoff[a] + tapoff + 8 * g);
#pragma unroll
                    for (int bb = 0; bb < C::AN; ++bb)
                        bv[bb] = *reinterpret_cast<const float4*>(wl + boff[bb] + t * 64 * C::CKP + 8 * g);
#pragma unroll
                    for (int a = 0; a < C::AM; ++a)
#pragma unroll
                        for (int bb = 0; bb < C::AN; ++bb) {
                            acc[a][bb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a].x, bv[bb].x, acc[a][bb], 0, 0, 0);
                            acc[a][bb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a].y, bv[bb].y, acc[a][bb], 0, 0, 0);
                            acc[a][bb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a].z, bv[bb].z, acc[a][bb], 0, 0, 0);
                            acc[a][bb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a].w, bv[bb].w, acc[a][bb], 0, 0, 0);
                        }
                }
            }
        }
    }

    // epilogue: lane holds column co, rows (r&3)+8*(r>>2)+4*lh of each sub-tile
#pragma unroll
    for (int bb = 0; bb < C::AN; ++bb) {
        const int co = co0 + (wn * C::AN + bb) * 32 + li;
        const float bv = bias ? bias[co] : 0.f;
#pragma unroll
        for (int a = 0; a < C::AM; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (wm * C::AM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int gz = z0 + m / (BH * BW), gy = y0 + (m / BW) % BH, gx = x0 + m % BW;
                if (gz < D && gy < H && gx < W)
                    y[((((long)n * D + gz) * H + gy) * W + gx) * Cout + co] = acc[a][bb][r] + bv;
            }
        }
    }
}

// wt[ci][27 - 1 - tap][co] = w[co][tap][ci]: the weights of the convolution that maps dy to dx
__global__ __launch_bounds__(256) void k3_flip_transpose(const float* __restrict__ w,
                                                         float* __restrict__ wt, int Cin, int Cout) {
    const long total = (long)Cout * 27 * Cin;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const int co = (int)(i % Cout);
        const int tap = (int)((i / Cout) % 27);
        const int ci = (int)(i / ((long)Cout * 27));
        wt[i] = w[((long)co * 27 + (26 - tap)) * Cin + ci];
    }
}

template <int BD, int BH, int BW, int WM, int WN, int CK, int UPS>
static int launch_fwd(const float* x, const float* w, const float* bias, float* y, int N, int D,
                      int H, int W, int Cin, int Cout, hipStream_t s) {
    using C = FwdCfg<BD, BH, BW, WM, WN, CK, UPS>;
    const int nbd = (D + BD - 1) / BD, nbh = (H + BH - 1) / BH, nbw = (W + BW - 1) / BW;
    const long nblk = (long)N * nbd * nbh * nbw;
    if (nblk > 0x7fffffffL) {
        set_error("conv3d_k3(mfma): too many bricks");
        return SIV_EINVAL;
    }
    auto kern = k3_fwd_mfma<BD, BH, BW, WM, WN, CK, UPS>;
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)C::LDS_BYTES);
        attr_done = true;
    }
    kern<<<dim3((unsigned)nblk, (unsigned)(Cout / 64)), 256, C::LDS_BYTES, s>>>(
        x, w, bias, y, N, D, H, W, Cin, Cout, nbd, nbh, nbw);
    SIV_CHECK_LAUNCH("conv3d_k3(mfma)");
    return SIV_OK;
}

bool k3_fwd_mfma_supported(int Cin, int Cout) { return Cin % 8 == 0 && Cout % 64 == 0; }

// Volumes whose bricks would be mostly padding use the small brick.
static bool use_small_brick(int D, int H, int W) {
    auto waste = [&](int bd, int bh, int bw) {
        const double full = (double)((D + bd - 1) / bd * bd) * ((H + bh - 1) / bh * bh) *
                            ((W + bw - 1) / bw * bw);
        return full / ((double)D * H * W);
    };
    return waste(2, 4, 8) * 1.15 < waste(4, 8, 8);
}

int k3_fwd_mfma_launch(const float* x, const float* w, const float* bias, float* y, int N, int D,
                       int H, int W, int Cin, int Cout, int ups, hipStream_t s) {
    const bool small = use_small_brick(D, H, W);
    if (ups == 1) {
        return small ? launch_fwd<2, 4, 8, 2, 2, 8, 1>(x, w, bias, y, N, D, H, W, Cin, Cout, s)
                     : launch_fwd<4, 8, 8, 4, 1, 8, 1>(x, w, bias, y, N, D, H, W, Cin, Cout, s);
    }
    return small ? launch_fwd<2, 4, 8, 2, 2, 8, 2>(x, w, bias, y, N, D, H, W, Cin, Cout, s)
                 : launch_fwd<4, 8, 8, 4, 1, 8, 2>(x, w, bias, y, N, D, H, W, Cin, Cout, s);
}

int k3_flip_transpose_launch(const float* w, float* wt, int Cin, int Cout, hipStream_t s) {
    const long total = (long)Cout * 27 * Cin;
    long g = cdiv(total, 256);
    if (g > 4096) g = 4096;
    k3_flip_transpose<<<(unsigned)g, 256, 0, s>>>(w, wt, Cin, Cout);
    SIV_CHECK_LAUNCH("conv3d_k3_dgrad(weights)");
    return SIV_OK;
}

// ---------------------------------------------------------------------------------
// wgrad
// ---------------------------------------------------------------------------------
// Block = (range of 2x4x8-voxel bricks) x (one kd plane of taps) x (64 co x 64 ci tile).
// Wave w owns co half (w&1), ci half (w>>1) and the 9 taps (kh,kw) of the plane.
template <int UPS>
struct WgCfg {
    static constexpr int BD = 2, BH = 4, BW = 8;
    static constexpr int M = BD * BH * BW;       // 64 voxels = 32 k-steps
    static constexpr int HH = BH + 2, HW = BW + 2;
    static constexpr int HV = BD * HH * HW;      // 120 halo voxels in one kd plane
    static constexpr int CP = 64;
    static constexpr size_t LDS_BYTES = (size_t)(M * CP + HV * CP) * sizeof(float);
};

template <int UPS>
__global__ __launch_bounds__(256) void k3_wgrad_mfma(const float* __restrict__ x,
                                                     const float* __restrict__ dy,
                                                     float* __restrict__ part, int N, int D, int H,
                                                     int W, int Cin, int Cout, int nbd, int nbh,
                                                     int nbw, int bricks_per_block) {
    using C = WgCfg<UPS>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* dyl = lds;                 // [64 voxels][64 co]
    float* xl = lds + C::M * C::CP;   // [120 halo voxels][64 ci]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int cs = wave & 1, is = wave >> 1;
    const int kd = blockIdx.y;
    const int ntci = Cin / 64;
    const int co0 = (blockIdx.z / ntci) * 64, ci0 = (blockIdx.z % ntci) * 64;
    const int Di = D / UPS, Hi = H / UPS, Wi = W / UPS;
    const long nbricks = (long)N * nbd * nbh * nbw;
    const long b0 = (long)blockIdx.x * bricks_per_block;
    long b1 = b0 + bricks_per_block;
    if (b1 > nbricks) b1 = nbricks;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    for (long br = b0; br < b1; ++br) {
        long b = br;
        const int bx = (int)(b % nbw); b /= nbw;
        const int by = (int)(b % nbh); b /= nbh;
        const int bz = (int)(b % nbd);
        const int n = (int)(b / nbd);
        const int z0 = bz * C::BD, y0 = by * C::BH, x0 = bx * C::BW;
        __syncthreads();
        // dy tile: 64 voxels x 64 channels = 1024 float4
        for (int idx = tid; idx < C::M * 16; idx += 256) {
            const int m = idx >> 4, q = idx & 15;
            const int gz = z0 + (m >> 5), gy = y0 + ((m >> 3) & 3), gx = x0 + (m & 7);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gz < D && gy < H && gx < W)
                v = *reinterpret_cast<const float4*>(
                    dy + ((((long)n * D + gz) * H + gy) * W + gx) * Cout + co0 + 4 * q);
            *reinterpret_cast<float4*>(dyl + m * C::CP + 4 * q) = v;
        }
        // x halo plane for this kd: z in [z0+kd-1, z0+kd-1+BD), y/x with a 1-voxel rim
        for (int idx = tid; idx < C::HV * 16; idx += 256) {
            const int hv = idx >> 4, q = idx & 15;
            const int hx = hv % C::HW, hy = (hv / C::HW) % C::HH, hz = hv / (C::HW * C::HH);
            const int gz = z0 + hz + kd - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W)
                v = *reinterpret_cast<const float4*>(
                    x + ((((long)n * Di + gz / UPS) * Hi + gy / UPS) * Wi + gx / UPS) * Cin + ci0 + 4 * q);
            *reinterpret_cast<float4*>(xl + hv * C::CP + 4 * q) = v;
        }
        __syncthreads();
        const float* ap = dyl + lh * C::CP + cs * 32 + li;
        const float* bp = xl + lh * C::CP + is * 32 + li;
#pragma unroll 4
        for (int kk = 0; kk < 32; ++kk) {
            // voxels 2kk + lh: dx = 2*(kk&3) + lh, dy = (kk>>2)&3, dz = kk>>4
            const int hbase = ((kk >> 4) * C::HH + ((kk >> 2) & 3)) * C::HW + 2 * (kk & 3);
            const float a = ap[2 * kk * C::CP];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float bvv = bp[(hbase + (t / 3) * C::HW + (t % 3)) * C::CP];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bvv, acc[t], 0, 0, 0);
            }
        }
    }

    // partial slab: part[blockIdx.x][co][tap][ci]
    float* pp = part + (long)blockIdx.x * Cout * 27 * Cin;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + cs * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int ci = ci0 + is * 32 + li;
            pp[((long)co * 27 + kd * 9 + t) * Cin + ci] = acc[t][r];
        }
    }
}

// dw[e] = (accumulate ? dw[e] : 0) + sum_s part[s][e], fixed order
__global__ __launch_bounds__(256) void slab_reduce(const float* __restrict__ part, int nslab, long n,
                                                   float* __restrict__ out, int accumulate) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < nslab; ++k) s += part[(long)k * n + i];
        out[i] = accumulate ? out[i] + s : s;
    }
}

bool k3_wgrad_mfma_supported(int Cin, int Cout) { return Cin % 64 == 0 && Cout % 64 == 0; }

int k3_wgrad_mfma_launch(const float* x, const float* dy, float* dw, int N, int D, int H, int W,
                         int Cin, int Cout, int ups, int accumulate, void* ws, size_t ws_bytes,
                         hipStream_t s) {
    const int nbd = (D + 1) / 2, nbh = (H + 3) / 4, nbw = (W + 7) / 8;
    const long nbricks = (long)N * nbd * nbh * nbw;
    const int tiles = (Cout / 64) * (Cin / 64);
    // enough blocks to fill the chip twice over, bounded by the workspace
    const long wsize = (long)Cout * 27 * Cin;
    long nsplit = cdiv(1024, 3L * tiles);
    if (nsplit > nbricks) nsplit = nbricks;
    const long max_by_ws = (long)(ws_bytes / (wsize * sizeof(float)));
    if (nsplit > max_by_ws) nsplit = max_by_ws;
    if (nsplit < 1) {
        set_error("conv3d_k3_wgrad: workspace too small (need %zu, have %zu)",
                  (size_t)wsize * sizeof(float), ws_bytes);
        return SIV_EWORKSPACE;
    }
    const int bpb = (int)cdiv(nbricks, nsplit);
    nsplit = cdiv(nbricks, bpb);
    dim3 grid((unsigned)nsplit, 3, (unsigned)tiles);
    if (ups == 1) {
        auto kern = k3_wgrad_mfma<1>;
        kern<<<grid, 256, WgCfg<1>::LDS_BYTES, s>>>(x, dy, (float*)ws, N, D, H, W, Cin, Cout, nbd, nbh,
                                                    nbw, bpb);
    } else {
        auto kern = k3_wgrad_mfma<2>;
        kern<<<grid, 256, WgCfg<2>::LDS_BYTES, s>>>(x, dy, (float*)ws, N, D, H, W, Cin, Cout, nbd, nbh,
                                                    nbw, bpb);
    }
    SIV_CHECK_LAUNCH("conv3d_k3_wgrad(mfma)");
    long g = cdiv(wsize, 256);
    if (g > 4096) g = 4096;
    slab_reduce<<<(unsigned)g, 256, 0, s>>>((const float*)ws, (int)nsplit, wsize, dw, accumulate);
    SIV_CHECK_LAUNCH("conv3d_k3_wgrad(reduce)");
    return SIV_OK;
}

}  // namespace siv
