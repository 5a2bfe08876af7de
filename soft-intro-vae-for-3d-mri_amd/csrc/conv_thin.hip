// 3x3x3 convolutions with one channel on one side: the encoder stem (1 -> C), the decoder
// output layer (C -> 1), their input gradients and their weight gradients.  These layers
// move a full-resolution C-channel tensor exactly once and do little arithmetic: they are
// HBM bound, so the kernels are organised around one coalesced pass over the wide tensor
// with the single-channel side held as a halo brick in LDS.
//
//   one_to_c : out[v][c] = b[c] + sum_tap a[v+tap-1] * wk[c][tap]          (stem fwd, tail dgrad)
//   c_to_one : out[v]    = b + sum_tap sum_c a[v+tap-1][c] * wk[tap][c]    (tail fwd, stem dgrad)
//   thin_wg  : T[c][tap] = sum_u a[u+tap-1] * g[u][c]                      (both weight gradients)
#include "siv_common.h"

namespace siv {

constexpr int THIN_CUS = 256;                         // MI355X
constexpr int TB_D = 4, TB_H = 8, TB_W = 8;           // output brick
constexpr int TH_D = 6, TH_H = 10, TH_W = 10;         // halo brick
constexpr int TB_V = TB_D * TB_H * TB_W;              // 256
constexpr int TH_V = TH_D * TH_H * TH_W;              // 600

__device__ __forceinline__ void brick_coords(int b, int nbd, int nbh, int nbw, int& n, int& z0,
                                             int& y0, int& x0) {
    const int bx = b % nbw; b /= nbw;
    const int by = b % nbh; b /= nbh;
    const int bz = b % nbd;
    n = b / nbd;
    z0 = bz * TB_D; y0 = by * TB_H; x0 = bx * TB_W;
}

// Load the single-channel halo brick (zero padded) into LDS.
__device__ __forceinline__ void load_halo1(const float* __restrict__ a, float* hal, int n, int z0,
                                           int y0, int x0, int D, int H, int W) {
    for (int hv = threadIdx.x; hv < TH_V; hv += 256) {
        const int hx = hv % TH_W, hy = (hv / TH_W) % TH_H, hz = hv / (TH_W * TH_H);
        const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
        float v = 0.f;
        if (gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W)
            v = a[(((long)n * D + gz) * H + gy) * W + gx];
        hal[hv] = v;
    }
}

// The same in two halves for kernels that walk several bricks: the halo of brick b + 1 is fetched into
// registers while brick b is being worked on and put into LDS at the top of the next round, so its
// global-memory latency is not paid between two barriers.
constexpr int TH_PER_THREAD = (TH_V + 255) / 256;      // 3
__device__ __forceinline__ void halo1_fetch(const float* __restrict__ a, float (&hr)[TH_PER_THREAD], int br,
                                            int nbd, int nbh, int nbw, int D, int H, int W) {
    int n, z0, y0, x0;
    brick_coords(br, nbd, nbh, nbw, n, z0, y0, x0);
#pragma unroll
    for (int i = 0; i < TH_PER_THREAD; ++i) {
        const int hv = threadIdx.x + 256 * i;
        const int hx = hv % TH_W, hy = (hv / TH_W) % TH_H, hz = hv / (TH_W * TH_H);
        const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
        const bool ok = hv < TH_V && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W;
        const float v = a[ok ? (((long)n * D + gz) * H + gy) * W + gx : 0];     // clamped, unconditional
        hr[i] = ok ? v : 0.f;
    }
}
__device__ __forceinline__ void halo1_put(float* hal, const float (&hr)[TH_PER_THREAD]) {
#pragma unroll
    for (int i = 0; i < TH_PER_THREAD; ++i) {
        const int hv = threadIdx.x + 256 * i;
        if (hv < TH_V) hal[hv] = hr[i];
    }
}

// ---------------------------------------------------------------------------------
// one_to_c.  Thread = (voxel lane, channel quad); the quad's 27x4 weights live in registers.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void one_to_c_k(const float* __restrict__ a,
                                                  const float* __restrict__ w,
                                                  const float* __restrict__ bias,
                                                  float* __restrict__ out, int N, int D, int H, int W,
                                                  int C, int flip, int nbd, int nbh, int nbw,
                                                  int bricks_per_block, int nbricks) {
    __shared__ float hal[TH_V];
    const int QC = C >> 2;
    const int lanes = 256 / QC;
    const int q = threadIdx.x % QC, vl = threadIdx.x / QC;
    const bool active = vl < lanes;
    // the weights are loaded once and reused for every brick of the block
    float4 wr[27];
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) {
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            if (!flip) {  // w is [C][27][1]
                wr[t] = make_float4(w[(4 * q + 0) * 27 + t], w[(4 * q + 1) * 27 + t],
                                    w[(4 * q + 2) * 27 + t], w[(4 * q + 3) * 27 + t]);
            } else {      // w is [1][27][C], taps reversed
                wr[t] = *reinterpret_cast<const float4*>(w + (26 - t) * C + 4 * q);
            }
        }
        if (bias) bv = *reinterpret_cast<const float4*>(bias + 4 * q);
    }
    const int b0 = blockIdx.x * bricks_per_block;
    const int b1 = b0 + bricks_per_block < nbricks ? b0 + bricks_per_block : nbricks;
    for (int br = b0; br < b1; ++br) {
        int n, z0, y0, x0;
        brick_coords(br, nbd, nbh, nbw, n, z0, y0, x0);
        __syncthreads();
        load_halo1(a, hal, n, z0, y0, x0, D, H, W);
        __syncthreads();
        if (!active) continue;
        for (int m = vl; m < TB_V; m += lanes) {
            const int dz = m >> 6, dy = (m >> 3) & 7, dx = m & 7;
            const int gz = z0 + dz, gy = y0 + dy, gx = x0 + dx;
            if (gz >= D || gy >= H || gx >= W) continue;
            const float* hp = hal + (dz * TH_H + dy) * TH_W + dx;
            float4 acc = bv;
#pragma unroll
            for (int t = 0; t < 27; ++t) {
                const float av = hp[((t / 9) * TH_H + (t / 3) % 3) * TH_W + t % 3];
                acc.x = fmaf(av, wr[t].x, acc.x);
                acc.y = fmaf(av, wr[t].y, acc.y);
                acc.z = fmaf(av, wr[t].z, acc.z);
                acc.w = fmaf(av, wr[t].w, acc.w);
            }
            *reinterpret_cast<float4*>(out + ((((long)n * D + gz) * H + gy) * W + gx) * C + 4 * q) = acc;
        }
    }
}

// ---------------------------------------------------------------------------------
// one_to_c on the matrix pipe (C <= 64): out = im2col(a) · Wk^T, rows = voxels, K = the 27 taps padded
// to 28 = seven steps of v_mfma_f32_16x16x4_f32.  Four instructions per step: instruction j owns
// the channels 4 q + j as its 16 columns, so that a lane (column q) ends up with the four consecutive
// channels 4 q .. 4 q + 3 of a voxel and the 16 lanes of a row group write one voxel's 256 bytes as one
// contiguous run.  The weights are the B operands and stay in registers (lane (q, kk): channel
// 4 q + j, tap 4 s + kk); the A operand of a step is the halo value under tap 4 s + kk of voxel
// (lane & 15): seven LDS reads feed the 28 instructions of a 16-voxel tile.  Wave wv owns plane wv of
// every 4x8x8 brick.  The VALU kernel above issues 27 LDS reads and 108 multiply-adds per float4 of
// output.
// ---------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16t __attribute__((ext_vector_type(16)));

// Four consecutive channels of a voxel held as bf16 (8 bytes): the storage format of activations that are
// kept for the backward pass when bf16 activation storage is on.  `e` is the ELEMENT index of the first channel.
__device__ __forceinline__ void thin_st_bf4(uint16_t* __restrict__ p, long e, float4 v) {
    *reinterpret_cast<uint2*>(p + e) = make_uint2(bf16_rne(v.x) | (bf16_rne(v.y) << 16), bf16_rne(v.z) | (bf16_rne(v.w) << 16));
}
// a float4 of four channels from an fp32 tensor, or (BF) from its bf16 copy
template <bool BF>
__device__ __forceinline__ float4 thin_ld4(const float* __restrict__ p, long e) {
    if constexpr (BF) {
        const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(p) + e);
        return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                           __uint_as_float(u.y & 0xffff0000u));
    } else {
        return *reinterpret_cast<const float4*>(p + e);
    }
}

__global__ __launch_bounds__(256, 3) void one_to_c_mfma_k(const float* __restrict__ a,
                                                          const float* __restrict__ w,
                                                          const float* __restrict__ bias,
                                                          float* __restrict__ out, int N, int D, int H,
                                                          int W, int C, int flip, int nbd, int nbh, int nbw,
                                                          int bricks_per_block, int nbricks,
                                                          float* __restrict__ stat_part) {
    __shared__ float hal[2][TH_V];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int q = lane & 15, kk = lane >> 4;
    const bool qok = 4 * q < C;
    // per-channel sum and sum of squares of the bias-free outputs this lane stores (for a BatchNorm behind)
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = make_float4(0.f, 0.f, 0.f, 0.f);
    float wb[7][4];
    int toff[7];
#pragma unroll
    for (int st = 0; st < 7; ++st) {
        const int t = 4 * st + kk;
        toff[st] = t < 27 ? ((t / 9) * TH_H + (t / 3) % 3) * TH_W + t % 3 : 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = 4 * q + j;
            wb[st][j] = (t < 27 && qok) ? (flip ? w[(26 - t) * C + c] : w[c * 27 + t]) : 0.f;
        }
    }
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias && qok) b4 = *reinterpret_cast<const float4*>(bias + 4 * q);
    const int b0 = blockIdx.x * bricks_per_block;
    const int b1 = b0 + bricks_per_block < nbricks ? b0 + bricks_per_block : nbricks;
    int buf = 0;
    float hr[TH_PER_THREAD];
    if (b0 < b1) halo1_fetch(a, hr, b0, nbd, nbh, nbw, D, H, W);
    for (int br = b0; br < b1; ++br, buf ^= 1) {
        int n, z0, y0, x0;
        brick_coords(br, nbd, nbh, nbw, n, z0, y0, x0);
        // two halo images: a wave still working on the previous brick reads the other one
        halo1_put(hal[buf], hr);
        __syncthreads();
        if (br + 1 < b1) halo1_fetch(a, hr, br + 1, nbd, nbh, nbw, D, H, W);
        __builtin_amdgcn_sched_barrier(0);
        const int gz = z0 + wv;
        if (gz >= D) continue;
        float* op = out + ((((long)n * D + gz) * H + y0) * W + x0) * C + 4 * q;
#pragma unroll
        for (int tl = 0; tl < 4; ++tl) {                // 16 voxels = brick rows 2 tl, 2 tl + 1
            const float* hp = hal[buf] + (wv * TH_H + 2 * tl + (q >> 3)) * TH_W + (q & 7);
            float av[7];
#pragma unroll
            for (int st = 0; st < 7; ++st) av[st] = hp[toff[st]];
            f32x4 acc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < 7; ++st)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[st], wb[st][j], acc[j], 0, 0, 0);
            // register e of every accumulator belongs to voxel 4 kk + e of the tile
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = 4 * kk + e;
                const int dy = 2 * tl + (m >> 3), dx = m & 7;
                if (qok && y0 + dy < H && x0 + dx < W) {
                    const float4 v = make_float4(acc[0][e], acc[1][e], acc[2][e], acc[3][e]);
                    s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
                    s2.x = fmaf(v.x, v.x, s2.x); s2.y = fmaf(v.y, v.y, s2.y);
                    s2.z = fmaf(v.z, v.z, s2.z); s2.w = fmaf(v.w, v.w, s2.w);
                    *reinterpret_cast<float4*>(op + ((long)dy * W + dx) * C) =
                        make_float4(v.x + b4.x, v.y + b4.y, v.z + b4.z, v.w + b4.w);
                }
            }
        }
    }
    if (stat_part) {            // part[block][2][C]: the four lanes of a channel quad, then the four waves
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) {
            s1.x += __shfl_xor(s1.x, o, 64); s1.y += __shfl_xor(s1.y, o, 64);
            s1.z += __shfl_xor(s1.z, o, 64); s1.w += __shfl_xor(s1.w, o, 64);
            s2.x += __shfl_xor(s2.x, o, 64); s2.y += __shfl_xor(s2.y, o, 64);
            s2.z += __shfl_xor(s2.z, o, 64); s2.w += __shfl_xor(s2.w, o, 64);
        }
        __syncthreads();                       // every wave is done with the halo images
        float* red = hal[0];                   // [4 waves][2][64 channels]
        if (lane < 16) {
            *reinterpret_cast<float4*>(red + (wv * 2 + 0) * 64 + 4 * q) = s1;
            *reinterpret_cast<float4*>(red + (wv * 2 + 1) * 64 + 4 * q) = s2;
        }
        __syncthreads();
        if (threadIdx.x < 128) {
            const int k = threadIdx.x >> 6, c = threadIdx.x & 63;
            if (c < C)
                stat_part[((long)blockIdx.x * 2 + k) * C + c] =
                    (red[(0 * 2 + k) * 64 + c] + red[(1 * 2 + k) * 64 + c]) +
                    (red[(2 * 2 + k) * 64 + c] + red[(3 * 2 + k) * 64 + c]);
        }
    }
}

// ---------------------------------------------------------------------------------
// Backward of [BatchNorm + LeakyReLU] -> Conv3d(C, 1, 3) with respect to the BatchNorm input, without the
// gradient of the activation ever reaching memory.  That gradient, da[v][c] = sum_tap g1[v - tap + 1] *
// w[tap][c], is the 1 -> C kernel above applied to the single-channel output gradient g1 -- 27
// multiply-adds per element on the matrix pipe -- so both passes of the BatchNorm backward recompute it
// in registers instead of one kernel writing it (157 MB per volume) and two reading it:
//   STAGE 0: dz = da * act'(z), z = y*a + b;  per-block sums of dz and dz * xhat  -> part[block][2][C]
//   STAGE 1: dy = a * (dz - sum(dz)/M - xhat * sum(dz*xhat)/M)                     (sums[2][C] finalised)
// Layout and tiling as one_to_c_mfma_k: a lane holds channels 4q .. 4q+3 of voxels 4 kk + e.
// ---------------------------------------------------------------------------------
// YBF: y points at the bf16 copy of the BatchNorm input.
template <int STAGE, bool YBF>
__global__ __launch_bounds__(256, 3) void tail_bn_bwd_k(const float* __restrict__ g1,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ y, BnPrologue bn,
                                                        const float* __restrict__ sums, float invM,
                                                        float* __restrict__ part, float* __restrict__ dy,
                                                        int N, int D, int H, int W, int C, int nbd, int nbh,
                                                        int nbw, int bricks_per_block, int nbricks) {
    __shared__ float hal[2][TH_V];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int q = lane & 15, kk = lane >> 4;
    const bool qok = 4 * q < C;
    float wb[7][4];
    int toff[7];
#pragma unroll
    for (int st = 0; st < 7; ++st) {
        const int t = 4 * st + kk;
        toff[st] = t < 27 ? ((t / 9) * TH_H + (t / 3) % 3) * TH_W + t % 3 : 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)     // input gradient: w is [1][27][C], taps reversed
            wb[st][j] = (t < 27 && qok) ? w[(26 - t) * C + 4 * q + j] : 0.f;
    }
    float ca[4], cb[4], cm[4], ci[4], k1[4], k2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = qok ? 4 * q + j : 0;
        // the association of bn_act.hip: a = gamma / sqrt(var + eps), xhat = (y - mean) * (1 / sqrt(var + eps))
        ci[j] = 1.0f / sqrtf(bn.var[c] + bn.eps);
        ca[j] = bn.gamma[c] / sqrtf(bn.var[c] + bn.eps);
        cm[j] = bn.mean[c];
        cb[j] = bn.beta[c] - cm[j] * ca[j];
        k1[j] = STAGE == 1 ? sums[c] * invM : 0.f;
        k2[j] = STAGE == 1 ? sums[C + c] * invM : 0.f;
    }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const int b0 = blockIdx.x * bricks_per_block;
    const int b1 = b0 + bricks_per_block < nbricks ? b0 + bricks_per_block : nbricks;
    int buf = 0;
    float hr[TH_PER_THREAD];
    if (b0 < b1) halo1_fetch(g1, hr, b0, nbd, nbh, nbw, D, H, W);
    for (int br = b0; br < b1; ++br, buf ^= 1) {
        int n, z0, y0, x0;
        brick_coords(br, nbd, nbh, nbw, n, z0, y0, x0);
        halo1_put(hal[buf], hr);
        __syncthreads();
        if (br + 1 < b1) halo1_fetch(g1, hr, br + 1, nbd, nbh, nbw, D, H, W);
        __builtin_amdgcn_sched_barrier(0);
        const int gz = z0 + wv;
        if (gz >= D) continue;
        const long base = ((((long)n * D + gz) * H + y0) * W + x0) * C + 4 * q;
#pragma unroll
        for (int tl = 0; tl < 4; ++tl) {                // 16 voxels = brick rows 2 tl, 2 tl + 1
            // this lane's four voxels of the tile: y goes out first and has the products to arrive
            float4 yv[4];
            bool ok[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = 4 * kk + e;
                const int dyy = 2 * tl + (m >> 3), dx = m & 7;
                ok[e] = qok && y0 + dyy < H && x0 + dx < W;
                yv[e] = thin_ld4<YBF>(y, ok[e] ? base + ((long)dyy * W + dx) * C : 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            const float* hp = hal[buf] + (wv * TH_H + 2 * tl + (q >> 3)) * TH_W + (q & 7);
            float av[7];
#pragma unroll
            for (int st = 0; st < 7; ++st) av[st] = hp[toff[st]];
            f32x4 acc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < 7; ++st)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[st], wb[st][j], acc[j], 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float yy[4] = {yv[e].x, yv[e].y, yv[e].z, yv[e].w};
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float z = fmaf(yy[j], ca[j], cb[j]);
                    const float dz = acc[j][e] * (z > 0.f ? 1.f : bn.slope);
                    const float xh = (yy[j] - cm[j]) * ci[j];
                    if (STAGE == 0) {
                        if (ok[e]) {
                            s1[j] += dz;
                            s2[j] = fmaf(dz, xh, s2[j]);
                        }
                    } else {
                        o[j] = ca[j] * (dz - k1[j] - xh * k2[j]);
                    }
                }
                if (STAGE == 1 && ok[e]) {
                    const int m = 4 * kk + e;
                    *reinterpret_cast<float4*>(dy + base + ((long)(2 * tl + (m >> 3)) * W + (m & 7)) * C) =
                        make_float4(o[0], o[1], o[2], o[3]);
                }
            }
        }
    }
    if (STAGE == 0) {           // part[block][2][C]: the four lanes of a channel quad, then the four waves
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int o = 16; o < 64; o <<= 1) {
                s1[j] += __shfl_xor(s1[j], o, 64);
                s2[j] += __shfl_xor(s2[j], o, 64);
            }
        __syncthreads();
        float* red = hal[0];                   // [4 waves][2][64 channels]
        if (lane < 16) {
            *reinterpret_cast<float4*>(red + (wv * 2 + 0) * 64 + 4 * q) = make_float4(s1[0], s1[1], s1[2], s1[3]);
            *reinterpret_cast<float4*>(red + (wv * 2 + 1) * 64 + 4 * q) = make_float4(s2[0], s2[1], s2[2], s2[3]);
        }
        __syncthreads();
        if (threadIdx.x < 128) {
            const int k = threadIdx.x >> 6, c = threadIdx.x & 63;
            if (c < C)
                part[((long)blockIdx.x * 2 + k) * C + c] =
                    (red[(0 * 2 + k) * 64 + c] + red[(1 * 2 + k) * 64 + c]) +
                    (red[(2 * 2 + k) * 64 + c] + red[(3 * 2 + k) * 64 + c]);
        }
    }
}

// ---------------------------------------------------------------------------------
// The encoder stem Conv3d(1, C, 3) -> BatchNorm -> LeakyReLU -> Dropout without its convolution output
// ever reaching memory: y[v][c] = bias[c] + sum_tap a[v + tap - 1] * w[c][tap] costs 27 multiply-adds
// per element from a single-channel halo brick, so every pass that needs y recomputes it on the matrix
// pipe (the tiling of one_to_c_mfma_k) instead of one kernel writing it (157 MB per volume) and up to
// four reading it:
//   SM_STATS: per-block sums of (y - bias) and its square                  -> part[block][2][C]
//   SM_APPLY: out = dropout(act(y * a + b))                                 (forward)
//   SM_BSUM:  dz = dout * keep/(1-p) * act'(z); per-block sums of dz, dz * xhat -> part[block][2][C]
//   SM_BAPP:  dy = a * (dz - sum(dz)/M - xhat * sum(dz * xhat)/M)            (gradient of y, for the
//             stem's weight gradient)
// Recomputed values are bit-identical in every pass (same instruction sequence).
// BITS: the forward pass also leaves two bits per output element -- kept by the dropout, z > 0 -- as one byte
// per channel quad of a voxel (bits 0..3: kept, bits 4..7: positive; byte index = element index / 4), and the
// backward passes take the dropout decision from there instead of a mask tensor or the generator.  (One
// 32-bit word per lane and tile -- four voxels along x -- was measured too: one store instead of four, but
// its index arithmetic costs the apply pass more than the stores do; 470 against 500 us.)
// ---------------------------------------------------------------------------------
enum { SM_STATS = 0, SM_APPLY = 1, SM_BSUM = 2, SM_BAPP = 3 };

// MASK / EMIT: -1 = decided at run time from the arguments; 0, 1, 2 = compiled for no dropout / a mask tensor /
// the generator, and without / with the bf16 copy (the production forms: the other branches are half of the
// apply pass's 50 KB of code otherwise)
template <int MODE, bool BITS = false, int MASK = -1, int EMIT = -1>
__global__ __launch_bounds__(256, 3) void stem_bn_k(const float* __restrict__ a,
                                                    const float* __restrict__ w,
                                                    const float* __restrict__ bias, BnPrologue bn,
                                                    const uint8_t* __restrict__ keep, PhiloxMask rk, float p,
                                                    const float* __restrict__ dout,
                                                    const float* __restrict__ sums, float invM,
                                                    float* __restrict__ part, float* __restrict__ out,
                                                    int N, int D, int H, int W, int C, int nbd, int nbh,
                                                    int nbw, int bricks_per_block, int nbricks,
                                                    uint16_t* __restrict__ out_bf16,
                                                    uint8_t* __restrict__ bits = nullptr) {
    // out_bf16 (SM_APPLY, optional): a bf16 copy of the output, written beside it
    __shared__ float hal[2][TH_V];
    constexpr bool BITS_IN = BITS && (MODE == SM_BSUM || MODE == SM_BAPP);
    // dropout: a mask tensor (keep), or the mask drawn here from the element index (rk.on), or none
    const bool has_keep = MASK < 0 ? keep != nullptr : MASK == 1;
    const bool has_rk = MASK < 0 ? rk.on != 0 : MASK == 2;
    const bool emit = MODE == SM_APPLY && (EMIT < 0 ? out_bf16 != nullptr : EMIT == 1);
    const bool masked = MODE != SM_STATS && (BITS_IN || has_keep || has_rk);
    const uint64_t rsid = (has_rk && rk.epoch) ? rk.sid + *rk.epoch * SIV_EPOCH_STRIDE : rk.sid;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int q = lane & 15, kk = lane >> 4;
    const bool qok = 4 * q < C;
    float wb[7][4];
    int toff[7];
#pragma unroll
    for (int st = 0; st < 7; ++st) {
        const int t = 4 * st + kk;
        toff[st] = t < 27 ? ((t / 9) * TH_H + (t / 3) % 3) * TH_W + t % 3 : 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) wb[st][j] = (t < 27 && qok) ? w[(4 * q + j) * 27 + t] : 0.f;
    }
    float bq[4], ca[4], cb[4], cm[4], ci[4], k1[4], k2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = qok ? 4 * q + j : 0;
        bq[j] = (bias && qok) ? bias[c] : 0.f;
        if (MODE != SM_STATS) {
            // the association of bn_act.hip: a = gamma / sqrt(var + eps), xhat = (y - mean) * (1 / sqrt(var + eps))
            ci[j] = 1.0f / sqrtf(bn.var[c] + bn.eps);
            ca[j] = bn.gamma[c] / sqrtf(bn.var[c] + bn.eps);
            cm[j] = bn.mean[c];
            cb[j] = bn.beta[c] - cm[j] * ca[j];
        }
        k1[j] = MODE == SM_BAPP ? sums[c] * invM : 0.f;
        k2[j] = MODE == SM_BAPP ? sums[C + c] * invM : 0.f;
    }
    const float dscale = 1.0f / (1.0f - p);
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const int b0 = blockIdx.x * bricks_per_block;
    const int b1 = b0 + bricks_per_block < nbricks ? b0 + bricks_per_block : nbricks;
    int buf = 0;
    float hr[TH_PER_THREAD];
    if (b0 < b1) halo1_fetch(a, hr, b0, nbd, nbh, nbw, D, H, W);
    for (int br = b0; br < b1; ++br, buf ^= 1) {
        int n, z0, y0, x0;
        brick_coords(br, nbd, nbh, nbw, n, z0, y0, x0);
        halo1_put(hal[buf], hr);
        __syncthreads();
        if (br + 1 < b1) halo1_fetch(a, hr, br + 1, nbd, nbh, nbw, D, H, W);
        __builtin_amdgcn_sched_barrier(0);
        const int gz = z0 + wv;
        if (gz >= D) continue;
        const long base = ((((long)n * D + gz) * H + y0) * W + x0) * C + 4 * q;
#pragma unroll
        for (int tl = 0; tl < 4; ++tl) {                // 16 voxels = brick rows 2 tl, 2 tl + 1
            bool ok[4];
            long off[4];
            float4 gv[4];
            uint32_t kv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = 4 * kk + e;
                const int dyy = 2 * tl + (m >> 3), dx = m & 7;
                ok[e] = qok && y0 + dyy < H && x0 + dx < W;
                off[e] = ok[e] ? base + ((long)dyy * W + dx) * C : 0;
                // what the pass reads goes out first (clamped, unconditional) and has the products to arrive
                if (MODE == SM_BSUM || MODE == SM_BAPP) gv[e] = *reinterpret_cast<const float4*>(dout + off[e]);
                if (BITS_IN) kv[e] = bits[off[e] >> 2];
                else if (MODE != SM_STATS && has_keep) kv[e] = *reinterpret_cast<const uint32_t*>(keep + off[e]);
            }
            __builtin_amdgcn_sched_barrier(0);
            const float* hp = hal[buf] + (wv * TH_H + 2 * tl + (q >> 3)) * TH_W + (q & 7);
            float av[7];
#pragma unroll
            for (int st = 0; st < 7; ++st) av[st] = hp[toff[st]];
            f32x4 acc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < 7; ++st)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[st], wb[st][j], acc[j], 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gg[4] = {gv[e].x, gv[e].y, gv[e].z, gv[e].w};
                float o[4];
                bool kp[4] = {true, true, true, true};
                uint32_t pos = 0;
                if (BITS_IN) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) kp[j] = ((kv[e] >> j) & 1u) != 0;
                } else if (MODE != SM_STATS && has_keep) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) kp[j] = ((kv[e] >> (8 * j)) & 0xffu) != 0;
                } else if (MODE != SM_STATS && has_rk) {
                    const u4 r = philox4x32_10((uint64_t)(off[e] >> 2), rsid, rk.seed);
                    kp[0] = philox_keeps(r.x, p); kp[1] = philox_keeps(r.y, p);
                    kp[2] = philox_keeps(r.z, p); kp[3] = philox_keeps(r.w, p);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = acc[j][e];                  // bias-free convolution output
                    if (MODE == SM_STATS) {
                        if (ok[e]) {
                            s1[j] += v;
                            s2[j] = fmaf(v, v, s2[j]);
                        }
                        continue;
                    }
                    const float yv = v + bq[j];
                    const float z = fmaf(yv, ca[j], cb[j]);
                    if (MODE == SM_APPLY) {
                        float r = z > 0.f ? z : z * bn.slope;
                        if (masked) r = kp[j] ? r * dscale : 0.f;
                        o[j] = r;
                        if (BITS) pos |= (kp[j] ? 1u : 0u) << j | (z > 0.f ? 16u : 0u) << j;
                        continue;
                    }
                    float g = gg[j];
                    if (masked) g = kp[j] ? g * dscale : 0.f;
                    const float dz = g * (z > 0.f ? 1.f : bn.slope);
                    const float xh = (yv - cm[j]) * ci[j];
                    if (MODE == SM_BSUM) {
                        if (ok[e]) {
                            s1[j] += dz;
                            s2[j] = fmaf(dz, xh, s2[j]);
                        }
                    } else {
                        o[j] = ca[j] * (dz - k1[j] - xh * k2[j]);
                    }
                }
                if ((MODE == SM_APPLY || MODE == SM_BAPP) && ok[e]) {
                    *reinterpret_cast<float4*>(out + off[e]) = make_float4(o[0], o[1], o[2], o[3]);
                    if (emit) thin_st_bf4(out_bf16, off[e], make_float4(o[0], o[1], o[2], o[3]));
                    if (MODE == SM_APPLY && BITS) bits[off[e] >> 2] = (uint8_t)pos;
                }
            }
        }
    }
    if (MODE == SM_STATS || MODE == SM_BSUM) {   // part[block][2][C]: the lanes of a channel quad, then the waves
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int o = 16; o < 64; o <<= 1) {
                s1[j] += __shfl_xor(s1[j], o, 64);
                s2[j] += __shfl_xor(s2[j], o, 64);
            }
        __syncthreads();
        float* red = hal[0];                   // [4 waves][2][64 channels]
        if (lane < 16) {
            *reinterpret_cast<float4*>(red + (wv * 2 + 0) * 64 + 4 * q) = make_float4(s1[0], s1[1], s1[2], s1[3]);
            *reinterpret_cast<float4*>(red + (wv * 2 + 1) * 64 + 4 * q) = make_float4(s2[0], s2[1], s2[2], s2[3]);
        }
        __syncthreads();
        if (threadIdx.x < 128) {
            const int k = threadIdx.x >> 6, c = threadIdx.x & 63;
            if (c < C)
                part[((long)blockIdx.x * 2 + k) * C + c] =
                    (red[(0 * 2 + k) * 64 + c] + red[(1 * 2 + k) * 64 + c]) +
                    (red[(2 * 2 + k) * 64 + c] + red[(3 * 2 + k) * 64 + c]);
        }
    }
}

// ---------------------------------------------------------------------------------
// c_to_one.  Stage 1: p[u][tap] = <a[u][:], wk[tap][:]> for every halo voxel u, channels
// streamed through LDS in chunks of 8 with each thread keeping the 27 running dots of its
// voxels in registers.  Stage 2: out[v] = b + sum_tap p[v+tap-1][tap].
// ---------------------------------------------------------------------------------
constexpr int C1_CK = 8, C1_CKP = 12, C1_PP = 27;
constexpr int C1_ROUNDS = (TH_V + 255) / 256;  // 3
constexpr int C1_LDS_FLOATS = TH_V * C1_PP;    // p image; also covers the a chunk and weights
static_assert(TH_V * C1_CKP + 27 * C1_CK <= C1_LDS_FLOATS, "LDS carve");

__global__ __launch_bounds__(256) void c_to_one_k(const float* __restrict__ a,
                                                  const float* __restrict__ w,
                                                  const float* __restrict__ bias,
                                                  float* __restrict__ out, int N, int D, int H, int W,
                                                  int C, int flip, int nbd, int nbh, int nbw) {
    __shared__ __attribute__((aligned(16))) float lds[C1_LDS_FLOATS];
    float* ach = lds;                       // [600][12]
    float* wch = lds + TH_V * C1_CKP;       // [27][8]
    int n, z0, y0, x0;
    brick_coords(blockIdx.x, nbd, nbh, nbw, n, z0, y0, x0);
    float p[C1_ROUNDS][27];
#pragma unroll
    for (int r = 0; r < C1_ROUNDS; ++r)
#pragma unroll
        for (int t = 0; t < 27; ++t) p[r][t] = 0.f;

    for (int c0 = 0; c0 < C; c0 += C1_CK) {
        __syncthreads();
        for (int idx = threadIdx.x; idx < TH_V * 2; idx += 256) {
            const int hv = idx >> 1, qq = idx & 1;
            const int hx = hv % TH_W, hy = (hv / TH_W) % TH_H, hz = hv / (TH_W * TH_H);
            const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W)
                v = *reinterpret_cast<const float4*>(
                    a + ((((long)n * D + gz) * H + gy) * W + gx) * C + c0 + 4 * qq);
            *reinterpret_cast<float4*>(ach + hv * C1_CKP + 4 * qq) = v;
        }
        if (threadIdx.x < 27 * C1_CK) {
            const int t = threadIdx.x / C1_CK, c = threadIdx.x % C1_CK;
            wch[threadIdx.x] = flip ? w[(long)(c0 + c) * 27 + 26 - t] : w[(long)t * C + c0 + c];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < C1_ROUNDS; ++r) {
            const int u = threadIdx.x + 256 * r;
            if (u < TH_V) {
                const float4 a0 = *reinterpret_cast<const float4*>(ach + u * C1_CKP);
                const float4 a1 = *reinterpret_cast<const float4*>(ach + u * C1_CKP + 4);
#pragma unroll
                for (int t = 0; t < 27; ++t) {
                    const float4 w0 = *reinterpret_cast<const float4*>(wch + t * C1_CK);
                    const float4 w1 = *reinterpret_cast<const float4*>(wch + t * C1_CK + 4);
                    float s = p[r][t];
                    s = fmaf(a0.x, w0.x, s); s = fmaf(a0.y, w0.y, s);
                    s = fmaf(a0.z, w0.z, s); s = fmaf(a0.w, w0.w, s);
                    s = fmaf(a1.x, w1.x, s); s = fmaf(a1.y, w1.y, s);
                    s = fmaf(a1.z, w1.z, s); s = fmaf(a1.w, w1.w, s);
                    p[r][t] = s;
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < C1_ROUNDS; ++r) {
        const int u = threadIdx.x + 256 * r;
        if (u < TH_V) {
#pragma unroll
            for (int t = 0; t < 27; ++t) lds[u * C1_PP + t] = p[r][t];
        }
    }
    __syncthreads();
    const int m = threadIdx.x;
    const int dz = m >> 6, dy = (m >> 3) & 7, dx = m & 7;
    const int gz = z0 + dz, gy = y0 + dy, gx = x0 + dx;
    if (gz < D && gy < H && gx < W) {
        float s = bias ? bias[0] : 0.f;
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            const int hv = ((dz + t / 9) * TH_H + dy + (t / 3) % 3) * TH_W + dx + t % 3;
            s += lds[hv * C1_PP + t];
        }
        out[(((long)n * D + gz) * H + gy) * W + gx] = s;
    }
}

// ---------------------------------------------------------------------------------
// c_to_one on the matrix pipe (C = 8*KG <= 64).  Stage 1 is a small GEMM per brick:
//   p[u][tap] = sum_c a[u][c] * wk[tap][c],  u over the 600 halo voxels (19 tiles of 32 rows),
//   27 taps padded to the 32 columns of v_mfma_f32_32x32x2_f32, K = C.
// A rows come straight from global memory (lane (i,h) reads the 16-byte channel quad 4h of every
// 8-channel group of voxel i: the k order is permuted identically for A and B), the weights sit
// in registers, the p image goes to LDS and stage 2 gathers out[v] = b + sum_tap p[v+tap-1][tap].
// The VALU kernel above spends its time re-reading the weights from LDS for every voxel.
// ---------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int C1M_TILES = (TH_V + 31) / 32;   // 19

template <int KG>
__device__ __forceinline__ void c1m_load_tile(const float* __restrict__ a, int tile, int i, int h, int n,
                                              int z0, int y0, int x0, int D, int H, int W, int C,
                                              float4 (&av)[KG]) {
    const int u = tile * 32 + i;
    const int hx = u % TH_W, hy = (u / TH_W) % TH_H, hz = u / (TH_W * TH_H);
    const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
    const bool ok = u < TH_V && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W;
    const float* src = a + ((((long)n * D + gz) * H + gy) * W + gx) * C + 4 * h;
#pragma unroll
    for (int g = 0; g < KG; ++g)
        av[g] = ok ? *reinterpret_cast<const float4*>(src + 8 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
}

template <int KG>
__global__ __launch_bounds__(256) void c_to_one_mfma_k(const float* __restrict__ a,
                                                       const float* __restrict__ w,
                                                       const float* __restrict__ bias,
                                                       float* __restrict__ out, int N, int D, int H,
                                                       int W, int flip, int nbd, int nbh, int nbw) {
    constexpr int C = 8 * KG;
    __shared__ float pim[TH_V * C1_PP];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    // B fragments: column i = tap (27 used), k slots = channels 8g + 4h + {0..3}
    float4 bw[KG];
#pragma unroll
    for (int g = 0; g < KG; ++g) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < 27) {
            const int c = 8 * g + 4 * h;
            if (!flip) {   // w is [1][27][C]
                v = *reinterpret_cast<const float4*>(w + i * C + c);
            } else {       // w is [C][27][1], taps reversed
                v = make_float4(w[(c + 0) * 27 + 26 - i], w[(c + 1) * 27 + 26 - i],
                                w[(c + 2) * 27 + 26 - i], w[(c + 3) * 27 + 26 - i]);
            }
        }
        bw[g] = v;
    }
    int n, z0, y0, x0;
    brick_coords(blockIdx.x, nbd, nbh, nbw, n, z0, y0, x0);
    float4 av[KG], nx[KG];
    c1m_load_tile<KG>(a, wv, i, h, n, z0, y0, x0, D, H, W, C, av);
    for (int tile = wv; tile < C1M_TILES; tile += 4) {
        if (tile + 4 < C1M_TILES) c1m_load_tile<KG>(a, tile + 4, i, h, n, z0, y0, x0, D, H, W, C, nx);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g].x, bw[g].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g].y, bw[g].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g].z, bw[g].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g].w, bw[g].w, acc, 0, 0, 0);
        }
        if (i < 27) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int u = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (u < TH_V) pim[u * C1_PP + i] = acc[r];
            }
        }
#pragma unroll
        for (int g = 0; g < KG; ++g) av[g] = nx[g];
    }
    __syncthreads();
    const int m = threadIdx.x;
    const int dz = m >> 6, dy = (m >> 3) & 7, dx = m & 7;
    const int gz = z0 + dz, gy = y0 + dy, gx = x0 + dx;
    if (gz < D && gy < H && gx < W) {
        float s = bias ? bias[0] : 0.f;
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            const int hv = ((dz + t / 9) * TH_H + dy + (t / 3) % 3) * TH_W + dx + t % 3;
            s += pim[hv * C1_PP + t];
        }
        out[(((long)n * D + gz) * H + gy) * W + gx] = s;
    }
}

// ---------------------------------------------------------------------------------
// c_to_one for volumes that are long in z: a block owns an 8 x 16 (H, W) column and walks along z.
// Every input plane of the column (10 x 18 halo voxels = 6 row tiles of 32) goes through the same small
// GEMM as above exactly once -- the halo is re-read in the plane only, 1.4x the tensor instead of the 2.3x
// of the 4x8x8 bricks, which is what bounds the brick kernel (L2 traffic) -- and the 27-column tap image
// of the last three planes is kept in an LDS ring; output plane z - 1 is gathered when plane z is in.
// ---------------------------------------------------------------------------------
constexpr int ZW_H = 8, ZW_W = 16, ZW_HW = ZW_W + 2, ZW_HV = (ZW_H + 2) * ZW_HW;   // 180 halo voxels
constexpr int ZW_TILES = (ZW_HV + 31) / 32;                                        // 6

template <int KG>
__global__ __launch_bounds__(256) void c_to_one_zwalk_k(const float* __restrict__ a,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ bias,
                                                        float* __restrict__ out, int N, int D, int H, int W,
                                                        int flip, int nth, int ntw, int zsplit) {
    constexpr int C = 8 * KG;
    __shared__ float pim[3][ZW_HV * C1_PP];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    float4 bw[KG];
#pragma unroll
    for (int g = 0; g < KG; ++g) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < 27) {
            const int c = 8 * g + 4 * h;
            if (!flip) {
                v = *reinterpret_cast<const float4*>(w + i * C + c);
            } else {
                v = make_float4(w[(c + 0) * 27 + 26 - i], w[(c + 1) * 27 + 26 - i],
                                w[(c + 2) * 27 + 26 - i], w[(c + 3) * 27 + 26 - i]);
            }
        }
        bw[g] = v;
    }
    int b = blockIdx.x;
    const int zp = b % zsplit; b /= zsplit;
    const int tx = b % ntw; b /= ntw;
    const int ty = b % nth;
    const int n = b / nth;
    const int zlo = (int)((long)zp * D / zsplit), zhi = (int)((long)(zp + 1) * D / zsplit);
    const int y0 = ty * ZW_H, x0 = tx * ZW_W;

    // this wave's row tiles wv and wv + 4 (the latter only for wv < 2): in-plane offset of the lane's voxel
    int off[2];
    bool okv[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int u = (wv + 4 * q) * 32 + i;
        const int hy = u / ZW_HW, hx = u % ZW_HW;
        const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        okv[q] = wv + 4 * q < ZW_TILES && u < ZW_HV && gy >= 0 && gy < H && gx >= 0 && gx < W;
        off[q] = okv[q] ? (gy * W + gx) * C + 4 * h : 0;
    }
    const float* an = a + (long)n * D * H * W * C;
    float4 av[2][KG];
    auto load_plane = [&](int z) {
        const bool zok = z >= 0 && z < D;
        const float* pl = an + (long)(zok ? z : 0) * H * W * C;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int g = 0; g < KG; ++g)
                av[q][g] = (zok && okv[q]) ? *reinterpret_cast<const float4*>(pl + off[q] + 8 * g)
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    const float b0 = bias ? bias[0] : 0.f;
    const int m = threadIdx.x;
    const int dy = m >> 4, dx = m & 15;                 // gather: threads 0..127

    load_plane(zlo - 1);
    for (int z = zlo - 1; z <= zhi; ++z) {
        float* pz = pim[(z + 3) % 3];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (wv + 4 * q < ZW_TILES) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int g = 0; g < KG; ++g) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][g].x, bw[g].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][g].y, bw[g].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][g].z, bw[g].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][g].w, bw[g].w, acc, 0, 0, 0);
                }
                if (i < 27) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int u = (wv + 4 * q) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        if (u < ZW_HV) pz[u * C1_PP + i] = acc[r];
                    }
                }
            }
        }
        if (z < zhi) load_plane(z + 1);                 // in flight during the gather
        __syncthreads();
        const int zo = z - 1;
        if (zo >= zlo && m < ZW_H * ZW_W) {
            const int gy = y0 + dy, gx = x0 + dx;
            if (gy < H && gx < W) {
                float s = b0;
#pragma unroll
                for (int t = 0; t < 27; ++t) {
                    const float* pp = pim[(zo + t / 9 + 2) % 3];         // plane zo - 1 + kd
                    s += pp[((dy + (t / 3) % 3) * ZW_HW + dx + t % 3) * C1_PP + t];
                }
                out[(((long)n * D + zo) * H + gy) * W + gx] = s;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------
// The column walk with 12 x 16 columns: the 14 x 18 halo plane is 252 voxels = eight full row tiles, two
// per wave (the 8 x 16 column has six: two SIMDs carry two tiles, two carry one), and 1.31x the tensor
// is read.  Only the tap image of the CURRENT plane is kept (two copies, written alternately, so one
// barrier per plane suffices): the gather thread of an output voxel adds the nine taps of each kd
// from it and carries the sums for output planes z and z + 1 in registers; output plane z - 1 is
// complete when input plane z has been gathered.  The loads of plane z + 1 are issued before the
// products of plane z, so they have the whole GEMM phase to arrive.
// ---------------------------------------------------------------------------------
constexpr int ZX_H = 12, ZX_W = 16, ZX_HW = ZX_W + 2, ZX_HV = (ZX_H + 2) * ZX_HW;   // 252 halo voxels
constexpr int ZX_TILES = (ZX_HV + 31) / 32;                                          // 8

// BatchNorm of the layer in front, for the kernels that apply it (and the LeakyReLU behind it) to their
// operand instead of reading an activated tensor: z = x * a[c] + b[c] with a = gamma / sqrt(var + eps),
// b = beta - mean * a (the arithmetic of bn_act_fwd_k), act(z) = z > 0 ? z : z * slope.
__device__ __forceinline__ void bn_coeff(const BnPrologue& bn, int c, float& a, float& b) {
    a = bn.gamma[c] / sqrtf(bn.var[c] + bn.eps);
    b = bn.beta[c] - bn.mean[c] * a;
}
__device__ __forceinline__ float bn_act1(float x, float a, float b, float slope) {
    const float z = fmaf(x, a, b);
    return z > 0.f ? z : z * slope;
}

// BN: the operand a is the raw output of the convolution in front; BatchNorm + LeakyReLU are applied to
// every loaded value (coefficients in LDS), out-of-volume voxels stay zero (the padding of the ACTIVATED
// tensor).  The activated tensor is then never written or read.
// a_bf16 (optional): a bf16 copy of the operand, written by the block that owns a voxel (the interior of its
// column, its own z range) from the very values it loaded.
template <int KG, bool BN, bool EMIT = false>
__global__ __launch_bounds__(256, 2) void c_to_one_zwalk12_k(const float* __restrict__ a,
                                                             const float* __restrict__ w,
                                                             const float* __restrict__ bias,
                                                             float* __restrict__ out, int N, int D, int H,
                                                             int W, int flip, int nth, int ntw, int zsplit,
                                                             BnPrologue bn, uint16_t* __restrict__ a_bf16) {
    constexpr int C = 8 * KG;
    static_assert(ZX_TILES == 8, "two row tiles per wave");
    __shared__ float pim[2][ZX_TILES * 32 * C1_PP];    // all 256 rows of the eight tiles are written
    __shared__ __attribute__((aligned(16))) float bnab[2][BN ? C : 4];
    if (BN) {
        if (threadIdx.x < C) bn_coeff(bn, threadIdx.x, bnab[0][threadIdx.x], bnab[1][threadIdx.x]);
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    float4 bw[KG];
#pragma unroll
    for (int g = 0; g < KG; ++g) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < 27) {
            const int c = 8 * g + 4 * h;
            if (!flip) {
                v = *reinterpret_cast<const float4*>(w + i * C + c);
            } else {
                v = make_float4(w[(c + 0) * 27 + 26 - i], w[(c + 1) * 27 + 26 - i],
                                w[(c + 2) * 27 + 26 - i], w[(c + 3) * 27 + 26 - i]);
            }
        }
        bw[g] = v;
    }
    // Consecutive block ids go to different XCDs (each with an L2 of its own): an XCD is given a contiguous
    // eighth of the columns instead, so that the halo voxels two neighbouring columns share are found in its L2.
    int b = blockIdx.x;
    if ((gridDim.x & 7) == 0) b = (b & 7) * (gridDim.x >> 3) + (b >> 3);
    // (and within a sample the columns of one z segment follow each other: the blocks resident on an XCD at a
    // time are then in-plane neighbours walking the same planes)
    const int tx = b % ntw; b /= ntw;
    const int ty = b % nth; b /= nth;
    const int zp = b % zsplit;
    const int n = b / zsplit;
    const int zlo = (int)((long)zp * D / zsplit), zhi = (int)((long)(zp + 1) * D / zsplit);
    const int y0 = ty * ZX_H, x0 = tx * ZX_W;

    int off[2];
    bool okv[2], own[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int u = (wv + 4 * q) * 32 + i;
        const int hy = u / ZX_HW, hx = u % ZX_HW;
        const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        okv[q] = u < ZX_HV && gy >= 0 && gy < H && gx >= 0 && gx < W;
        own[q] = okv[q] && hy >= 1 && hy <= ZX_H && hx >= 1 && hx <= ZX_W;     // not a halo voxel of this column
        off[q] = okv[q] ? (gy * W + gx) * C + 4 * h : 0;
    }
    uint16_t* an16 = EMIT ? a_bf16 + (long)n * D * H * W * C : nullptr;
    const float* an = a + (long)n * D * H * W * C;
    float4 av[2][2][KG];
    auto load_plane = [&](float4 (&dst)[2][KG], int z) {
        const bool zok = z >= 0 && z < D;
        const float* pl = an + (long)(zok ? z : 0) * H * W * C;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int g = 0; g < KG; ++g)   // always a valid address (offset 0 / plane 0 when masked):
                dst[q][g] = *reinterpret_cast<const float4*>(pl + off[q] + 8 * g);   // masked at use
    };
    const float b0 = bias ? bias[0] : 0.f;
    const int m = threadIdx.x;
    const int dy = m >> 4, dx = m & 15;                 // gather: threads 0..191
    const bool gok = m < ZX_H * ZX_W && y0 + dy < H && x0 + dx < W;
    float* op = out + (long)n * D * H * W + (long)(y0 + dy) * W + x0 + dx;
    float o_z = b0, o_z1 = b0;                          // running sums of output planes z - 1 and z

    // Planes zlo - 1 .. zhi, two per trip with nothing conditional in between (a branch around the
    // loads would make the compiler wait for them at the join): an odd count runs one plane more,
    // whose output plane is outside the range and not stored; loads beyond the volume read plane 0.
    load_plane(av[0], zlo - 1);
    for (int zz = zlo - 1; zz <= zhi; zz += 2) {
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const int z = zz + par;
            // (the stores of the bf16 copy go out BEFORE the next plane's loads are issued: loads and stores share
            // one in-order counter, and a store behind those loads would make every later wait for them a wait
            // for everything)
            if (EMIT && z >= zlo && z < zhi) {         // the raw values of the voxels this block owns, as bf16
                // Lanes i and i + 32 hold the two channel quads of a voxel's 8-channel group.  They swap one
                // quad of every pair of groups, so that each stores 8 channels = 16 contiguous bytes (lane h = 0
                // group g, lane h = 1 group g + 1): half as many, twice as wide store requests per voxel.
                uint16_t* pl16 = an16 + (long)z * H * W * C;
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int g = 0; g < KG; g += 2) {
                        const float4 a4 = av[par][q][g], b4 = av[par][q][g + 1];
                        const float4 send = h ? a4 : b4;
                        float4 recv;
                        recv.x = __shfl_xor(send.x, 32, 64); recv.y = __shfl_xor(send.y, 32, 64);
                        recv.z = __shfl_xor(send.z, 32, 64); recv.w = __shfl_xor(send.w, 32, 64);
                        const float4 lo = h ? recv : a4, hi = h ? b4 : recv;
                        if (own[q]) {
                            uint4 o;
                            o.x = bf16_rne(lo.x) | (bf16_rne(lo.y) << 16); o.y = bf16_rne(lo.z) | (bf16_rne(lo.w) << 16);
                            o.z = bf16_rne(hi.x) | (bf16_rne(hi.y) << 16); o.w = bf16_rne(hi.z) | (bf16_rne(hi.w) << 16);
                            *reinterpret_cast<uint4*>(pl16 + off[q] - 4 * h + 8 * (g + h)) = o;
                        }
                    }
            }
            load_plane(av[par ^ 1], z + 1);
            __builtin_amdgcn_sched_barrier(0);          // keep the loads ahead of the products
            float* pz = pim[par];
            f32x16 acc[2];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
            const bool zin = z >= 0 && z < D;
            const bool ok0 = zin && okv[0], ok1 = zin && okv[1];
#pragma unroll
            for (int g = 0; g < KG; ++g) {              // the two row tiles alternate: independent chains
                float4 v0 = av[par][0][g], v1 = av[par][1][g];
                if (BN) {
                    const float4 ca = *reinterpret_cast<const float4*>(&bnab[0][8 * g + 4 * h]);
                    const float4 cb = *reinterpret_cast<const float4*>(&bnab[1][8 * g + 4 * h]);
                    v0 = make_float4(bn_act1(v0.x, ca.x, cb.x, bn.slope), bn_act1(v0.y, ca.y, cb.y, bn.slope),
                                     bn_act1(v0.z, ca.z, cb.z, bn.slope), bn_act1(v0.w, ca.w, cb.w, bn.slope));
                    v1 = make_float4(bn_act1(v1.x, ca.x, cb.x, bn.slope), bn_act1(v1.y, ca.y, cb.y, bn.slope),
                                     bn_act1(v1.z, ca.z, cb.z, bn.slope), bn_act1(v1.w, ca.w, cb.w, bn.slope));
                }
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ok0 ? v0.x : 0.f, bw[g].x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ok1 ? v1.x : 0.f, bw[g].x, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ok0 ? v0.y : 0.f, bw[g].y, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ok1 ? v1.y : 0.f, bw[g].y, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ok0 ? v0.z : 0.f, bw[g].z, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ok1 ? v1.z : 0.f, bw[g].z, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ok0 ? v0.w : 0.f, bw[g].w, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ok1 ? v1.w : 0.f, bw[g].w, acc[1], 0, 0, 0);
            }
            if (i < 27) {
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int u = (wv + 4 * q) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        pz[u * C1_PP + i] = acc[q][r];
                    }
            }
            __syncthreads();
            if (gok) {
                float s[3] = {0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 27; ++t)
                    s[t / 9] += pz[((dy + (t / 3) % 3) * ZX_HW + dx + t % 3) * C1_PP + t];
                // input plane z is under tap kd of output plane z + 1 - kd
                const float done = o_z + s[2];
                if (z - 1 >= zlo && z - 1 < zhi) op[(long)(z - 1) * H * W] = done;
                o_z = o_z1 + s[1];
                o_z1 = b0 + s[0];
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// thin_wg.  Thread = (voxel lane, channel quad) with 27 (+1 for the plain sum of g) float4
// accumulators; every block walks a contiguous range of bricks, then the voxel lanes are
// combined through LDS and one partial row per block is written: part[block][C][28].
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void thin_wg_k(const float* __restrict__ a,
                                                 const float* __restrict__ g,
                                                 float* __restrict__ part, int N, int D, int H, int W,
                                                 int C, int nbd, int nbh, int nbw,
                                                 int bricks_per_block) {
    __shared__ float hal[TH_V];
    __shared__ __attribute__((aligned(16))) float red[256 * 4];
    const int QC = C >> 2;
    const int lanes = 256 / QC;
    const int q = threadIdx.x % QC, vl = threadIdx.x / QC;
    const bool active = vl < lanes;
    float4 acc[28];
#pragma unroll
    for (int t = 0; t < 28; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    const long nbricks = (long)N * nbd * nbh * nbw;
    const long b0 = (long)blockIdx.x * bricks_per_block;
    long b1 = b0 + bricks_per_block;
    if (b1 > nbricks) b1 = nbricks;
    for (long br = b0; br < b1; ++br) {
        int n, z0, y0, x0;
        brick_coords((int)br, nbd, nbh, nbw, n, z0, y0, x0);
        __syncthreads();
        load_halo1(a, hal, n, z0, y0, x0, D, H, W);
        __syncthreads();
        if (active) {
            for (int m = vl; m < TB_V; m += lanes) {
                const int dz = m >> 6, dy = (m >> 3) & 7, dx = m & 7;
                const int gz = z0 + dz, gy = y0 + dy, gx = x0 + dx;
                if (gz >= D || gy >= H || gx >= W) continue;
                const float4 gv = *reinterpret_cast<const float4*>(
                    g + ((((long)n * D + gz) * H + gy) * W + gx) * C + 4 * q);
                const float* hp = hal + (dz * TH_H + dy) * TH_W + dx;
#pragma unroll
                for (int t = 0; t < 27; ++t) {
                    const float av = hp[((t / 9) * TH_H + (t / 3) % 3) * TH_W + t % 3];
                    acc[t].x = fmaf(av, gv.x, acc[t].x);
                    acc[t].y = fmaf(av, gv.y, acc[t].y);
                    acc[t].z = fmaf(av, gv.z, acc[t].z);
                    acc[t].w = fmaf(av, gv.w, acc[t].w);
                }
                acc[27].x += gv.x; acc[27].y += gv.y; acc[27].z += gv.z; acc[27].w += gv.w;
            }
        }
    }
    // combine voxel lanes, tap by tap, in fixed order
    float* pp = part + (long)blockIdx.x * C * 28;
#pragma unroll
    for (int t = 0; t < 28; ++t) {
        __syncthreads();
        *reinterpret_cast<float4*>(red + threadIdx.x * 4) = active ? acc[t] : make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
        if (threadIdx.x < C) {
            const int c = threadIdx.x;
            float s = 0.f;
            for (int l = 0; l < lanes; ++l) s += red[(l * QC + (c >> 2)) * 4 + (c & 3)];
            pp[c * 28 + t] = s;
        }
    }
}

// ---------------------------------------------------------------------------------
// thin_wg on the matrix pipe (C <= 64).  T = G^T · im2col(a): rows = channels, columns = the 27 taps
// plus a column of ones (the plain sum of g), K = voxels, four per v_mfma_f32_16x16x4_f32.
// Wave wv owns plane wv of every 4x8x8 brick of the block and walks it in groups of four voxels
// along x: lane (q, kk) loads the channel quad q of voxel kk as one float4 -- a load instruction
// covers 1 KB of contiguous memory -- and its four components are the A operands of four
// instructions whose row m stands for channel 4m + j; the B operand is the halo value under tap
// q (+16) of voxel kk, one LDS read per tap half.  The VALU kernel above needs 27 LDS reads and
// 108 multiply-adds per float4 of g and is bound by them; here four voxels x 64 channels cost
// eight 32-cycle matrix instructions.  The four waves are combined through LDS in fixed order and
// one partial row per block is written, part[block][C][28], as above.
// ---------------------------------------------------------------------------------
// GBF: g points at the bf16 copy of the (raw) multi-channel operand.
// SB: g is the gradient behind the stem's Dropout and LeakyReLU; every element is first taken back through
// them with the two bits the stem's forward pass left per element (stem_bn_k): g * keep / (1 - p) * act'(z).
// (Four waves per SIMD at 128 registers: the plain form spills 12 bytes per lane and gains 6-10 %, the forms with
// a prologue spill 36-96 bytes and lose 4-14 %; the model uses only those.)
template <bool BN, bool GBF, bool SB = false>
__global__ __launch_bounds__(256, 3) void thin_wg_mfma_k(const float* __restrict__ a,
                                                      const float* __restrict__ g,
                                                      float* __restrict__ part, int N, int D, int H, int W,
                                                      int C, int nbd, int nbh, int nbw,
                                                      int bricks_per_block, BnPrologue bn,
                                                      const uint8_t* __restrict__ sbits = nullptr,
                                                      float dscale = 1.f, float* __restrict__ rpart = nullptr) {
    // SB also forms the tap correlation of the single-channel operand (which the stem's weight-only backward
    // needs beside T) from the halo images it stages anyway: rpart[block][32][32]
    __shared__ float hal[2][TH_V];
    __shared__ float red[64 * 28];                      // T image of the block (1792 floats), then the 32 x 32 tap correlation
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int q = lane & 15, kk = lane >> 4;
    const bool qok = 4 * q < C;
    // BN: g is the raw output of the convolution in front; the lane's four channels get BatchNorm +
    // LeakyReLU applied as they are loaded
    float4 ca = make_float4(1.f, 1.f, 1.f, 1.f), cb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (BN && qok) {
        bn_coeff(bn, 4 * q + 0, ca.x, cb.x); bn_coeff(bn, 4 * q + 1, ca.y, cb.y);
        bn_coeff(bn, 4 * q + 2, ca.z, cb.z); bn_coeff(bn, 4 * q + 3, ca.w, cb.w);
    }
    // tap columns q and q + 16 of this lane: halo offset, or the constant column
    int toff[2];
    float bconst[2];
    bool btap[2];
#pragma unroll
    for (int cg = 0; cg < 2; ++cg) {
        const int t = q + 16 * cg;
        btap[cg] = t < 27;
        toff[cg] = btap[cg] ? ((t / 9) * TH_H + (t / 3) % 3) * TH_W + t % 3 : 0;
        bconst[cg] = t == 27 ? 1.f : 0.f;
    }
    f32x4 acc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int cg = 0; cg < 2; ++cg) acc[j][cg] = f32x4{0.f, 0.f, 0.f, 0.f};
    // SB: P'^T P' of the halo (lane = (column t32, voxel k32) of a 32x32x2 instruction, rows and columns = taps)
    f32x16t racc;
    const int t32 = lane & 31, k32 = lane >> 5;
    const int roff = t32 < 27 ? ((t32 / 9) * TH_H + (t32 / 3) % 3) * TH_W + t32 % 3 : 0;
    if (SB) {
#pragma unroll
        for (int r = 0; r < 16; ++r) racc[r] = 0.f;
    }
    const long nbricks = (long)N * nbd * nbh * nbw;
    const long b0 = (long)blockIdx.x * bricks_per_block;
    long b1 = b0 + bricks_per_block;
    if (b1 > nbricks) b1 = nbricks;
    int buf = 0;
    float hr[TH_PER_THREAD];
    // The multi-channel operand is fetched one group of four brick rows ahead of its use, ACROSS bricks: the
    // first group of brick b + 1 is requested while the last group of brick b is multiplied (its addresses do
    // not depend on the halo image the barrier waits for), so no group's latency is paid in the open.
    // group u (0..15) of a brick plane: brick row u >> 1, voxels 4 (u & 1) + kk
    auto load4 = [&](float4 (&dst)[4], uint32_t (&bd)[4], int c4, int n, int z0, int y0, int x0) {
        const int gz = z0 + wv;
        const long gbase = ((((long)n * D + gz) * H + y0) * W + x0) * C + 4 * q;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int dy = 2 * c4 + (u >> 1), dx = 4 * (u & 1) + kk;
            const bool ok = qok && gz < D && y0 + dy < H && x0 + dx < W;
            // masked lanes read the start of the tensor (no branch around the load) and are
            // zeroed where the value is used, so nothing waits for the load here
            const long e = ok ? gbase + (long)(dy * W + dx) * C : 0;
            dst[u] = thin_ld4<GBF>(g, e);
            if (SB) bd[u] = sbits[e >> 2];              // the decision byte of (this voxel, channel quad q)
        }
    };
    float4 av[2][4];
    uint32_t ab[2][4];
    int n = 0, z0 = 0, y0 = 0, x0 = 0;                 // the brick being multiplied
    if (b0 < b1) {
        halo1_fetch(a, hr, (int)b0, nbd, nbh, nbw, D, H, W);
        brick_coords((int)b0, nbd, nbh, nbw, n, z0, y0, x0);
        load4(av[0], ab[0], 0, n, z0, y0, x0);
    }
    for (long br = b0; br < b1; ++br, buf ^= 1) {
        int nn = 0, nz0 = 0, ny0 = 0, nx0 = 0;          // the brick after it
        // two halo images: a wave still multiplying the previous brick reads the other one
        halo1_put(hal[buf], hr);
        __syncthreads();
        if (br + 1 < b1) halo1_fetch(a, hr, (int)br + 1, nbd, nbh, nbw, D, H, W);
        __builtin_amdgcn_sched_barrier(0);
        const bool zin = z0 + wv < D;                   // a plane below the volume contributes zeros
        const float* hp = hal[buf] + wv * TH_H * TH_W + kk;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            if (c4 < 3) {
                load4(av[(c4 + 1) & 1], ab[(c4 + 1) & 1], c4 + 1, n, z0, y0, x0);
            } else if (br + 1 < b1) {
                brick_coords((int)br + 1, nbd, nbh, nbw, nn, nz0, ny0, nx0);
                load4(av[0], ab[0], 0, nn, nz0, ny0, nx0);
            }
            __builtin_amdgcn_sched_barrier(0);          // keep the loads ahead of the products
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int dy = 2 * c4 + (u >> 1), dx0 = 4 * (u & 1);
                float bv[2];
#pragma unroll
                for (int cg = 0; cg < 2; ++cg)
                    bv[cg] = btap[cg] ? hp[dy * TH_W + dx0 + toff[cg]] : bconst[cg];
                const bool ok = qok && zin && y0 + dy < H && x0 + dx0 + kk < W;
                float4 v = av[c4 & 1][u];
                if (BN)
                    v = make_float4(bn_act1(v.x, ca.x, cb.x, bn.slope), bn_act1(v.y, ca.y, cb.y, bn.slope),
                                    bn_act1(v.z, ca.z, cb.z, bn.slope), bn_act1(v.w, ca.w, cb.w, bn.slope));
                if (SB) {
                    // the order of stem_bn_k's backward passes: mask and scale, then the slope
                    const uint32_t b = ab[c4 & 1][u];
                    v.x = ((b & 1u) ? v.x * dscale : 0.f) * ((b & 16u) ? 1.f : bn.slope);
                    v.y = ((b & 2u) ? v.y * dscale : 0.f) * ((b & 32u) ? 1.f : bn.slope);
                    v.z = ((b & 4u) ? v.z * dscale : 0.f) * ((b & 64u) ? 1.f : bn.slope);
                    v.w = ((b & 8u) ? v.w * dscale : 0.f) * ((b & 128u) ? 1.f : bn.slope);
                }
                v = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
#pragma unroll
                for (int cg = 0; cg < 2; ++cg) {
                    acc[0][cg] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.x, bv[cg], acc[0][cg], 0, 0, 0);
                    acc[1][cg] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.y, bv[cg], acc[1][cg], 0, 0, 0);
                    acc[2][cg] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.z, bv[cg], acc[2][cg], 0, 0, 0);
                    acc[3][cg] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.w, bv[cg], acc[3][cg], 0, 0, 0);
                }
                if (SB) {
                    // two voxel pairs of the plane per group: rows 2 c4 + (u >> 1), x = 4 (u & 1) + 2 i + k32
                    const float* rp = hal[buf] + wv * TH_H * TH_W + roff;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int rx = 4 * (u & 1) + 2 * i + k32;
                        float pv = rp[dy * TH_W + rx];
                        pv = t32 < 27 ? pv : (t32 == 27 ? 1.f : 0.f);
                        pv = (zin && y0 + dy < H && x0 + rx < W) ? pv : 0.f;
                        racc = __builtin_amdgcn_mfma_f32_32x32x2f32(pv, pv, racc, 0, 0, 0);
                    }
                }
            }
        }
        n = nn; z0 = nz0; y0 = ny0; x0 = nx0;
    }
    // combine the four waves in fixed order: wave w adds its tiles into the LDS image in round w.
    // Lane (q, kk), register e of acc[j][cg] holds channel 4 (4 kk + e) + j under tap q + 16 cg.
    float* pp = part + (long)blockIdx.x * C * 28;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wv == w) {
#pragma unroll
            for (int cg = 0; cg < 2; ++cg) {
                const int t = q + 16 * cg;
                if (t < 28) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int c = 4 * (4 * kk + e) + j;
                            red[c * 28 + t] = w == 0 ? acc[j][cg][e] : red[c * 28 + t] + acc[j][cg][e];
                        }
                }
            }
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < C * 28; idx += 256) pp[idx] = red[idx];
    if (SB) {       // the tap correlation, the four waves in fixed order: register r of lane (t, k) is row (r & 3) + 8 (r >> 2) + 4 k
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            __syncthreads();
            if (wv == w) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * k32;
                    red[row * 32 + t32] = w == 0 ? racc[r] : red[row * 32 + t32] + racc[r];
                }
            }
        }
        __syncthreads();
        float* rr = rpart + (long)blockIdx.x * 1024;
        for (int idx = threadIdx.x; idx < 1024; idx += 256) rr[idx] = red[idx];
    }
}

// mode 0 (stem): dw[c*27+tap] = T[c][tap], dbias[c] = T[c][27]
// mode 1 (tail): dw[(26-tap)*C+c] = T[c][tap]
// One wave per output element: lanes stride over the per-block partials, fixed-order wave sum.
__global__ __launch_bounds__(256) void thin_wg_reduce(const float* __restrict__ part, int nblk, int C,
                                                      int mode, float* __restrict__ dw,
                                                      float* __restrict__ dbias, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= C * 28) return;
    const int c = i / 28, t = i % 28;
    double s = 0.0;
    for (int b = lane; b < nblk; b += 64) s += (double)part[(long)b * C * 28 + i];
    s = wave_sum_d(s);
    if (lane != 0) return;
    float* dst = nullptr;
    if (t < 27) dst = mode == 0 ? dw + c * 27 + t : dw + (26 - t) * C + c;
    else if (mode == 0 && dbias) dst = dbias + c;
    if (dst) *dst = accumulate ? *dst + (float)s : (float)s;
}

// ---------------------------------------------------------------------------------
// Weight-only backward of the encoder stem (Conv3d(1, C, 3) -> BatchNorm -> LeakyReLU -> Dropout) without
// the gradient of the convolution output ever being formed.  With dz = dout * keep / (1 - p) * act'(z),
// p[v][t] = a[v + tap t - 1] (zero padded), y = w p + bias, xhat = (y - mean) * inv, BatchNorm's backward is
//   dy = ca * (dz - S / M - xhat * Q / M),   S = sum_v dz,   Q = sum_v dz * xhat,
// and the convolution's weight gradient sum_v dy[v][c] p[v][t] is linear in it, so per channel c
//   T[c][t]  = sum_v dz[v][c] p[v][t]                 (one pass over dout: thin_wg_mfma_k<.., SB>)
//   Q        = inv * (sum_t w[c][t] T[c][t] + (bias - mean) S)
//   X[c][t]  = sum_v xhat[v][c] p[v][t] = inv * (sum_t' w[c][t'] R[t'][t] + (bias - mean) A[t])
//   dW[c][t] = ca * (T[c][t] - S / M * A[t] - Q / M * X[c][t])
// where R[t'][t] = sum_v p[v][t'] p[v][t] and A[t] = sum_v p[v][t] depend on the single-channel input alone:
// P'^T P' with P'[v] = (the 27 taps, 1, 0, 0, 0, 0), 32x32x2 matrix instructions whose two operands are the same
// register, formed by the same pass from the halo images it stages anyway (as a kernel of its own it took
// 0.11 ms; inside the HBM-bound pass it is free).  The sums over the per-block partials and the combination
// run in double precision.
// ---------------------------------------------------------------------------------

// Td[i] = sum_b tpart[b][i] (i < C * 28), Rd[i] = sum_b rpart[b][i] (i < 1024): one wave per element, double
__global__ __launch_bounds__(256) void stem_wg_reduce_k(const float* __restrict__ tpart, int nblk_t, int nt,
                                                        const float* __restrict__ rpart, int nblk_r,
                                                        double* __restrict__ Td, double* __restrict__ Rd) {
    const int lane = threadIdx.x & 63;
    int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const float* src = tpart;
    int nblk = nblk_t, n = nt;
    double* dst = Td;
    if (i >= nt) {
        i -= nt;
        src = rpart; nblk = nblk_r; n = 1024; dst = Rd;
        if (i >= 1024) return;
    }
    double s = 0.0;
    for (int b = lane; b < nblk; b += 64) s += (double)src[(long)b * n + i];
    s = wave_sum_d(s);
    if (lane == 0) dst[i] = s;
}

// one wave per channel: lanes 0..26 the taps, lane 27 the bias and the affine gradients
__global__ __launch_bounds__(64) void stem_wg_final_k(const double* __restrict__ Td, const double* __restrict__ Rd,
                                                      const float* __restrict__ w, const float* __restrict__ bias,
                                                      const float* __restrict__ mean, const float* __restrict__ var,
                                                      const float* __restrict__ gamma, float eps, double M,
                                                      float* __restrict__ dw, float* __restrict__ dbias,
                                                      float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                      int accumulate) {
    const int c = blockIdx.x, t = threadIdx.x;
    if (t > 27) return;
    // the constants of the kernels that apply this BatchNorm (fp32, their association)
    const double inv = (double)(1.0f / sqrtf(var[c] + eps));
    const double ca = (double)(gamma[c] / sqrtf(var[c] + eps));
    const double bm = (double)(bias ? bias[c] : 0.f) - (double)mean[c];
    const double* T = Td + c * 28;
    const double S = T[27];
    double wT = 0.0;
    for (int u = 0; u < 27; ++u) wT += (double)w[c * 27 + u] * T[u];
    const double Q = inv * (wT + bm * S);
    const double k1 = S / M, k2 = Q / M;
    if (t < 27) {
        double wR = 0.0;
        for (int u = 0; u < 27; ++u) wR += (double)w[c * 27 + u] * Rd[u * 32 + t];
        const double At = Rd[27 * 32 + t];
        const double X = inv * (wR + bm * At);
        const float r = (float)(ca * (T[t] - k1 * At - k2 * X));
        if (dw) dw[c * 27 + t] = accumulate ? dw[c * 27 + t] + r : r;
    } else {
        double wA = 0.0;
        for (int u = 0; u < 27; ++u) wA += (double)w[c * 27 + u] * Rd[27 * 32 + u];
        const double sx = inv * (wA + bm * M);                  // sum of xhat: zero up to the rounding of the mean
        const float r = (float)(ca * (S - k1 * M - k2 * sx));
        if (dbias) dbias[c] = accumulate ? dbias[c] + r : r;
        if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)Q : (float)Q;
        if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)S : (float)S;
    }
}

// sum of a flat array (gradient of the single-channel bias), two deterministic stages
__global__ __launch_bounds__(256) void flat_sum_partial(const float* __restrict__ a, long n,
                                                        float* __restrict__ part) {
    __shared__ float sh[4];
    float acc = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long)gridDim.x * blockDim.x)
        acc += a[i];
    acc = block_sum_256(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}
// one wave: lanes stride over the partials, fixed-order wave sum (a single thread took 50 us for 1024 partials)
__global__ __launch_bounds__(64) void flat_sum_final(const float* __restrict__ part, int nblk, float* __restrict__ out,
                                                     int accumulate) {
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 64) s += (double)part[b];
    s = wave_sum_d(s);
    if (threadIdx.x == 0) *out = accumulate ? *out + (float)s : (float)s;
}

bool k3_thin_supported(int Cin, int Cout) {
    if (Cin == 1) return Cout % 4 == 0 && Cout >= 4 && Cout <= 256;
    if (Cout == 1) return Cin % 8 == 0 && Cin >= 8 && Cin <= 256;
    return false;
}

// number of blocks (= rows of per-block statistics) of the 1 -> C matrix-pipe kernel, 0 if it does not apply
static inline void thin_1c_grid(int N, int D, int H, int W, long& nblk, int& bpb) {
    const int nbd = (D + TB_D - 1) / TB_D, nbh = (H + TB_H - 1) / TB_H, nbw = (W + TB_W - 1) / TB_W;
    nblk = (long)N * nbd * nbh * nbw;
    // four bricks per block measured best (2, 8: +3 %; one round of 19-brick blocks: +12 %)
    bpb = (int)cdiv(nblk, 4096) < 4 ? (int)cdiv(nblk, 4096) : 4;
}
long k3_thin_fwd_stat_blocks(int N, int D, int H, int W, int Ca, int Cb) {
    if (Ca != 1 || Cb > 64) return 0;
    long nblk;
    int bpb;
    thin_1c_grid(N, D, H, W, nblk, bpb);
    return nblk > 0x7fffffffL ? 0 : cdiv(nblk, bpb);
}

// 12 x 16 column walk of the 64 -> 1 kernel: applies to 64 channels on volumes of at least 16 x 12 x 16
// unless 12-row columns pad H by 10 % more than 8-row ones; two blocks per CU
struct ZxPlan {
    int nth, ntw, zsplit;
    long nb;
};
static bool zx_plan(int N, int D, int H, int W, int Ca, ZxPlan& p) {
    if (!(Ca == 64 && D >= 16 && H >= 12 && W >= 16 && (long)H * W * Ca < (1L << 30) &&
          cdiv(H, ZX_H) * ZX_H * 10 <= cdiv(H, ZW_H) * ZW_H * 11))
        return false;
    p.nth = (H + ZX_H - 1) / ZX_H, p.ntw = (W + ZX_W - 1) / ZX_W;
    // z segments: each repeats two planes, and the two blocks of a CU share its matrix pipe --
    // take the split with the fewest plane steps on the fullest CU
    p.zsplit = 1;
    long best = -1;
    for (int zs = 1; zs <= D / 8; ++zs) {
        const long rounds = cdiv((long)N * p.nth * p.ntw * zs, THIN_CUS);
        const long cost = rounds * (cdiv(D, zs) + 2);
        if (best < 0 || cost < best) best = cost, p.zsplit = zs;
    }
    p.nb = (long)N * p.nth * p.ntw * p.zsplit;
    return p.nb <= 0x7fffffffL;
}

// C -> 1 forward with BatchNorm + LeakyReLU of the layer in front applied to the operand (x = that layer's
// raw convolution output).  Only the column-walking kernel has this form.
bool k3_thin_bn_fwd_supported(int N, int D, int H, int W, int Cin) {
    ZxPlan zx;
    return zx_plan(N, D, H, W, Cin, zx);
}
int k3_thin_bn_fwd_launch(const float* x, const BnPrologue& bn, const float* w, const float* bias, float* out,
                          int N, int D, int H, int W, int Cin, hipStream_t s, void* x_bf16) {
    ZxPlan zx;
    if (!zx_plan(N, D, H, W, Cin, zx)) {
        set_error("bnact_conv3d_k3_to1: unsupported shape (needs 64 channels, D >= 16, H >= 12, W >= 16)");
        return SIV_EINVAL;
    }
    if (x_bf16)
        c_to_one_zwalk12_k<8, true, true><<<(unsigned)zx.nb, 256, 0, s>>>(x, w, bias, out, N, D, H, W, 0, zx.nth, zx.ntw,
                                                                         zx.zsplit, bn, (uint16_t*)x_bf16);
    else
        c_to_one_zwalk12_k<8, true, false><<<(unsigned)zx.nb, 256, 0, s>>>(x, w, bias, out, N, D, H, W, 0, zx.nth, zx.ntw,
                                                                          zx.zsplit, bn, nullptr);
    SIV_CHECK_LAUNCH("bnact_conv3d_k3_to1(fwd)");
    return SIV_OK;
}

int bn_bwd_finalize_launch(const float* part, int nblk, int C, float* sums, float* dgamma, float* dbeta,
                           int accumulate, hipStream_t s);

// dx (= gradient of the BatchNorm input), dgamma, dbeta of [BatchNorm + LeakyReLU] -> Conv3d(Cin, 1, 3)
// from the convolution's output gradient g1[N][D][H][W]
int k3_thin_bn_dgrad_launch(const float* g1, const float* w, const float* x, const BnPrologue& bn, float* dx,
                            float* dgamma, float* dbeta, int accumulate, int N, int D, int H, int W, int Cin,
                            void* ws, size_t ws_bytes, hipStream_t s, bool x_bf16) {
    if (Cin % 4 != 0 || Cin > 64) {
        set_error("bnact_conv3d_k3_to1(dgrad): needs Cin %% 4 == 0 and Cin <= 64 (got %d)", Cin);
        return SIV_EINVAL;
    }
    const int nbd = (D + TB_D - 1) / TB_D, nbh = (H + TB_H - 1) / TB_H, nbw = (W + TB_W - 1) / TB_W;
    long nbricks;
    int bpb;
    thin_1c_grid(N, D, H, W, nbricks, bpb);
    if (nbricks > 0x7fffffffL) {
        set_error("bnact_conv3d_k3_to1(dgrad): too many bricks");
        return SIV_EINVAL;
    }
    const long nblk = cdiv(nbricks, bpb);
    const size_t part_bytes = align256((size_t)nblk * 2 * Cin * sizeof(float));
    SIV_CHECK_WS(part_bytes + (size_t)2 * Cin * sizeof(float), ws_bytes, "bnact_conv3d_k3_to1(dgrad)");
    float* part = (float*)ws;
    float* sums = (float*)((char*)ws + part_bytes);
    const float invM = 1.0f / (float)((long)N * D * H * W);
    if (x_bf16)
        tail_bn_bwd_k<0, true><<<(unsigned)nblk, 256, 0, s>>>(g1, w, x, bn, nullptr, invM, part, nullptr, N, D, H, W, Cin,
                                                             nbd, nbh, nbw, bpb, (int)nbricks);
    else
        tail_bn_bwd_k<0, false><<<(unsigned)nblk, 256, 0, s>>>(g1, w, x, bn, nullptr, invM, part, nullptr, N, D, H, W, Cin,
                                                              nbd, nbh, nbw, bpb, (int)nbricks);
    SIV_CHECK_LAUNCH("bnact_conv3d_k3_to1(dgrad, sums)");
    const int rc = bn_bwd_finalize_launch(part, (int)nblk, Cin, sums, dgamma, dbeta, accumulate, s);
    if (rc) return rc;
    if (x_bf16)
        tail_bn_bwd_k<1, true><<<(unsigned)nblk, 256, 0, s>>>(g1, w, x, bn, sums, invM, nullptr, dx, N, D, H, W, Cin, nbd,
                                                             nbh, nbw, bpb, (int)nbricks);
    else
        tail_bn_bwd_k<1, false><<<(unsigned)nblk, 256, 0, s>>>(g1, w, x, bn, sums, invM, nullptr, dx, N, D, H, W, Cin, nbd,
                                                              nbh, nbw, bpb, (int)nbricks);
    SIV_CHECK_LAUNCH("bnact_conv3d_k3_to1(dgrad, apply)");
    return SIV_OK;
}

int bn_finalize_launch(const float* shift, const float* part, int nblk, long M, int C, float* mean,
                       float* var, float* running_mean, float* running_var, int64_t* nbt,
                       float momentum, hipStream_t s);

struct StemGrid {
    int nbd, nbh, nbw, bpb;
    long nbricks, nblk;
    size_t part_bytes;
};
static int stem_grid(const char* what, int N, int D, int H, int W, int C, size_t ws_bytes, StemGrid& g) {
    if (C % 4 != 0 || C > 64) {
        set_error("%s: needs Cout %% 4 == 0 and Cout <= 64 (got %d)", what, C);
        return SIV_EINVAL;
    }
    g.nbd = (D + TB_D - 1) / TB_D, g.nbh = (H + TB_H - 1) / TB_H, g.nbw = (W + TB_W - 1) / TB_W;
    thin_1c_grid(N, D, H, W, g.nbricks, g.bpb);
    if (g.nbricks > 0x7fffffffL || (long)N * D * H * W * C > 0x7fffffffffL) {
        set_error("%s: volume too large", what);
        return SIV_EINVAL;
    }
    g.nblk = cdiv(g.nbricks, g.bpb);
    g.part_bytes = align256((size_t)g.nblk * 2 * C * sizeof(float));
    SIV_CHECK_WS(g.part_bytes + (size_t)2 * C * sizeof(float), ws_bytes, what);
    return SIV_OK;
}

// Forward of Conv3d(1, C, 3) -> BatchNorm -> LeakyReLU -> Dropout.  train: batch statistics are computed
// (mean, var written, running statistics updated); otherwise mean / var are used as given.
int stem_bnact_fwd_launch(const float* x, const float* w, const float* bias, const float* gamma, const float* beta,
                          const uint8_t* keep, const PhiloxMask& rk, float p, float slope, float eps, int train,
                          float* mean, float* var, float* running_mean, float* running_var, int64_t* nbt,
                          float momentum, float* out, int N, int D, int H, int W, int C, void* ws, size_t ws_bytes,
                          hipStream_t s, void* out_bf16, void* bits) {
    StemGrid g;
    int rc = stem_grid("conv1_bnact_fwd", N, D, H, W, C, ws_bytes, g);
    if (rc) return rc;
    const BnPrologue bn{mean, var, gamma, beta, eps, slope};
    if (train) {
        float* part = (float*)ws;
        stem_bn_k<SM_STATS><<<(unsigned)g.nblk, 256, 0, s>>>(x, w, bias, bn, nullptr, PhiloxMask{}, 0.f, nullptr, nullptr, 0.f, part,
                                                            nullptr, N, D, H, W, C, g.nbd, g.nbh, g.nbw, g.bpb,
                                                            (int)g.nbricks, nullptr);
        SIV_CHECK_LAUNCH("conv1_bnact_fwd(statistics)");
        rc = bn_finalize_launch(bias, part, (int)g.nblk, (long)N * D * H * W, C, mean, var, running_mean, running_var,
                                nbt, momentum, s);
        if (rc) return rc;
    }
#define SIV_STEM_APPLY(...)                                                                                              \
    stem_bn_k<SM_APPLY, __VA_ARGS__><<<(unsigned)g.nblk, 256, 0, s>>>(x, w, bias, bn, keep, rk, p, nullptr, nullptr, 0.f, nullptr, \
                                                                     out, N, D, H, W, C, g.nbd, g.nbh, g.nbw, g.bpb,        \
                                                                     (int)g.nbricks, (uint16_t*)out_bf16, (uint8_t*)bits)
    if (bits && rk.on && !keep && !out_bf16)            // training on the device generator, fp32 storage
        SIV_STEM_APPLY(true, 2, 0);
    else if (bits && rk.on && !keep)                    // ... bf16 storage
        SIV_STEM_APPLY(true, 2, 1);
    else if (bits)
        SIV_STEM_APPLY(true);
#undef SIV_STEM_APPLY
    else
        stem_bn_k<SM_APPLY><<<(unsigned)g.nblk, 256, 0, s>>>(x, w, bias, bn, keep, rk, p, nullptr, nullptr, 0.f, nullptr, out, N,
                                                            D, H, W, C, g.nbd, g.nbh, g.nbw, g.bpb, (int)g.nbricks,
                                                            (uint16_t*)out_bf16);
    SIV_CHECK_LAUNCH("conv1_bnact_fwd(apply)");
    return SIV_OK;
}

// Backward of the same down to the convolution output: dy (gradient of the never-stored y), dgamma, dbeta.
int stem_bnact_bwd_launch(const float* dout, const float* x, const float* w, const float* bias, const float* mean,
                          const float* var, const float* gamma, const float* beta, const uint8_t* keep,
                          const PhiloxMask& rk, float p, float slope, float eps, float* dy, float* dgamma,
                          float* dbeta, int accumulate, int N, int D, int H, int W, int C, void* ws, size_t ws_bytes,
                          hipStream_t s, const void* bits) {
    StemGrid g;
    int rc = stem_grid("conv1_bnact_bwd", N, D, H, W, C, ws_bytes, g);
    if (rc) return rc;
    const BnPrologue bn{mean, var, gamma, beta, eps, slope};
    float* part = (float*)ws;
    float* sums = (float*)((char*)ws + g.part_bytes);
    const float invM = 1.0f / (float)((long)N * D * H * W);
    uint8_t* bt = (uint8_t*)bits;           // read only by the backward instantiations
    if (bt)
        stem_bn_k<SM_BSUM, true><<<(unsigned)g.nblk, 256, 0, s>>>(x, w, bias, bn, nullptr, PhiloxMask{}, p, dout, nullptr, invM,
                                                                 part, nullptr, N, D, H, W, C, g.nbd, g.nbh, g.nbw, g.bpb,
                                                                 (int)g.nbricks, nullptr, bt);
    else
        stem_bn_k<SM_BSUM><<<(unsigned)g.nblk, 256, 0, s>>>(x, w, bias, bn, keep, rk, p, dout, nullptr, invM, part, nullptr, N,
                                                           D, H, W, C, g.nbd, g.nbh, g.nbw, g.bpb, (int)g.nbricks, nullptr);
    SIV_CHECK_LAUNCH("conv1_bnact_bwd(sums)");
    rc = bn_bwd_finalize_launch(part, (int)g.nblk, C, sums, dgamma, dbeta, accumulate, s);
    if (rc) return rc;
    if (bt)
        stem_bn_k<SM_BAPP, true><<<(unsigned)g.nblk, 256, 0, s>>>(x, w, bias, bn, nullptr, PhiloxMask{}, p, dout, sums, invM,
                                                                 nullptr, dy, N, D, H, W, C, g.nbd, g.nbh, g.nbw, g.bpb,
                                                                 (int)g.nbricks, nullptr, bt);
    else
        stem_bn_k<SM_BAPP><<<(unsigned)g.nblk, 256, 0, s>>>(x, w, bias, bn, keep, rk, p, dout, sums, invM, nullptr, dy, N, D, H,
                                                           W, C, g.nbd, g.nbh, g.nbw, g.bpb, (int)g.nbricks, nullptr);
    SIV_CHECK_LAUNCH("conv1_bnact_bwd(apply)");
    return SIV_OK;
}

// Weight-only backward of the same (dw, dbias of the convolution, dgamma, dbeta) from the gradient behind the
// dropout and the bits of the forward pass: one pass over dout, one over the single-channel input; the
// gradient of the convolution output is never formed (algebra at stem_wg_reduce_k).
int stem_bnact_wgrad_launch(const float* dout, const float* x, const float* w, const float* bias, const float* mean,
                            const float* var, const float* gamma, const void* bits, float p, float slope, float eps,
                            float* dw, float* dbias, float* dgamma, float* dbeta, int accumulate, int N, int D,
                            int H, int W, int C, void* ws, size_t ws_bytes, hipStream_t s) {
    if (C % 4 != 0 || C > 64) {
        set_error("conv1_bnact_wgrad: needs Cout %% 4 == 0 and Cout <= 64 (got %d)", C);
        return SIV_EINVAL;
    }
    const int nbd = (D + TB_D - 1) / TB_D, nbh = (H + TB_H - 1) / TB_H, nbw = (W + TB_W - 1) / TB_W;
    const long nbricks = (long)N * nbd * nbh * nbw;
    if (nbricks > 0x7fffffffL || (long)N * D * H * W * C > 0x7fffffffffL) {
        set_error("conv1_bnact_wgrad: volume too large");
        return SIV_EINVAL;
    }
    long nblk = nbricks < 4 * THIN_CUS ? nbricks : 4 * THIN_CUS;      // one round of four blocks per CU
    const int bpb = (int)cdiv(nbricks, nblk);
    nblk = cdiv(nbricks, bpb);
    const size_t tpart_bytes = align256((size_t)nblk * C * 28 * sizeof(float));
    const size_t rpart_bytes = align256((size_t)nblk * 1024 * sizeof(float));
    const size_t td_bytes = align256((size_t)C * 28 * sizeof(double));
    SIV_CHECK_WS(tpart_bytes + rpart_bytes + td_bytes + 1024 * sizeof(double), ws_bytes, "conv1_bnact_wgrad");
    float* tpart = (float*)ws;
    float* rpart = (float*)((char*)ws + tpart_bytes);
    double* Td = (double*)((char*)ws + tpart_bytes + rpart_bytes);
    double* Rd = (double*)((char*)ws + tpart_bytes + rpart_bytes + td_bytes);
    BnPrologue bn{};
    bn.slope = slope;
    // one pass: T from dout and the bits, the input's tap correlation from the halo images it stages anyway
    thin_wg_mfma_k<false, false, true><<<(unsigned)nblk, 256, 0, s>>>(x, dout, tpart, N, D, H, W, C, nbd, nbh, nbw, bpb, bn,
                                                                     (const uint8_t*)bits, 1.0f / (1.0f - p), rpart);
    SIV_CHECK_LAUNCH("conv1_bnact_wgrad(gradient pass)");
    stem_wg_reduce_k<<<(C * 28 + 1024 + 3) / 4, 256, 0, s>>>(tpart, (int)nblk, C * 28, rpart, (int)nblk, Td, Rd);
    SIV_CHECK_LAUNCH("conv1_bnact_wgrad(reduce)");
    stem_wg_final_k<<<C, 64, 0, s>>>(Td, Rd, w, bias, mean, var, gamma, eps, (double)((long)N * D * H * W), dw, dbias,
                                     dgamma, dbeta, accumulate);
    SIV_CHECK_LAUNCH("conv1_bnact_wgrad(final)");
    return SIV_OK;
}

int k3_thin_fwd_launch(const float* a, const float* w, const float* bias, float* out, int N, int D,
                       int H, int W, int Ca, int Cb, int flip, hipStream_t s, float* stat_part) {
    const int nbd = (D + TB_D - 1) / TB_D, nbh = (H + TB_H - 1) / TB_H, nbw = (W + TB_W - 1) / TB_W;
    const long nblk = (long)N * nbd * nbh * nbw;
    if (nblk > 0x7fffffffL) {
        set_error("conv3d_k3(thin): too many bricks");
        return SIV_EINVAL;
    }
    // several bricks per block amortise the per-block weight loads; keep >= ~2000 blocks
    int bpb = (int)(nblk / 2048);
    if (bpb < 1) bpb = 1;
    if (bpb > 8) bpb = 8;
    const int nbr = (int)nblk;
    if (Ca == 1 && Cb <= 64) {
        long nb;
        thin_1c_grid(N, D, H, W, nb, bpb);
        const unsigned grid = (unsigned)cdiv(nblk, bpb);
        one_to_c_mfma_k<<<grid, 256, 0, s>>>(a, w, bias, out, N, D, H, W, Cb, flip, nbd, nbh, nbw, bpb, nbr,
                                             stat_part);
        SIV_CHECK_LAUNCH("conv3d_k3(thin, matrix)");
        return SIV_OK;
    }
    if (stat_part) {
        set_error("conv3d_k3(thin): this kernel emits no statistics");
        return SIV_EINVAL;
    }
    if (Ca == 1) {
        one_to_c_k<<<(unsigned)cdiv(nblk, bpb), 256, 0, s>>>(a, w, bias, out, N, D, H, W, Cb, flip, nbd, nbh,
                                                             nbw, bpb, nbr);
        SIV_CHECK_LAUNCH("conv3d_k3(thin)");
        return SIV_OK;
    }
    // the matrix-core kernel runs one brick per block: with several, the barrier between the GEMM and
    // the gather phase of a brick serialises what independent blocks otherwise overlap (measured)
    const unsigned grid = (unsigned)nblk;
    ZxPlan zx;
    if (zx_plan(N, D, H, W, Ca, zx)) {
        c_to_one_zwalk12_k<8, false><<<(unsigned)zx.nb, 256, 0, s>>>(a, w, bias, out, N, D, H, W, flip, zx.nth,
                                                                    zx.ntw, zx.zsplit, BnPrologue{}, nullptr);
        SIV_CHECK_LAUNCH("conv3d_k3(thin, z-walk 12x16)");
        return SIV_OK;
    }
    if (Ca == 64 && D >= 16 && H >= 8 && W >= 16) {
        // long volumes: z-walking columns (less halo traffic); cut along z until ~1000 blocks
        const int nth = (H + ZW_H - 1) / ZW_H, ntw = (W + ZW_W - 1) / ZW_W;
        int zsplit = 1;
        while ((long)N * nth * ntw * zsplit < 900 && D / (zsplit * 2) >= 8) zsplit *= 2;
        const long nb = (long)N * nth * ntw * zsplit;
        if (nb <= 0x7fffffffL && (long)H * W * Ca < (1L << 30)) {
            c_to_one_zwalk_k<8><<<(unsigned)nb, 256, 0, s>>>(a, w, bias, out, N, D, H, W, flip, nth, ntw, zsplit);
            SIV_CHECK_LAUNCH("conv3d_k3(thin, z-walk)");
            return SIV_OK;
        }
    }
    if (Ca == 1)
        ;
    else if (Ca == 64)
        c_to_one_mfma_k<8><<<grid, 256, 0, s>>>(a, w, bias, out, N, D, H, W, flip, nbd, nbh, nbw);
    else if (Ca == 32)
        c_to_one_mfma_k<4><<<grid, 256, 0, s>>>(a, w, bias, out, N, D, H, W, flip, nbd, nbh, nbw);
    else if (Ca == 16)
        c_to_one_mfma_k<2><<<grid, 256, 0, s>>>(a, w, bias, out, N, D, H, W, flip, nbd, nbh, nbw);
    else if (Ca == 8)
        c_to_one_mfma_k<1><<<grid, 256, 0, s>>>(a, w, bias, out, N, D, H, W, flip, nbd, nbh, nbw);
    else
        c_to_one_k<<<(unsigned)nblk, 256, 0, s>>>(a, w, bias, out, N, D, H, W, Ca, flip, nbd, nbh, nbw);
    SIV_CHECK_LAUNCH("conv3d_k3(thin)");
    return SIV_OK;
}

int k3_thin_wgrad_launch(const float* x, const float* dy, float* dw, float* dbias, int N, int D, int H,
                         int W, int Cin, int Cout, int accumulate, void* ws, size_t ws_bytes,
                         hipStream_t s, const BnPrologue* bn, bool x_bf16) {
    const int nbd = (D + TB_D - 1) / TB_D, nbh = (H + TB_H - 1) / TB_H, nbw = (W + TB_W - 1) / TB_W;
    const long nbricks = (long)N * nbd * nbh * nbw;
    const int mode = (Cin == 1) ? 0 : 1;
    const int C = mode == 0 ? Cout : Cin;
    if (C % 4 != 0 || C > 256) {
        set_error("conv3d_k3_wgrad(thin): unsupported channel count %d", C);
        return SIV_EINVAL;
    }
    const bool mfma = C <= 64;
    const long cap = mfma ? 4 * THIN_CUS : 1024;      // one round of four blocks per CU
    long nblk = nbricks < cap ? nbricks : cap;
    const int bpb = (int)cdiv(nbricks, nblk);
    nblk = cdiv(nbricks, bpb);
    const size_t part_bytes = align256((size_t)nblk * C * 28 * sizeof(float));
    SIV_CHECK_WS(part_bytes + 1024 * sizeof(float), ws_bytes, "conv3d_k3_wgrad(thin)");
    float* part = (float*)ws;
    const float* a = mode == 0 ? x : dy;
    const float* g = mode == 0 ? dy : x;
    if (bn && !(mfma && mode == 1)) {
        set_error("bnact_conv3d_k3_to1(wgrad): needs at most 64 input channels");
        return SIV_EINVAL;
    }
    if (x_bf16 && !(mfma && bn)) {
        set_error("conv3d_k3_wgrad(thin): a bf16 operand is read only by the BatchNorm-prologue form");
        return SIV_EINVAL;
    }
    if (mfma && bn && x_bf16)
        thin_wg_mfma_k<true, true><<<(unsigned)nblk, 256, 0, s>>>(a, g, part, N, D, H, W, C, nbd, nbh, nbw, bpb, *bn);
    else if (mfma && bn)
        thin_wg_mfma_k<true, false><<<(unsigned)nblk, 256, 0, s>>>(a, g, part, N, D, H, W, C, nbd, nbh, nbw, bpb, *bn);
    else if (mfma)
        thin_wg_mfma_k<false, false><<<(unsigned)nblk, 256, 0, s>>>(a, g, part, N, D, H, W, C, nbd, nbh, nbw, bpb,
                                                                   BnPrologue{});
    else
        thin_wg_k<<<(unsigned)nblk, 256, 0, s>>>(a, g, part, N, D, H, W, C, nbd, nbh, nbw, bpb);
    SIV_CHECK_LAUNCH("conv3d_k3_wgrad(thin)");
    thin_wg_reduce<<<(C * 28 + 3) / 4, 256, 0, s>>>(part, (int)nblk, C, mode, dw, dbias, accumulate);
    SIV_CHECK_LAUNCH("conv3d_k3_wgrad(thin reduce)");
    if (mode == 1 && dbias) {
        float* p2 = (float*)((char*)ws + part_bytes);
        const long n = (long)N * D * H * W;
        long gb = cdiv(n, 256 * 8);
        if (gb > 1024) gb = 1024;
        flat_sum_partial<<<(unsigned)gb, 256, 0, s>>>(dy, n, p2);
        SIV_CHECK_LAUNCH("conv3d_k3_wgrad(thin bias)");
        flat_sum_final<<<1, 64, 0, s>>>(p2, (int)gb, dbias, accumulate);
        SIV_CHECK_LAUNCH("conv3d_k3_wgrad(thin bias final)");
    }
    return SIV_OK;
}

}  // namespace siv
