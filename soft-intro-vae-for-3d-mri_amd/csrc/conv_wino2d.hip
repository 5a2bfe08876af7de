// 3x3x3 convolution, forward / input gradient, with the Winograd F(2x2, 3x3) minimal-filtering
// identity applied over (H, W) and the fp32 matrix cores doing the channel contraction.
//
// A 2x2 output tile of one z-plane depends on a 4x4 input patch of each of the three planes kd:
//   Y = A^T [ sum_kd (G g_kd G^T) .* (B^T d_kd B) ] A
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],  A^T = [1 1 1 0; 0 1 -1 -1]
// i.e. 16 positions x 3 planes = 48 multiply-adds per 4 outputs and channel pair: 12 per output
// instead of the 27 of the direct kernel or the 18 of the one-dimensional F(2,3) kernel.  Everything
// stays fp32; only the additions of the transforms change the rounding (the factors 1/2, 1/4 are exact).
//
//   M rows  = 2x2 output tiles of a 4x8x8-voxel brick (64 tiles = 2 row tiles), N = 64 output channels
//   K       = 3 planes x Cin per position; 16 positions = 16 independent accumulator sets
//   a block = 8 waves: wave (ph, wn) owns the four positions (ph, 0..3) of both row tiles for output
//             channel half wn (8 accumulators = 128 registers)
//
// The transformed input V[pos][plane][tile][ci] of one 8-channel chunk takes 48 KB of LDS and the epilogue
// 144 KB, so one block runs per CU and nothing else would cover its barriers: V is double buffered.  While the waves multiply
// chunk c out of one buffer, six of the eight waves cut the 4x4 patches of chunk c+1 out of a raw halo image
// in LDS (fetched one chunk earlier with coalesced 16-byte loads), transform them and fill the other buffer;
// one barrier per chunk.  The output transform along W is done on the accumulators, the one along H needs the four
// waves ph = 0..3 and goes through LDS together with the transposition for 16-byte stores.
#include "siv_common.h"

// SIV_W2EXP: timing-only ablation mask (never set in the library build; results are wrong with it):
// 1 = no staging of the next chunk, 2 = transform without the V writes, 4 = no patch reads.
// Measured with tools/ubench/wino2d_stamps (64 -> 64 @80x96x80, cycles per chunk, floor 12288):
// full 14165, no patch reads 13728, no staging 12820; the SIMDs that hold two staging waves set the pace.
#ifndef SIV_W2EXP
#define SIV_W2EXP 0
#endif

namespace siv {

// SIV_W2STAMP (diagnostic build only, tools/ubench/wino2d_stamps.hip): lane 0 of every wave of the first
// blocks records the shader clock at block entry, before and after every barrier of the chunk loop and at
// block exit, into a buffer of its own that nothing else reads.
#ifdef SIV_W2STAMP
__device__ unsigned long long w2_stamps[2048 * 8 * 32];
#define SIV_STAMP(i)                                                                                         \
    do {                                                                                                     \
        if (lane == 0 && blockIdx.x < 2048 && (i) < 32)                                                      \
            w2_stamps[((long)blockIdx.x * 8 + wave) * 32 + (i)] = __builtin_amdgcn_s_memtime();              \
    } while (0)
#else
#define SIV_STAMP(i)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));    // two channels per lane: one packed add per pair
constexpr unsigned W2_OUTSIDE = 0x80000000u;      // beyond any sample (the launcher keeps samples below 2 GiB)

struct W2Cfg {
    static constexpr int BD = 4, BH = 8, BW = 8;
    static constexpr int CK = 8;
    static constexpr int TPP = 16;                         // 4x4 tiles per plane
    static constexpr int HD = BD + 2;                      // planes incl. halo
    static constexpr int PLANE = HD * TPP * CK;            // floats of one position: 768
    static constexpr int V_FLOATS = 16 * PLANE;            // 12288 floats = 48 KB
    static constexpr int NTHR = 512, NT = 64;
    static constexpr int TAPS = 48, SLICE = NT * 8;        // (kd, ph, pw) slices of [64 co][8 ci]
#ifndef SIV_W2PD
#define SIV_W2PD 6
#endif
    static constexpr int PD = SIV_W2PD;                    // weight fragments fetched PD taps ahead (divides 12)
    static constexpr int EPI_STRIDE = 36, EPI_TILE = 32 * EPI_STRIDE;
    static constexpr int RAW_ITEMS = 3 * NTHR;             // 16-byte items of a raw halo image (1536 >= 2 * 600)
    static constexpr int RAW_FLOATS = RAW_ITEMS * 4;       // 6144 floats = 24 KB
    static constexpr int LDS_FLOATS = 2 * V_FLOATS + 2 * RAW_FLOATS;      // 144 KB: both V buffers, both raw images
    static constexpr size_t LDS_BYTES = (size_t)LDS_FLOATS * sizeof(float);
};

__device__ __forceinline__ float4 w2sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 w2add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float2 w2sub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 w2add(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }

// tap at which the staging of the next chunk starts (patch reads; the eight pieces follow over six taps)
#ifndef SIV_W2STAGE_T0
#define SIV_W2STAGE_T0 5
#endif
#ifndef SIV_W2ROUNDS
#define SIV_W2ROUNDS 1
#endif

// TPR = tiles per row of a brick plane: 4 (8 x 8 voxels, 16 tiles) or 5 (6 x 10 voxels, 15 tiles: slot 15
// empty).  ROUNDS = 1: a staging item is (tile-row, channel pair), 384 items = waves 0..5; ROUNDS = 2: an item
// is (tile-row, channel quad) handled as two pairs in turn, 192 items = waves 0..2.
//
// A block works through the bricks blockIdx.x, blockIdx.x + gridDim.x, ... (nbricks in all) as ONE stream of
// chunks: the raw image two chunks ahead and the transformed image one chunk ahead belong to the next brick
// when the current one runs out, so only the first brick of a block pays the prologue (a first global load
// that nothing covers, ~5 % of a brick).  Between two bricks the epilogue runs in two halves inside the 72 KB
// of LDS that the next brick's images do not occupy.  The launcher chains bricks only for an even chunk count
// of at least four (buffer parities then repeat per brick); otherwise gridDim.x = nbricks.
template <int TPR, int ROUNDS>
__global__ __launch_bounds__(512, 2) void k3_fwd_wino2d(const float* __restrict__ x,
                                                        const float* __restrict__ wp,
                                                        const float* __restrict__ bias,
                                                        float* __restrict__ y, int N, int D, int H, int W,
                                                        int Cin, int Cout, int nbd, int nbh, int nbw,
                                                        int nbricks, float* __restrict__ stat_part, int kshift) {
    using C = W2Cfg;
    constexpr int tpr = TPR, trows = 16 / TPR;
    constexpr int BHP = 2 * trows + 2, BWP = 2 * TPR + 2;      // halo plane: 10 x 10 or 8 x 12 voxels
    constexpr int NV = C::HD * BHP * BWP;                      // 600 or 576 halo voxels
    static_assert(2 * NV <= C::RAW_ITEMS, "three 16-byte items per thread cover the halo");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const rimg = lds + 2 * C::V_FLOATS;                 // two raw halo images
    // between two bricks V[0] holds the next brick's chunk 0 and raw image 1 its chunk 1: the epilogue has
    // V[1] and raw image 0, 72 KB in one piece = 16 exchange tiles
    float* const epi = lds + C::V_FLOATS;
    static_assert(C::V_FLOATS + C::RAW_FLOATS == 16 * C::EPI_TILE, "half of the exchange tiles fit between the live images");

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int ph = wave & 3, wn = wave >> 2;
    SIV_STAMP(0);

    // brick of a dispatch index: each XCD (index mod 8) walks a contiguous eighth of the bricks
    // (kshift > 0: an item is (brick, quarter of the input channels) and the block chooses its items itself,
    // below; see the launcher)
    auto item_of = [&](int b) { return !kshift && (nbricks & 7) == 0 ? (b & 7) * (nbricks >> 3) + (b >> 3) : b; };
    auto brick_of = [&](int b, int& bn, int& bz0, int& by0, int& bx0) {
        b = item_of(b) >> kshift;
        const int bx = b % nbw; b /= nbw;
        const int by = b % nbh; b /= nbh;
        bz0 = (b % nbd) * C::BD; by0 = by * 2 * trows; bx0 = bx * 2 * tpr;
        bn = b / nbd;
    };
    int vb = blockIdx.x, cot = blockIdx.y;
    // Channel contraction split into KP = 1 << kshift parts (2 or 4): item v = brick * KP + part; a block's
    // items vb, vb + gridDim.x, ... all take the same part of the input channels (gridDim.x is a multiple of 8)
    // and leave a bias-free partial output in slab kpart of y, which is then a buffer of KP outputs.  The 256 blocks of the launch are
    // dealt out so that an XCD (dispatch index mod 8) works on ONE output-channel tile and as few parts as
    // possible: the transformed weights of the layer (12.6 MB at 256 -> 256) do not fit an XCD's L2, the
    // 1.6 MB of one tile and two parts do.
    if (kshift) {
        const int lin = blockIdx.y * gridDim.x + blockIdx.x, xcd = lin & 7, j = lin >> 3;     // j: 0..31
        const int P = 8 / (int)gridDim.y, xc = xcd % P;        // XCDs per output-channel tile (8, 4, 2 or 1)
        const int KP = 1 << kshift;                            // channel parts (2 or 4)
        cot = xcd / P;
        int kp, slot;
        if (P >= KP) kp = xc % KP, slot = (xc / KP) * 32 + j;  // an XCD works on one part
        else kp = xc * (KP / P) + j % (KP / P), slot = j / (KP / P);
        vb = slot * KP + kp;
    }
    int n, z0, y0, x0;
    brick_of(vb, n, z0, y0, x0);
    const int co0 = cot * C::NT;
    const int kpart = vb & ((1 << kshift) - 1);
    const int nchunk = (Cin / C::CK) >> kshift;                // chunks of an item
    const int kc0 = kpart * nchunk * C::CK;                    // its first input channel
    if (kshift) {
        y += (long)kpart * N * D * H * W * Cout;
        bias = nullptr;
        stat_part = nullptr;
    }

    // A row m = a*32 + li is tile (dz = m >> 4, th, tw) and reads V row (dz + kd)*16 + tile = m + 16*kd.
    // A V row holds the 8 channels of a chunk without padding; the two 16-byte halves of the rows of an odd
    // halo plane are swapped, which spreads the 16 lanes that share an LDS cycle over all 64 banks (lanes li
    // and li + 16 meet in one cycle and are 128 floats apart) and makes the staging writes conflict-free.
    const int aoff0 = ph * 4 * C::PLANE + li * C::CK + 4 * (lh ^ ((li >> 4) & 1));   // kd even; + a*256 + kd*128 + pw*PLANE
    const int aoff1 = aoff0 ^ 4;                                                        // kd odd
    // this lane's B fragment inside a packed [64 co][8 ci] slice; slices of this wave: kd*16 + ph*4 + pw.
    // Every vector instruction beside the fp32 matrix instructions costs matrix-pipe time (the two do not
    // overlap on a SIMD), so the loop addresses global memory through buffer descriptors: per-lane offsets
    // stay constant and everything that moves is scalar.
    const __amdgpu_buffer_rsrc_t wrs = buf_rsrc(wp, ((long)gridDim.y * (Cin / C::CK) + 1) * C::TAPS * C::SLICE * 4);
    const unsigned wvo = ((ph * 4) * C::SLICE + (wn * 32 + li) * 8 + 4 * lh) * 4;
    const int wso0 = (cot * (Cin / C::CK) + kpart * nchunk) * C::TAPS * C::SLICE * 4;
    constexpr int PD = C::PD;
    float4 bring[PD];

    // ---- raw halo image: the 8 channels of a chunk of every halo voxel, [voxel][8], zeros outside the
    // volume.  Item i = p*512 + tid is half (i & 1) of voxel i >> 1: every wave fetches three 16-byte
    // items per chunk with plain coalesced loads (pairs of lanes read 32 contiguous bytes) two chunks ahead
    // and stores them ten taps later; the overlapping 4x4 patches are then cut out of LDS.
    const long sample = (long)D * H * W * Cin;
    __amdgpu_buffer_rsrc_t xrs = buf_rsrc(x + n * sample + kc0, (sample - kc0) * 4);   // sample of the brick being fetched
    unsigned roff[3];
    // (`opaque`: a copy of tid the compiler cannot see through, so that what depends only on the lane is
    // recomputed where it is used -- once per brick -- instead of living in registers through the chunk loop,
    // where there are none to spare)
    auto opaque = [](int v) { asm volatile("" : "+v"(v)); return v; };
    auto halo_offsets = [&](int bz0, int by0, int bx0) {
        const int tid_ = opaque(tid);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int i = p * C::NTHR + tid_, v = i >> 1;
            const int hx = v % BWP, hy = (v / BWP) % BHP, hz = v / (BWP * BHP);
            const int gz = bz0 - 1 + hz, gy = by0 - 1 + hy, gx = bx0 - 1 + hx;
            const bool ok = v < NV && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W;
            roff[p] = ok ? (unsigned)((((gz * H + gy) * W + gx) * Cin + 4 * (i & 1)) * 4) : W2_OUTSIDE;
        }
    };
    halo_offsets(z0, y0, x0);
    float4 raw[3];
    auto load_raw = [&](int c0) {
#pragma unroll
        for (int p = 0; p < 3; ++p) raw[p] = buf_load16(xrs, roff[p], c0 * 4);
    };
    auto store_raw = [&](float* img) {
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<float4*>(img + (p * C::NTHR + tid) * 4) = raw[p];
    };

    // ---- staging item of this lane: tile-row r of the 6 x 16 tiles, channel pair QI*h + qi in round h.
    // (Two channels at a time, not four: the 16-voxel patch then takes 32 registers beside the 128
    // accumulators instead of 64.)
    constexpr int QI = 4 / ROUNDS, NSW = 96 * QI / 64;
    const bool s_on = (SIV_W2EXP & 1) ? false : wave < NSW;
    const int s_item = s_on ? wave * 64 + lane : 0;
    const int s_qi = s_item % QI, s_r = s_item / QI;
    const int s_hz = s_r >> 4, s_th = (s_r & 15) / tpr, s_tw = (s_r & 15) - s_th * tpr;
    // the empty tile slot of a 15-tile plane reads patch 0 of its plane; its accumulator rows are never stored
    const int rbase = ((s_hz * BHP + (s_th < trows ? 2 * s_th : 0)) * BWP + (s_th < trows ? 2 * s_tw : 0)) * C::CK + 2 * s_qi;
    f32x2 patch[16];
    auto read_patch = [&](const float* img, int h) {
        if (SIV_W2EXP & 4) return;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                patch[rr * 4 + c] = *reinterpret_cast<const f32x2*>(img + rbase + 2 * QI * h + (rr * BWP + c) * C::CK);
    };
    // The transform is cut into eight pieces that the multiply loop issues between its MFMA groups, so
    // that the matrix pipe never waits for a long block of vector instructions: pieces 0..3 transform patch
    // row 0..3 along W in place, pieces 4..7 finish column pw = 0..3 along H and write its four positions.
    auto stage_piece = [&](int piece, int h, float* buf) {
        if (piece < 4) {
            const int rr = piece;
            const f32x2 t0 = patch[rr * 4 + 0] - patch[rr * 4 + 2];
            const f32x2 t1 = patch[rr * 4 + 1] + patch[rr * 4 + 2];
            const f32x2 t2 = patch[rr * 4 + 2] - patch[rr * 4 + 1];
            const f32x2 t3 = patch[rr * 4 + 1] - patch[rr * 4 + 3];
            patch[rr * 4 + 0] = t0; patch[rr * 4 + 1] = t1; patch[rr * 4 + 2] = t2; patch[rr * 4 + 3] = t3;
        } else {
            const int pw = piece - 4;
            if (SIV_W2EXP & 2) { patch[pw] = (patch[0 * 4 + pw] - patch[2 * 4 + pw]) + (patch[1 * 4 + pw] + patch[2 * 4 + pw]) + (patch[2 * 4 + pw] - patch[1 * 4 + pw]) + (patch[1 * 4 + pw] - patch[3 * 4 + pw]); return; }
            float* dst = buf + s_r * C::CK + ((2 * (QI * h + s_qi)) ^ (4 * (s_hz & 1))) + pw * C::PLANE;
            *reinterpret_cast<f32x2*>(dst + 0 * 4 * C::PLANE) = patch[0 * 4 + pw] - patch[2 * 4 + pw];
            *reinterpret_cast<f32x2*>(dst + 1 * 4 * C::PLANE) = patch[1 * 4 + pw] + patch[2 * 4 + pw];
            *reinterpret_cast<f32x2*>(dst + 2 * 4 * C::PLANE) = patch[2 * 4 + pw] - patch[1 * 4 + pw];
            *reinterpret_cast<f32x2*>(dst + 3 * 4 * C::PLANE) = patch[1 * 4 + pw] - patch[3 * 4 + pw];
        }
    };

    // ---- prologue (first brick of the block only): raw image 0 -> V buffer 0, raw image 1, the first weight
    // fragments (both images are requested before either is waited for)
    load_raw(0);
    f32x16 acc[4][2];          // [pw][row tile]
    auto clear_acc = [&]() {
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[p][a][r] = 0.f;
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int a = 0; a < 2; ++a) asm volatile("" : "+v"(acc[p][a]));
    };
    {
        float4 raw1[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) raw1[p] = buf_load16(xrs, roff[p], (nchunk > 1 ? 1 : 0) * C::CK * 4);
#pragma unroll
        for (int i = 0; i < PD; ++i) bring[i] = buf_load16(wrs, wvo, wso0 + ((i >> 2) * 16 + (i & 3)) * C::SLICE * 4);
        static_assert(12 % PD == 0, "ring index must not depend on the chunk");
        // the accumulators are cleared here, under the latency of the first loads: 128 vector instructions
        // per wave that would otherwise sit between the prologue's last barrier and the first matrix instruction
        __builtin_amdgcn_sched_barrier(0);
        clear_acc();
        __builtin_amdgcn_sched_barrier(0);
        store_raw(rimg);
#pragma unroll
        for (int p = 0; p < 3; ++p) raw[p] = raw1[p];
    }
    store_raw(rimg + C::RAW_FLOATS);
    SIV_STAMP(1);
    __syncthreads();
    if (s_on) {
#pragma unroll
        for (int h = 0; h < ROUNDS; ++h) {
            read_patch(rimg, h);
#pragma unroll
            for (int piece = 0; piece < 8; ++piece) stage_piece(piece, h, lds);
        }
    }
    __syncthreads();

    for (;;) {
        // the brick after this one (chained launches only: gridDim.x < nbricks)
        const int vnext = vb + (int)gridDim.x;
        const bool has_next = vnext < nbricks;
        int nn = n, nz0 = z0, ny0 = y0, nx0 = x0;
        if (has_next) brick_of(vnext, nn, nz0, ny0, nx0);
        SIV_STAMP(2);

        for (int cc = 0; cc < nchunk; ++cc) {
            const float* cur = lds + (cc & 1) * C::V_FLOATS;
            float* nxt = lds + ((cc + 1) & 1) * C::V_FLOATS;
            const float* rcur = rimg + ((cc + 1) & 1) * C::RAW_FLOATS;     // raw image of the next chunk of the stream
            float* rnxt = rimg + (cc & 1) * C::RAW_FLOATS;                 // receives the chunk after next
            const bool more = cc + 1 < nchunk || has_next;
            const int wso = wso0 + cc * C::TAPS * C::SLICE * 4;
            // the ring runs into the next chunk's slices: chunk 0 again behind the last chunk of a brick
            const int wsn = cc + 1 < nchunk ? wso + C::TAPS * C::SLICE * 4 : has_next ? wso0 : wso + C::TAPS * C::SLICE * 4;
            float4 anext[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) anext[a] = *reinterpret_cast<const float4*>(cur + aoff0 + a * 256);
#pragma unroll
            for (int t = 0; t < 12; ++t) {                     // t = kd*4 + pw
                const int pw = t & 3;
                float4 av[2];
                const float4 bv = bring[t % PD];
#pragma unroll
                for (int a = 0; a < 2; ++a) av[a] = anext[a];
                {   // fragment of tap t + PD
                    const int tn = t + PD;
                    bring[t % PD] = tn < 12 ? buf_load16(wrs, wvo, wso + ((tn >> 2) * 16 + (tn & 3)) * C::SLICE * 4)
                                            : buf_load16(wrs, wvo, wsn + (((tn - 12) >> 2) * 16 + ((tn - 12) & 3)) * C::SLICE * 4);
                }
                if (t + 1 < 12) {
                    const int t1 = t + 1;
                    const int tapoff = (t1 & 3) * C::PLANE + (t1 >> 2) * 128;
                    const int abase = (t1 >> 2) == 1 ? aoff1 : aoff0;
#pragma unroll
                    for (int a = 0; a < 2; ++a) anext[a] = *reinterpret_cast<const float4*>(cur + abase + a * 256 + tapoff);
                }
                __builtin_amdgcn_sched_barrier(0);
#define SIV_W2STEP(F)                                                                       \
    _Pragma("unroll") for (int a = 0; a < 2; ++a)                                           \
        acc[pw][a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a].F, bv.F, acc[pw][a], 0, 0, 0);
                SIV_W2STEP(x)
                SIV_W2STEP(y)
                SIV_W2STEP(z)
                SIV_W2STEP(w)
#undef SIV_W2STEP
                __builtin_amdgcn_sched_barrier(0);
                // the raw image two chunks down the stream: fetched now, stored at tap 10 (every wave).  From
                // the second-last chunk of a brick on it belongs to the next brick (offsets switched at the
                // end of the chunk before).
                if (t == 0) {
                    if (cc + 2 < nchunk) load_raw((cc + 2) * C::CK);
                    else if (has_next) load_raw((cc + 2 - nchunk) * C::CK);
                }
                if (t == 10) store_raw(rnxt);
                if (t == 11 && has_next && cc + 3 == nchunk) {
                    halo_offsets(nz0, ny0, nx0);
                    xrs = buf_rsrc(x + nn * sample + kc0, (sample - kc0) * 4);
                }
                // next chunk of the stream: patch reads one tap ahead of the first piece, two pieces per tap
                if (more && s_on) {
                    if (ROUNDS == 2) {
                        if (t == 0) read_patch(rcur, 0);
                        if (t >= 1 && t <= 4) { stage_piece(2 * t - 2, 0, nxt); stage_piece(2 * t - 1, 0, nxt); }
                        if (t == 5) read_patch(rcur, 1);
                        if (t >= 6 && t <= 9) { stage_piece(2 * t - 12, 1, nxt); stage_piece(2 * t - 11, 1, nxt); }
                    } else {
                        constexpr int T0 = SIV_W2STAGE_T0;
                        if (t == T0) read_patch(rcur, 0);
                        if (t == T0 + 1) { stage_piece(0, 0, nxt); stage_piece(1, 0, nxt); }
                        if (t == T0 + 2) { stage_piece(2, 0, nxt); stage_piece(3, 0, nxt); }
                        if (t >= T0 + 3 && t <= T0 + 6) stage_piece(t - T0 + 1, 0, nxt);
                    }
                }
            }
            SIV_STAMP(3 + 2 * cc);
            __syncthreads();
            SIV_STAMP(4 + 2 * cc);
        }

        // ---- output transform and store, one row tile (a) at a time ----
        // along W on the accumulators: Z[ow'] = sum_pw At[ow'][pw] M[ph][pw]; along H across the four ph waves
        // through the exchange tiles E[(ph*2 + wn)*2 + ow'] of 32 rows x 32 channels: oh = 0: Z0 + Z1 + Z2,
        // oh = 1: Z1 - Z2 - Z3
        float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = make_float4(0.f, 0.f, 0.f, 0.f);
        const int tid_ = opaque(tid);                          // lane-only quantities: recomputed per brick
        const int lane = tid_ & 63, wave = tid_ >> 6;
        const int li = lane & 31, lh = lane >> 5;
        const int ph = wave & 3, wn = wave >> 2;
        const int cq = lane & 7;
        const int co = co0 + wn * 32 + 4 * cq;
        float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias && co < Cout) bq = *reinterpret_cast<const float4*>(bias + co);
        const int oh = ph & 1, ow = ph >> 1;                   // the output parity this wave combines and stores
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                float* tile = epi + ((ph * 2 + wn) * 2 + o) * C::EPI_TILE;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = o == 0 ? (acc[0][a][r] + acc[1][a][r]) + acc[2][a][r]
                                           : (acc[1][a][r] - acc[2][a][r]) - acc[3][a][r];
                    tile[((r & 3) + 8 * (r >> 2) + 4 * lh) * C::EPI_STRIDE + li] = v;
                }
            }
            if (a == 1 && has_next) clear_acc();           // for the next brick, while the others catch up
            __syncthreads();
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int row = ps * 8 + (lane >> 3);
                const int off = row * C::EPI_STRIDE + 4 * cq;
                const float4 za = *reinterpret_cast<const float4*>(epi + (((oh + 0) * 2 + wn) * 2 + ow) * C::EPI_TILE + off);
                const float4 zb = *reinterpret_cast<const float4*>(epi + (((oh + 1) * 2 + wn) * 2 + ow) * C::EPI_TILE + off);
                const float4 zc = *reinterpret_cast<const float4*>(epi + (((oh + 2) * 2 + wn) * 2 + ow) * C::EPI_TILE + off);
                float4 v = oh == 0 ? w2add(w2add(za, zb), zc) : w2sub(w2sub(za, zb), zc);
                const int m = a * 32 + row;
                const int et = (m & 15) / tpr;
                const int gz = z0 + (m >> 4), gy = y0 + 2 * et + oh, gx = x0 + 2 * ((m & 15) - et * tpr) + ow;
                if (et < trows && gz < D && gy < H && gx < W && co < Cout) {
                    s1 = w2add(s1, v);
                    s2.x = fmaf(v.x, v.x, s2.x); s2.y = fmaf(v.y, v.y, s2.y);
                    s2.z = fmaf(v.z, v.z, s2.z); s2.w = fmaf(v.w, v.w, s2.w);
                    v = w2add(v, bq);
                    *reinterpret_cast<float4*>(y + ((((long)n * D + gz) * H + gy) * W + gx) * Cout + co) = v;
                }
            }
            __syncthreads();                   // the tiles have been read
        }
        SIV_STAMP(29);
        if (stat_part) {
            // lanes with equal cq (bits 3..5 differ) hold the same four channels
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) {
                s1.x += __shfl_xor(s1.x, o, 64); s1.y += __shfl_xor(s1.y, o, 64);
                s1.z += __shfl_xor(s1.z, o, 64); s1.w += __shfl_xor(s1.w, o, 64);
                s2.x += __shfl_xor(s2.x, o, 64); s2.y += __shfl_xor(s2.y, o, 64);
                s2.z += __shfl_xor(s2.z, o, 64); s2.w += __shfl_xor(s2.w, o, 64);
            }
            float* red = epi;                  // [8 waves][2 (sum, sumsq)][32 channels]
            if (lane < 8) {
                *reinterpret_cast<float4*>(red + (wave * 2 + 0) * 32 + 4 * cq) = s1;
                *reinterpret_cast<float4*>(red + (wave * 2 + 1) * 32 + 4 * cq) = s2;
            }
            __syncthreads();
            if (tid < 128) {
                const int k = tid >> 6, c = tid & 63, w0 = (c >> 5) * 4;
                float s = 0.f;
#pragma unroll
                for (int p = 0; p < 4; ++p) s += red[((w0 + p) * 2 + k) * 32 + (c & 31)];
                if (co0 + c < Cout) stat_part[((long)vb * 2 + k) * Cout + co0 + c] = s;
            }
            __syncthreads();                   // before the next brick's staging writes here
        }
        SIV_STAMP(30);
        if (!has_next) break;
        vb = vnext; n = nn; z0 = nz0; y0 = ny0; x0 = nx0;
    }
}

// Packed, transformed weights: wp[ct][cc][kd*16 + ph*4 + pw][co (64)][ci (8)] = (G g_kd G^T)[ph][pw].
//   flip == 0: g[kh][kw] = w[kco][kd][kh][kw][kci]              (w is [Kout][27][Kin])
//   flip == 1: g[kh][kw] = w[kci][2-kd][2-kh][2-kw][kco]        (w is [Kin][27][Kout]; input gradient)
__global__ __launch_bounds__(256) void k3_pack_weights_wino2d(const float* __restrict__ w,
                                                              float* __restrict__ wp, int Kin, int Kout,
                                                              int flip) {
    const int nchunk = Kin / 8, ntile = Kout / 64;
    const long total = (long)ntile * nchunk * 48 * 64 * 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7);
        long r = i >> 3;
        const int co = (int)(r & 63); r >>= 6;
        const int t = (int)(r % 48); r /= 48;
        const int cc = (int)(r % nchunk);
        const int ct = (int)(r / nchunk);
        const int pw = t & 3, ph = (t >> 2) & 3, kd = t >> 4;
        const int kco = ct * 64 + co, kci = cc * 8 + j;
        float g[3][3];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int tap = (kd * 3 + kh) * 3 + kw;
                g[kh][kw] = flip ? w[((long)kci * 27 + (26 - tap)) * Kout + kco] : w[((long)kco * 27 + tap) * Kin + kci];
            }
        // rows of G applied along kw, then along kh
        float u[3];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
            u[kh] = pw == 0 ? g[kh][0] : pw == 1 ? 0.5f * ((g[kh][0] + g[kh][1]) + g[kh][2])
                  : pw == 2 ? 0.5f * ((g[kh][0] - g[kh][1]) + g[kh][2]) : g[kh][2];
        wp[i] = ph == 0 ? u[0] : ph == 1 ? 0.5f * ((u[0] + u[1]) + u[2]) : ph == 2 ? 0.5f * ((u[0] - u[1]) + u[2]) : u[2];
    }
}

// ---------------------------------------------------------------------------------------------
// The same kernel with F(4, 3) along W and F(2, 3) along H: a 2 x 4 output tile from a 4 x 6 patch,
//   Y = A_h^T [ sum_kd (G_h g_kd G_w^T) .* (B_h^T d_kd B_w) ] A_w
//   B_w^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   G_w   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//   A_w^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
// 24 positions x 3 planes = 72 multiply-adds per 8 outputs and channel pair: 9 per output instead of 12.
//   M rows  = 2x4 output tiles of a 4x8x8-voxel brick: 8 per plane, 32 in all = ONE row tile
//   a block = 8 waves: wave (ph, wn) owns the six positions (ph, 0..5) for output channel half wn
//             (6 accumulators = 96 registers; 72 instead of 96 matrix instructions per chunk)
//   V[pos][plane][tile][ci] of a chunk = 36 KB; staging items = (plane, tile, channel pair) = 192 = waves 0..2
// Brick planes of 8 x 8 voxels only, no split of the channel contraction; everything else (raw halo image,
// one chunk stream per block, epilogue between the live images, BatchNorm sums) as above.
// ---------------------------------------------------------------------------------------------
struct W4Cfg {
    static constexpr int BD = 4, CK = 8;
    static constexpr int TPP = 8, HD = BD + 2;
    static constexpr int PLANE = HD * TPP * CK;            // floats of one position: 384
    static constexpr int NPOS = 24;
    static constexpr int V_FLOATS = NPOS * PLANE;          // 9216 floats = 36 KB
    static constexpr int NTHR = 512, NT = 64;
    static constexpr int TAPS = 72, SLICE = NT * 8;        // (kd, ph, pw) slices of [64 co][8 ci]
#ifndef SIV_W4PD
#define SIV_W4PD 6
#endif
    static constexpr int PD = SIV_W4PD;                    // weight fragments fetched PD taps ahead (divides 18)
    static constexpr int EPI_STRIDE = 36, EPI_TILE = 32 * EPI_STRIDE;
    static constexpr int RAW_ITEMS = 3 * NTHR, RAW_FLOATS = RAW_ITEMS * 4;      // 24 KB
    // V0 | V1 | raw image 0 | padding | raw image 1: between two bricks V0 and raw image 1 are live and the
    // epilogue has the 16 exchange tiles in between
    static constexpr int PAD_FLOATS = 16 * EPI_TILE - V_FLOATS - RAW_FLOATS;     // 3072
    static constexpr int RAW0 = 2 * V_FLOATS, RAW1 = RAW0 + RAW_FLOATS + PAD_FLOATS;
    static constexpr int LDS_FLOATS = RAW1 + RAW_FLOATS;                         // 33792 floats = 132 KB
    static constexpr size_t LDS_BYTES = (size_t)LDS_FLOATS * sizeof(float);
    static_assert(PAD_FLOATS >= 0, "exchange tiles fit between the live images");
};

#ifndef SIV_W4STAGE_T0
#define SIV_W4STAGE_T0 8
#endif

__global__ __launch_bounds__(512, 2) void k3_fwd_wino2x4(const float* __restrict__ x,
                                                         const float* __restrict__ wp,
                                                         const float* __restrict__ bias,
                                                         float* __restrict__ y, int N, int D, int H, int W,
                                                         int Cin, int Cout, int nbd, int nbh, int nbw,
                                                         int nbricks, float* __restrict__ stat_part) {
    using C = W4Cfg;
    constexpr int BHP = 10, BWP = 10, NV = C::HD * BHP * BWP;      // halo plane 10 x 10, 600 voxels
    extern __shared__ __attribute__((aligned(16))) float lds[];
    auto rawp = [&](int j) { return lds + (j ? C::RAW1 : C::RAW0); };
    float* const epi = lds + C::V_FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int ph = wave & 3, wn = wave >> 2;

    auto item_of = [&](int b) { return (nbricks & 7) == 0 ? (b & 7) * (nbricks >> 3) + (b >> 3) : b; };
    auto brick_of = [&](int b, int& bn, int& bz0, int& by0, int& bx0) {
        b = item_of(b);
        const int bx = b % nbw; b /= nbw;
        const int by = b % nbh; b /= nbh;
        bz0 = (b % nbd) * C::BD; by0 = by * 8; bx0 = bx * 8;
        bn = b / nbd;
    };
    int vb = blockIdx.x;
    const int cot = blockIdx.y;
    int n, z0, y0, x0;
    brick_of(vb, n, z0, y0, x0);
    const int co0 = cot * C::NT;
    const int nchunk = Cin / C::CK;

    // A row m = li is tile (dz = m >> 3, th, tw) and reads V row (dz + kd) * 8 + tile = m + 8 kd
    const int aoff = (ph * 6) * C::PLANE + li * C::CK + 4 * lh;          // + kd * 64 + pw * PLANE
    const __amdgpu_buffer_rsrc_t wrs = buf_rsrc(wp, ((long)gridDim.y * nchunk + 1) * C::TAPS * C::SLICE * 4);
    const unsigned wvo = ((ph * 6) * C::SLICE + (wn * 32 + li) * 8 + 4 * lh) * 4;
    const int wso0 = cot * nchunk * C::TAPS * C::SLICE * 4;
    constexpr int PD = C::PD;
    float4 bring[PD];
    auto tap_slice = [](int t) { return (t / 6) * 24 + t % 6; };        // slice of tap t = kd * 6 + pw (+ ph * 6 in wvo)

    const long sample = (long)D * H * W * Cin;
    __amdgpu_buffer_rsrc_t xrs = buf_rsrc(x + n * sample, sample * 4);
    unsigned roff[3];
    auto opaque = [](int v) { asm volatile("" : "+v"(v)); return v; };
    auto halo_offsets = [&](int bz0, int by0, int bx0) {
        const int tid_ = opaque(tid);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int i = p * C::NTHR + tid_, v = i >> 1;
            const int hx = v % BWP, hy = (v / BWP) % BHP, hz = v / (BWP * BHP);
            const int gz = bz0 - 1 + hz, gy = by0 - 1 + hy, gx = bx0 - 1 + hx;
            const bool ok = v < NV && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W;
            roff[p] = ok ? (unsigned)((((gz * H + gy) * W + gx) * Cin + 4 * (i & 1)) * 4) : W2_OUTSIDE;
        }
    };
    halo_offsets(z0, y0, x0);
    float4 raw[3];
    auto load_raw = [&](int c0) {
#pragma unroll
        for (int p = 0; p < 3; ++p) raw[p] = buf_load16(xrs, roff[p], c0 * 4);
    };
    auto store_raw = [&](float* img) {
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<float4*>(img + (p * C::NTHR + tid) * 4) = raw[p];
    };

    // ---- staging item of this lane: (halo plane hz, tile (th, tw), channel pair qi); 192 items = waves 0..2
    const bool s_on = wave < 3;
    const int s_item = s_on ? wave * 64 + lane : 0;
    const int s_qi = s_item & 3, s_r = s_item >> 2;                      // s_r = hz * 8 + th * 2 + tw
    const int s_hz = s_r >> 3, s_th = (s_r & 7) >> 1, s_tw = s_r & 1;
    const int rbase = ((s_hz * BHP + 2 * s_th) * BWP + 4 * s_tw) * C::CK + 2 * s_qi;
    f32x2 patch[24];                                                      // [row 0..3][column 0..5]
    auto read_patch = [&](const float* img) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int c = 0; c < 6; ++c)
                patch[rr * 6 + c] = *reinterpret_cast<const f32x2*>(img + rbase + (rr * BWP + c) * C::CK);
    };
    // pieces 0..3: patch row rr transformed along W in place (B_w^T); pieces 4..9: column pw finished along H
    // (the rows of B_h^T as in the F(2x2) kernel) and its four positions written
    auto stage_piece = [&](int piece, float* buf) {
        const f32x2 k2 = {2.f, 2.f}, k4 = {4.f, 4.f}, m4 = {-4.f, -4.f}, m5 = {-5.f, -5.f}, m2 = {-2.f, -2.f};
        if (piece < 4) {
            f32x2* d = patch + piece * 6;
            const f32x2 a = pk_fma(m4, d[2], d[4]);            // d4 - 4 d2
            const f32x2 b = pk_fma(m4, d[1], d[3]);            // d3 - 4 d1
            const f32x2 c = d[4] - d[2];
            const f32x2 e = d[3] - d[1];
            const f32x2 t0 = pk_fma(k4, d[0], pk_fma(m5, d[2], d[4]));
            const f32x2 t5 = pk_fma(k4, d[1], pk_fma(m5, d[3], d[5]));
            d[0] = t0;
            d[1] = a + b;
            d[2] = a - b;
            d[3] = pk_fma(k2, e, c);
            d[4] = pk_fma(m2, e, c);
            d[5] = t5;
        } else {
            const int pw = piece - 4;
            float* dst = buf + s_r * C::CK + 2 * s_qi + pw * C::PLANE;
            *reinterpret_cast<f32x2*>(dst + 0 * 6 * C::PLANE) = patch[0 * 6 + pw] - patch[2 * 6 + pw];
            *reinterpret_cast<f32x2*>(dst + 1 * 6 * C::PLANE) = patch[1 * 6 + pw] + patch[2 * 6 + pw];
            *reinterpret_cast<f32x2*>(dst + 2 * 6 * C::PLANE) = patch[2 * 6 + pw] - patch[1 * 6 + pw];
            *reinterpret_cast<f32x2*>(dst + 3 * 6 * C::PLANE) = patch[1 * 6 + pw] - patch[3 * 6 + pw];
        }
    };

    // ---- prologue (first brick of the block only)
    load_raw(0);
    f32x16 acc[6];             // [pw]
    auto clear_acc = [&]() {
#pragma unroll
        for (int p = 0; p < 6; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
#pragma unroll
        for (int p = 0; p < 6; ++p) asm volatile("" : "+v"(acc[p]));
    };
    {
        float4 raw1[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) raw1[p] = buf_load16(xrs, roff[p], (nchunk > 1 ? 1 : 0) * C::CK * 4);
#pragma unroll
        for (int i = 0; i < PD; ++i) bring[i] = buf_load16(wrs, wvo, wso0 + tap_slice(i) * C::SLICE * 4);
        static_assert(18 % PD == 0, "ring index must not depend on the chunk");
        __builtin_amdgcn_sched_barrier(0);
        clear_acc();
        __builtin_amdgcn_sched_barrier(0);
        store_raw(rawp(0));
#pragma unroll
        for (int p = 0; p < 3; ++p) raw[p] = raw1[p];
    }
    store_raw(rawp(1));
    __syncthreads();
    if (s_on) {
        read_patch(rawp(0));
#pragma unroll
        for (int piece = 0; piece < 10; ++piece) stage_piece(piece, lds);
    }
    __syncthreads();

    for (;;) {
        const int vnext = vb + (int)gridDim.x;
        const bool has_next = vnext < nbricks;
        int nn = n, nz0 = z0, ny0 = y0, nx0 = x0;
        if (has_next) brick_of(vnext, nn, nz0, ny0, nx0);

        for (int cc = 0; cc < nchunk; ++cc) {
            const float* cur = lds + (cc & 1) * C::V_FLOATS;
            float* nxt = lds + ((cc + 1) & 1) * C::V_FLOATS;
            const float* rcur = rawp((cc + 1) & 1);                 // raw image of the next chunk of the stream
            float* rnxt = rawp(cc & 1);                             // receives the chunk after next
            const bool more = cc + 1 < nchunk || has_next;
            const int wso = wso0 + cc * C::TAPS * C::SLICE * 4;
            const int wsn = cc + 1 < nchunk ? wso + C::TAPS * C::SLICE * 4 : has_next ? wso0 : wso + C::TAPS * C::SLICE * 4;
            float4 anext = *reinterpret_cast<const float4*>(cur + aoff);
#pragma unroll
            for (int t = 0; t < 18; ++t) {                     // t = kd * 6 + pw
                const int pw = t % 6;
                const float4 av = anext;
                const float4 bv = bring[t % PD];
                {   // fragment of tap t + PD
                    const int tn = t + PD;
                    bring[t % PD] = tn < 18 ? buf_load16(wrs, wvo, wso + tap_slice(tn) * C::SLICE * 4)
                                            : buf_load16(wrs, wvo, wsn + tap_slice(tn - 18) * C::SLICE * 4);
                }
                if (t + 1 < 18) {
                    const int t1 = t + 1;
                    anext = *reinterpret_cast<const float4*>(cur + aoff + (t1 % 6) * C::PLANE + (t1 / 6) * 64);
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[pw] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[pw], 0, 0, 0);
                acc[pw] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[pw], 0, 0, 0);
                acc[pw] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[pw], 0, 0, 0);
                acc[pw] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[pw], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (t == 0) {
                    if (cc + 2 < nchunk) load_raw((cc + 2) * C::CK);
                    else if (has_next) load_raw((cc + 2 - nchunk) * C::CK);
                }
                if (t == 16) store_raw(rnxt);
                if (t == 17 && has_next && cc + 3 == nchunk) {
                    halo_offsets(nz0, ny0, nx0);
                    xrs = buf_rsrc(x + nn * sample, sample * 4);
                }
                if (more && s_on) {
                    constexpr int T0 = SIV_W4STAGE_T0;
                    if (t == T0) read_patch(rcur);
                    if (t == T0 + 1) { stage_piece(0, nxt); stage_piece(1, nxt); }
                    if (t == T0 + 2) { stage_piece(2, nxt); stage_piece(3, nxt); }
                    if (t >= T0 + 3 && t <= T0 + 8) stage_piece(t - T0 + 1, nxt);
                }
            }
            __syncthreads();
        }

        // ---- output transform and store: columns ow = 2 p + o of the tile in pass p
        float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = make_float4(0.f, 0.f, 0.f, 0.f);
        const int tid_ = opaque(tid);
        const int lane = tid_ & 63, wave = tid_ >> 6;
        const int li = lane & 31, lh = lane >> 5;
        const int ph = wave & 3, wn = wave >> 2;
        const int cq = lane & 7;
        const int co = co0 + wn * 32 + 4 * cq;
        float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias && co < Cout) bq = *reinterpret_cast<const float4*>(bias + co);
        const int oh = ph & 1, owl = ph >> 1;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                float* tile = epi + ((ph * 2 + wn) * 2 + o) * C::EPI_TILE;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float dm = acc[1][r] - acc[2][r], sm = acc[1][r] + acc[2][r];
                    const float dn = acc[3][r] - acc[4][r], sn = acc[3][r] + acc[4][r];
                    float v;
                    if (p == 0 && o == 0) v = (acc[0][r] + sm) + sn;
                    else if (p == 0) v = fmaf(2.f, dn, dm);
                    else if (o == 0) v = fmaf(4.f, sn, sm);
                    else v = fmaf(8.f, dn, dm) + acc[5][r];
                    tile[((r & 3) + 8 * (r >> 2) + 4 * lh) * C::EPI_STRIDE + li] = v;
                }
            }
            if (p == 1 && has_next) clear_acc();
            __syncthreads();
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int row = ps * 8 + (lane >> 3);
                const int off = row * C::EPI_STRIDE + 4 * cq;
                const float4 za = *reinterpret_cast<const float4*>(epi + (((oh + 0) * 2 + wn) * 2 + owl) * C::EPI_TILE + off);
                const float4 zb = *reinterpret_cast<const float4*>(epi + (((oh + 1) * 2 + wn) * 2 + owl) * C::EPI_TILE + off);
                const float4 zc = *reinterpret_cast<const float4*>(epi + (((oh + 2) * 2 + wn) * 2 + owl) * C::EPI_TILE + off);
                float4 v = oh == 0 ? w2add(w2add(za, zb), zc) : w2sub(w2sub(za, zb), zc);
                const int m = row;
                const int gz = z0 + (m >> 3), gy = y0 + 2 * ((m & 7) >> 1) + oh, gx = x0 + 4 * (m & 1) + 2 * p + owl;
                if (gz < D && gy < H && gx < W && co < Cout) {
                    s1 = w2add(s1, v);
                    s2.x = fmaf(v.x, v.x, s2.x); s2.y = fmaf(v.y, v.y, s2.y);
                    s2.z = fmaf(v.z, v.z, s2.z); s2.w = fmaf(v.w, v.w, s2.w);
                    v = w2add(v, bq);
                    *reinterpret_cast<float4*>(y + ((((long)n * D + gz) * H + gy) * W + gx) * Cout + co) = v;
                }
            }
            __syncthreads();
        }
        if (stat_part) {
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) {
                s1.x += __shfl_xor(s1.x, o, 64); s1.y += __shfl_xor(s1.y, o, 64);
                s1.z += __shfl_xor(s1.z, o, 64); s1.w += __shfl_xor(s1.w, o, 64);
                s2.x += __shfl_xor(s2.x, o, 64); s2.y += __shfl_xor(s2.y, o, 64);
                s2.z += __shfl_xor(s2.z, o, 64); s2.w += __shfl_xor(s2.w, o, 64);
            }
            float* red = epi;                  // [8 waves][2 (sum, sumsq)][32 channels]
            if (lane < 8) {
                *reinterpret_cast<float4*>(red + (wave * 2 + 0) * 32 + 4 * cq) = s1;
                *reinterpret_cast<float4*>(red + (wave * 2 + 1) * 32 + 4 * cq) = s2;
            }
            __syncthreads();
            if (tid < 128) {
                const int k = tid >> 6, c = tid & 63, w0 = (c >> 5) * 4;
                float s = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) s += red[((w0 + q) * 2 + k) * 32 + (c & 31)];
                if (co0 + c < Cout) stat_part[((long)vb * 2 + k) * Cout + co0 + c] = s;
            }
            __syncthreads();
        }
        if (!has_next) break;
        vb = vnext; n = nn; z0 = nz0; y0 = ny0; x0 = nx0;
    }
}

// wp[ct][cc][kd*24 + ph*6 + pw][co (64)][ci (8)] = (G_h g_kd G_w^T)[ph][pw]; g as in k3_pack_weights_wino2d
__global__ __launch_bounds__(256) void k3_pack_weights_wino2x4(const float* __restrict__ w,
                                                               float* __restrict__ wp, int Kin, int Kout,
                                                               int flip) {
    const int nchunk = Kin / 8, ntile = Kout / 64;
    const long total = (long)ntile * nchunk * 72 * 64 * 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7);
        long r = i >> 3;
        const int co = (int)(r & 63); r >>= 6;
        const int t = (int)(r % 72); r /= 72;
        const int cc = (int)(r % nchunk);
        const int ct = (int)(r / nchunk);
        const int pw = t % 6, ph = (t / 6) % 4, kd = t / 24;
        const int kco = ct * 64 + co, kci = cc * 8 + j;
        float g[3][3];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int tap = (kd * 3 + kh) * 3 + kw;
                g[kh][kw] = flip ? w[((long)kci * 27 + (26 - tap)) * Kout + kco] : w[((long)kco * 27 + tap) * Kin + kci];
            }
        // rows of G_w along kw (in double: 1/6 and 1/24 are not exact), then rows of G_h along kh
        double u[3];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const double g0 = g[kh][0], g1 = g[kh][1], g2 = g[kh][2];
            u[kh] = pw == 0 ? g0 / 4.0 : pw == 1 ? -(g0 + g1 + g2) / 6.0 : pw == 2 ? (-g0 + g1 - g2) / 6.0
                  : pw == 3 ? g0 / 24.0 + g1 / 12.0 + g2 / 6.0 : pw == 4 ? g0 / 24.0 - g1 / 12.0 + g2 / 6.0 : g2;
        }
        const double v = ph == 0 ? u[0] : ph == 1 ? 0.5 * (u[0] + u[1] + u[2]) : ph == 2 ? 0.5 * (u[0] - u[1] + u[2]) : u[2];
        wp[i] = (float)v;
    }
}

// SIV_W2CHAIN=0 (build flag) launches one block per brick as before
#ifndef SIV_W2CHAIN
#define SIV_W2CHAIN 1
#endif
constexpr bool W2_CHAIN = SIV_W2CHAIN != 0;
constexpr long W2_CUS = 256;           // CUs of the device (gfx950)
bool k3_fwd_wino2d_supported(int Cin, int Cout) { return Cin % 8 == 0 && Cout % 64 == 0; }
size_t k3_fwd_wino2d_ws_bytes(int Cin, int Cout) {     // one chunk of padding behind the last slice (72: the F(2x4) form)
    return ((size_t)(Cout / 64) * (Cin / 8) + 1) * 72 * 512 * sizeof(float);
}
// Which convolutions take the F(2,3) x F(4,3) form on volumes with 8 x 8 brick planes and an unsplit contraction
// (environment SIV_W2_W4, read at every launch): unset or "2" = the input gradients, "1" = forward too, "0" =
// none.  The input gradients are the default because their rounding adds to a gradient once, linearly, while
// forward rounding moves activations across LeakyReLU kinks and BatchNorm statistics: at batch 1 the end-to-end
// gradients of the benchmark model move by up to 3.5 % of a tensor's maximum with the F(2x4) forward (0.8 % with
// F(2x2), 1.5 % with the direct kernels, 0.6 % for the fp32 reference itself), and not at all with the F(2x4)
// input gradient.
static bool w2d_use_w4(int flip) {
    const char* e = getenv("SIV_W2_W4");
    const char m = e && e[0] ? e[0] : '2';
    return m == '1' || (m == '2' && flip);
}

// Brick planes of 6 x 10 voxels where they pad the volume less than 8 x 8 ones (10- and 20-wide levels).
static int w2d_tpr(int H, int W) {
    return cdiv(H, 6) * 6 * cdiv(W, 10) * 10 * 16 < cdiv(H, 8) * 8 * cdiv(W, 8) * 8 * 15 ? 5 : 4;
}
long k3_fwd_wino2d_bricks(int N, int D, int H, int W) {
    const int tpr = w2d_tpr(H, W);
    return (long)N * cdiv(D, 4) * cdiv(H, 2 * (16 / tpr)) * cdiv(W, 2 * tpr);
}
// Same contract as k3_fwd_wino_launch with ups == 1 (stat_part rows: k3_fwd_wino2d_bricks()).
// ---- channel contraction split -----------------------------------------------------------------------
// A level with fewer (brick, output-channel tile) units than CUs runs one round of blocks that each walk all
// of Cin: 256 -> 256 @10x12x10, batch 8 = 192 blocks of 32 chunks on 256 CUs, 0.20 ms at 44 % of the peak.
// Cut into quarters of the input channels the same work is 768 items = three per CU as one chunk stream
// (24 chunks instead of 32 per block); each item leaves a partial output in a buffer of four, and
// w2d_ksum_k adds them in a fixed order, adds the bias and forms the BatchNorm sums of its rows.  Two parts
// where the units are a few rounds of blocks with a last round half empty (128 -> 128 @20x24x20, batch 8:
// 640 units = 3 rounds for the work of 2.5; as 1280 items exactly 5 per CU).
constexpr int W2_KSUM_BLOCKS = 256;
// SIV_W2KSPLIT (build flag): 0 = never split, 2 = four parts only
#ifndef SIV_W2KSPLIT
#define SIV_W2KSPLIT 1
#endif
// log2 of the number of channel parts (0: not split)
static int w2d_kshift(long nbricks, long M, int Cin, int Cout) {
    const int nchunk = Cin / 8, ncot = Cout / 64;
    if (!SIV_W2KSPLIT || !W2_CHAIN || !(ncot == 1 || ncot == 2 || ncot == 4 || ncot == 8)    // 256 blocks, whole XCDs per output tile
        || 256 % (Cout / 4) != 0)                                                             // w2d_ksum_k's thread layout
        return 0;
    const long g = W2_CUS / ncot, units = nbricks * ncot;
    // one under-filled round today: four parts (items of >= 4 chunks, an even number; chained)
    if (units * 5 < W2_CUS * 4 && nchunk % 8 == 0 && nchunk >= 16 && g < nbricks * 4) return 2;
    // a few rounds, the last one at most 60 % full, and twice the output is little to move: two parts
    const long rounds = cdiv(units, W2_CUS);
    if (SIV_W2KSPLIT != 2 && units > W2_CUS && rounds <= 4 && units * 10 <= (rounds * 10 - 4) * W2_CUS && cdiv(2 * units, W2_CUS) < 2 * rounds
        && nchunk % 4 == 0 && nchunk >= 8 && 2 * M * Cout * 4 <= (96L << 20))
        return 1;
    return 0;
}
// part: [KP][M][Cout] -> y = bias + (part[0] + part[1]) [+ (part[2] + part[3])]; stat_part (optional): row
// blockIdx.x = per-channel sum and sum of squares of the bias-free output over the block's voxels.
template <int KP>
__global__ __launch_bounds__(256) void w2d_ksum_k(const float* __restrict__ part, const float* __restrict__ bias,
                                                  float* __restrict__ y, long M, int Cout,
                                                  float* __restrict__ stat_part) {
    __shared__ float4 red[2][256];
    const int Q = Cout >> 2, vpp = 256 / Q;                    // channel quads per voxel, voxels per pass
    const int q = threadIdx.x % Q, sub = threadIdx.x / Q;
    const long per = (M + gridDim.x - 1) / gridDim.x;
    const long v0 = blockIdx.x * per, v1 = v0 + per < M ? v0 + per : M;
    const long slab = M * Cout;
    float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) bq = *reinterpret_cast<const float4*>(bias + 4 * q);
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long v = v0 + sub; v < v1; v += vpp) {
        const float* p = part + v * Cout + 4 * q;
        const float4 a0 = *reinterpret_cast<const float4*>(p), a1 = *reinterpret_cast<const float4*>(p + slab);
        float4 a = w2add(a0, a1);
        if constexpr (KP == 4) {
            const float4 a2 = *reinterpret_cast<const float4*>(p + 2 * slab), a3 = *reinterpret_cast<const float4*>(p + 3 * slab);
            a = w2add(a, w2add(a2, a3));
        }
        s1 = w2add(s1, a);
        s2.x = fmaf(a.x, a.x, s2.x); s2.y = fmaf(a.y, a.y, s2.y); s2.z = fmaf(a.z, a.z, s2.z); s2.w = fmaf(a.w, a.w, s2.w);
        *reinterpret_cast<float4*>(y + v * Cout + 4 * q) = w2add(a, bq);
    }
    if (!stat_part) return;
    red[0][threadIdx.x] = s1;
    red[1][threadIdx.x] = s2;
    __syncthreads();
    if (threadIdx.x < Q) {
        float4 t1 = red[0][q], t2 = red[1][q];
        for (int j = 1; j < vpp; ++j) { t1 = w2add(t1, red[0][j * Q + q]); t2 = w2add(t2, red[1][j * Q + q]); }
        *reinterpret_cast<float4*>(stat_part + ((long)blockIdx.x * 2 + 0) * Cout + 4 * q) = t1;
        *reinterpret_cast<float4*>(stat_part + ((long)blockIdx.x * 2 + 1) * Cout + 4 * q) = t2;
    }
}
// Rows of BatchNorm partial sums the launch writes, and the bytes it wants behind the packed weights and those
// rows for the four partial outputs (0: the contraction is not split for this shape).
long k3_fwd_wino2d_stat_rows(int N, int D, int H, int W, int Cin, int Cout) {
    const long nb = k3_fwd_wino2d_bricks(N, D, H, W);
    return w2d_kshift(nb, (long)N * D * H * W, Cin, Cout) ? W2_KSUM_BLOCKS : nb;
}
size_t k3_fwd_wino2d_ksplit_bytes(int N, int D, int H, int W, int Cin, int Cout) {
    const long M = (long)N * D * H * W;
    const int ks = w2d_kshift(k3_fwd_wino2d_bricks(N, D, H, W), M, Cin, Cout);
    return ks ? ((size_t)M * Cout * sizeof(float)) << ks : 0;
}

// kws / kws_bytes: room for the four partial outputs of the split contraction (NULL / too small: not split).
int k3_fwd_wino2d_launch(const float* x, const float* w, const float* bias, float* y, int N, int D, int H,
                         int W, int Cin, int Cout, int flip, void* ws, size_t ws_bytes, hipStream_t s,
                         float* stat_part, void* kws, size_t kws_bytes) {
    SIV_CHECK_WS(k3_fwd_wino2d_ws_bytes(Cin, Cout), ws_bytes, "conv3d_k3(winograd 2-D)");
    SIV_CHECK_ARG((long)D * H * W * Cin < (1L << 29), "conv3d_k3(winograd 2-D): sample too large for 32-bit byte offsets");
    if (w2d_use_w4(flip) && w2d_tpr(H, W) == 4 && (Cin / 8) % 2 == 0 && Cin >= 32 &&
        w2d_kshift(k3_fwd_wino2d_bricks(N, D, H, W), (long)N * D * H * W, Cin, Cout) == 0) {
        const long total4 = (long)(Cout / 64) * (Cin / 8) * 72 * 512;
        long g4 = cdiv(total4, 256);
        if (g4 > 4096) g4 = 4096;
        bool need;
        const float* wpk4 = (const float*)pack_slot(w, PF_WINO2D, 1, flip, Cin, Cout, k3_fwd_wino2d_ws_bytes(Cin, Cout), ws, &need);
        if (need) {
            k3_pack_weights_wino2x4<<<(unsigned)g4, 256, 0, s>>>(w, const_cast<float*>(wpk4), Cin, Cout, flip);
            SIV_CHECK_PACK_LAUNCH("conv3d_k3(pack F(2x4) winograd weights)", wpk4);
        }
        const int nbd = (D + 3) / 4, nbh = (int)cdiv(H, 8), nbw = (int)cdiv(W, 8);
        const long nbricks = (long)N * nbd * nbh * nbw;
        if (nbricks > 0x7fffffffL) {
            set_error("conv3d_k3(winograd 2-D): too many bricks");
            return SIV_EINVAL;
        }
        long nblk = nbricks;
        const int ncot = Cout / 64;
        if (nbricks * ncot > W2_CUS) {
            const long g = (W2_CUS / ncot) & ~7L;
            if (g >= 8 && g < nbricks) nblk = g;
        }
        if (hipFuncSetAttribute((const void*)k3_fwd_wino2x4, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)W4Cfg::LDS_BYTES) != hipSuccess) {
            set_error("conv3d_k3(winograd F(2x4)): cannot reserve %zu bytes of LDS", W4Cfg::LDS_BYTES);
            return SIV_EHIP;
        }
        k3_fwd_wino2x4<<<dim3((unsigned)nblk, (unsigned)ncot), 512, W4Cfg::LDS_BYTES, s>>>(
            x, wpk4, bias, y, N, D, H, W, Cin, Cout, nbd, nbh, nbw, (int)nbricks, stat_part);
        SIV_CHECK_LAUNCH("conv3d_k3(winograd F(2x4))");
        note_conv_kernel("wino2x4");
        return SIV_OK;
    }
    const long total = (long)(Cout / 64) * (Cin / 8) * 48 * 512;
    long g = cdiv(total, 256);
    if (g > 4096) g = 4096;
    bool need_pack;
    const float* wpk = (const float*)pack_slot(w, PF_WINO2D, 0, flip, Cin, Cout, k3_fwd_wino2d_ws_bytes(Cin, Cout), ws, &need_pack);
    if (need_pack) {
        k3_pack_weights_wino2d<<<(unsigned)g, 256, 0, s>>>(w, const_cast<float*>(wpk), Cin, Cout, flip);
        SIV_CHECK_PACK_LAUNCH("conv3d_k3(pack 2-D winograd weights)", wpk);
    }
    const int tpr = w2d_tpr(H, W);
    const int nbd = (D + 3) / 4, nbh = (int)cdiv(H, 2 * (16 / tpr)), nbw = (int)cdiv(W, 2 * tpr);
    const long nbricks = (long)N * nbd * nbh * nbw;
    if (nbricks > 0x7fffffffL) {
        set_error("conv3d_k3(winograd 2-D): too many bricks");
        return SIV_EINVAL;
    }
    // One block per CU working through its bricks as one chunk stream, where the stream's buffer parities
    // repeat per brick (even chunk count) and a brick is long enough to fetch two chunks ahead inside it;
    // the grid's x extent stays a multiple of 8 so that a block's bricks share its XCD.
    long nblk = nbricks;
    const int nchunk = Cin / 8, ncot = Cout / 64;
    if (W2_CHAIN && nchunk >= 4 && nchunk % 2 == 0 && nbricks * ncot > W2_CUS) {
        long g = (W2_CUS / ncot) & ~7L;
        if (g >= 8 && g < nbricks) nblk = g;
    }
    const size_t kneed = k3_fwd_wino2d_ksplit_bytes(N, D, H, W, Cin, Cout);
    const bool ksplit = kneed > 0 && kws != nullptr && kws_bytes >= kneed && (((uintptr_t)kws) & 15) == 0;
    const int kshift = ksplit ? w2d_kshift(nbricks, (long)N * D * H * W, Cin, Cout) : 0;
    long nitems = nbricks;
    if (ksplit) {
        nitems = nbricks << kshift;
        nblk = W2_CUS / ncot;
    }
    auto kern = tpr == 5 ? k3_fwd_wino2d<5, SIV_W2ROUNDS> : k3_fwd_wino2d<4, SIV_W2ROUNDS>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)W2Cfg::LDS_BYTES) != hipSuccess) {
        set_error("conv3d_k3(winograd 2-D): cannot reserve %zu bytes of LDS", W2Cfg::LDS_BYTES);
        return SIV_EHIP;
    }
    kern<<<dim3((unsigned)nblk, (unsigned)(Cout / 64)), 512, W2Cfg::LDS_BYTES, s>>>(
        x, wpk, bias, ksplit ? (float*)kws : y, N, D, H, W, Cin, Cout, nbd, nbh, nbw, (int)nitems, stat_part,
        kshift);
    SIV_CHECK_LAUNCH("conv3d_k3(winograd 2-D)");
    if (ksplit) {
        if (kshift == 2)
            w2d_ksum_k<4><<<W2_KSUM_BLOCKS, 256, 0, s>>>((const float*)kws, bias, y, (long)N * D * H * W, Cout, stat_part);
        else
            w2d_ksum_k<2><<<W2_KSUM_BLOCKS, 256, 0, s>>>((const float*)kws, bias, y, (long)N * D * H * W, Cout, stat_part);
        SIV_CHECK_LAUNCH("conv3d_k3(winograd 2-D, sum of the channel parts)");
    }
    return SIV_OK;
}

}  // namespace siv
