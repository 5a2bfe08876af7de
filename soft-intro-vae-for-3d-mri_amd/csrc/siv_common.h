// Shared host/device helpers for libsiv (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "siv.h"

namespace siv {

void set_error(const char* fmt, ...);
int conv_path();   // 0 auto, 1 generic, 2 specialised

#define SIV_CHECK_ARG(cond, ...)                     \
    do {                                             \
        if (!(cond)) {                               \
            siv::set_error(__VA_ARGS__);             \
            return SIV_EINVAL;                       \
        }                                            \
    } while (0)

#define SIV_CHECK_WS(need, have, what)                                                   \
    do {                                                                                 \
        if ((size_t)(need) > (size_t)(have)) {                                           \
            siv::set_error("%s: workspace too small (need %zu, have %zu)", what,         \
                           (size_t)(need), (size_t)(have));                              \
            return SIV_EWORKSPACE;                                                       \
        }                                                                                \
    } while (0)

// to be used right after a kernel launch
#define SIV_CHECK_LAUNCH(what)                                                           \
    do {                                                                                 \
        hipError_t e__ = hipGetLastError();                                              \
        if (e__ != hipSuccess) {                                                         \
            siv::set_error("%s: launch failed: %s", what, hipGetErrorString(e__));       \
            return SIV_EHIP;                                                             \
        }                                                                                \
    } while (0)

// Packed-weight cache (siv_api.hip).  Kernel families that repack their weights ask for the place to
// pack into: a cached slot when the weight pointer lies inside memory the caller declared stable
// (siv_pack_cache_add_weights) and the cache has room, otherwise `ws`.  *need_pack is false when the slot
// already holds this packing (same pointer, family, flip and channel counts since the last invalidation).
enum PackFamily { PF_MFMA = 1, PF_WINO, PF_WINO_B3, PF_WINO2D, PF_UP_PROD, PF_UP_B3, PF_UP, PF_UPD_B3, PF_UPD_BOX, PF_UPD };
void* pack_slot(const void* w, int family, int variant, int flip, int Cin, int Cout, size_t bytes, void* ws,
                bool* need_pack);
// A packing kernel that was to fill `slot` did not launch: the cache forgets the slot, so that no later
// call takes it for a finished packing.
void pack_slot_failed(const void* slot);
void note_conv_kernel(const char* name);      // overrides the name siv_last_conv_kernel() reports for this call
// to be used right after the launch of a packing kernel whose destination came from pack_slot
#define SIV_CHECK_PACK_LAUNCH(what, slot)                                                \
    do {                                                                                 \
        hipError_t e__ = hipGetLastError();                                              \
        if (e__ != hipSuccess) {                                                         \
            siv::pack_slot_failed(slot);                                                 \
            siv::set_error("%s: launch failed: %s", what, hipGetErrorString(e__));       \
            return SIV_EHIP;                                                             \
        }                                                                                \
    } while (0)

// 16 bytes through a buffer descriptor: byte offset = per-lane voff + wave-uniform soff (+ an immediate the
// compiler folds).  No 64-bit address arithmetic on the vector ALU, and a lane whose voff lies beyond the
// descriptor's size reads zeros without touching memory (the range check looks at voff and the immediate,
// so whatever decides validity belongs in voff).
#ifdef __HIPCC__
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void* base, long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
// Packed two-element fp32 arithmetic, spelled as the instructions: on a SIMD that runs fp32 matrix
// instructions every vector instruction takes matrix-pipe time, and the compiler splits a packed operation
// whose halves are consumed separately into two scalar ones.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t pk_add(f32x2_t a, f32x2_t b) {
    f32x2_t r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2_t pk_sub(f32x2_t a, f32x2_t b) {
    f32x2_t r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2_t pk_fma(f32x2_t a, f32x2_t b, f32x2_t c) {      // a * b + c
    f32x2_t r;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
struct f32x2x2 { f32x2_t lo, hi; };                       // four channels as two register pairs
__device__ __forceinline__ f32x2x2 buf_load16p(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    const f32x4_t f = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    return f32x2x2{f32x2_t{f.x, f.y}, f32x2_t{f.z, f.w}};
}
__device__ __forceinline__ float4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    const f32x4_t f = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    return make_float4(f.x, f.y, f.z, f.w);
}
#endif

// BatchNorm of the layer in front, for kernels that apply it (and the LeakyReLU behind it) to their
// operand instead of reading an activated tensor
struct BnPrologue {
    const float* mean;
    const float* var;
    const float* gamma;
    const float* beta;
    float eps, slope;
};

static inline long cdiv(long a, long b) { return (a + b - 1) / b; }
static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
// what is left of a workspace behind its first `used` bytes (256-byte aligned; NULL / 0 if nothing is)
static inline void* ws_tail(void* ws, size_t ws_bytes, size_t used) {
    used = align256(used);
    return used < ws_bytes ? (void*)((char*)ws + used) : nullptr;
}
static inline size_t ws_tail_bytes(size_t ws_bytes, size_t used) {
    used = align256(used);
    return used < ws_bytes ? ws_bytes - used : 0;
}

// wave64 reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Deterministic block sum for 256-thread blocks (4 waves); result valid in every thread.
__device__ __forceinline__ float block_sum_256(float v, float* sh /* >= 4 floats */) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// ---- Philox-4x32-10: counter = element quad, high counter word = stream id, key = seed -----------
struct u4 { uint32_t x, y, z, w; };
__device__ __forceinline__ u4 philox4x32_10(uint64_t ctr_lo, uint64_t ctr_hi, uint64_t key) {
    uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32);
    uint32_t c2 = (uint32_t)ctr_hi, c3 = (uint32_t)(ctr_hi >> 32);
    uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return u4{c0, c1, c2, c3};
}
// the Bernoulli(1 - p) draw behind one 32-bit output: dropout keeps the element when this is true
__device__ __forceinline__ bool philox_keeps(uint32_t r, float p) {
    return ((float)(r >> 8) * (1.0f / 16777216.0f)) >= p;
}
// epoch (optional, device memory): added to the stream id in steps of 2^20, so that a launch captured in a
// HIP graph draws fresh numbers on every replay once the counter has been advanced
#define SIV_EPOCH_STRIDE (1ull << 20)
// Dropout mask drawn where it is used instead of read from a tensor: element i of the masked tensor is
// component i % 4 of philox(i / 4, stream id, seed) -- the very bits siv_dropout_keep writes.
struct PhiloxMask {
    uint64_t seed, sid;
    const unsigned long long* epoch;
    int on;
};

// fp32 -> bf16 bits with round-to-nearest-even (NaN stays NaN)
__device__ __forceinline__ uint32_t bf16_rne(float f) {
    const uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;        // quiet NaN
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

__device__ __forceinline__ float leaky(float z, float slope) { return z > 0.f ? z : z * slope; }
__device__ __forceinline__ float leaky_grad(float z, float slope) { return z > 0.f ? 1.f : slope; }

}  // namespace siv
