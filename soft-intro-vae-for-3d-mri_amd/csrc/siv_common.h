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

__device__ __forceinline__ float leaky(float z, float slope) { return z > 0.f ? z : z * slope; }
__device__ __forceinline__ float leaky_grad(float z, float slope) { return z > 0.f ? 1.f : slope; }

}  // namespace siv
