// fp32 values as three bf16 pieces (x = h + m + l, 24+ significant bits) for the kernels that form
// fp32 products on the bf16 matrix cores; see conv_wino_b3.hip for the scheme.
#pragma once
#include <hip/hip_runtime.h>

namespace siv {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// round-to-nearest-even pair conversion (v_cvt_pk_bf16_f32): low half = a, high half = b
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    const f32x2 v = {a, b};
    const bf16x2 r = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __uint_as_float(p & 0xFFFF0000u); }

// four fp32 values -> three pieces of four bf16 each
__device__ __forceinline__ void split3(float4 v, uint2& h, uint2& m, uint2& l) {
    h.x = pk_bf16(v.x, v.y); h.y = pk_bf16(v.z, v.w);
    const float rx = v.x - bf_lo(h.x), ry = v.y - bf_hi(h.x), rz = v.z - bf_lo(h.y), rw = v.w - bf_hi(h.y);
    m.x = pk_bf16(rx, ry); m.y = pk_bf16(rz, rw);
    l.x = pk_bf16(rx - bf_lo(m.x), ry - bf_hi(m.x));
    l.y = pk_bf16(rz - bf_lo(m.y), rw - bf_hi(m.y));
}

// one value -> (h, m, l)
__device__ __forceinline__ void split3_scalar(float v, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)v;
    const float r1 = v - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}

}  // namespace siv
