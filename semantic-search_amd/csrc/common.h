// common.h - shared host/device definitions for libssmi (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <string>
#include <vector>

namespace ssmi {

// A top-k candidate is one 64-bit key: (orderable score bits << 32) | (~id).
// Larger key == better candidate: higher score first, then lower id.
// Key 0 is "empty" (decodes to id -1 / score -FLT_MAX).
typedef unsigned long long skey_t;

__host__ __device__ inline uint32_t f32_to_ord(float f) {
#ifdef __HIP_DEVICE_COMPILE__
    uint32_t u = __float_as_uint(f);
#else
    union { float f; uint32_t u; } c; c.f = f; uint32_t u = c.u;
#endif
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float ord_to_f32(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
#ifdef __HIP_DEVICE_COMPILE__
    return __uint_as_float(u);
#else
    union { float f; uint32_t u; } c; c.u = u; return c.f;
#endif
}
__host__ __device__ inline skey_t make_key(float score, uint32_t id) {
    return ((skey_t)f32_to_ord(score) << 32) | (skey_t)(0xFFFFFFFFu - id);
}
__host__ __device__ inline float key_score(skey_t k) { return ord_to_f32((uint32_t)(k >> 32)); }
__host__ __device__ inline uint32_t key_id(skey_t k) { return 0xFFFFFFFFu - (uint32_t)(k & 0xFFFFFFFFu); }

#define SSMI_NEG_MAX (-3.402823466e+38f)

// element types of the row stores (include/ssmi.h: SSMI_F32 / SSMI_F16 / SSMI_BF16)
__host__ __device__ inline int dtype_elsize(int dtype) { return dtype == 0 ? 4 : 2; }
__host__ __device__ inline bool dtype_is16(int dtype) { return dtype == 1 || dtype == 2; }
__host__ __device__ inline bool dtype_known(int dtype) { return dtype >= 0 && dtype <= 2; }

// bfloat16 storage element: the upper half of an fp32.  Widening is a shift (exact), narrowing rounds to nearest
// even (NaN stays NaN); no arithmetic is ever done in this type.
struct bf16_t {
    uint16_t x;
    bf16_t() = default;
    __host__ __device__ explicit bf16_t(float f) {
#ifdef __HIP_DEVICE_COMPILE__
        uint32_t u = __float_as_uint(f);
#else
        union { float f; uint32_t u; } c; c.f = f; uint32_t u = c.u;
#endif
        if ((u & 0x7fffffffu) > 0x7f800000u) x = (uint16_t)((u >> 16) | 0x0040u);
        else x = (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    }
    __host__ __device__ operator float() const {
#ifdef __HIP_DEVICE_COMPILE__
        return __uint_as_float((uint32_t)x << 16);
#else
        union { float f; uint32_t u; } c; c.u = (uint32_t)x << 16; return c.f;
#endif
    }
};
static_assert(sizeof(bf16_t) == 2, "bf16_t is a 16-bit storage element");

// Row store handed to kernels: a table of equally sized segments.
struct RowView {
    const void* const* segs;  // device table of segment base pointers
    int seg_shift;            // rows per segment = 1 << seg_shift
    int ld;                   // padded row length in elements (multiple of 8)
    long long nrows;
};

static const int kMaxK = 1024;       // largest k the fused top-k supports
static const int kWave = 64;

}  // namespace ssmi
