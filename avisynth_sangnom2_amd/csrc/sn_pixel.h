// sn_pixel.h -- per-sample-type arithmetic of the SangNom2 path, device side.
//
// Semantics follow the reference's opt=0 helpers (/root/reference/src/SangNom2.cpp:25-72):
// integer narrowing wraps modulo 2^(8*sizeof T), `>>` on the SangNom sum is arithmetic, float
// code keeps the reference's operation order (the library is built with -ffp-contract=off so
// no multiply-add is fused).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sn {

template <class T>
struct Px;

template <>
struct Px<uint8_t> {
    using W = int32_t;  // working type (the reference uses int16_t; the values are identical)
    static __device__ __forceinline__ W narrow(W v) { return v & 0xFF; }
    static __device__ __forceinline__ W sg(W p1, W p2, W p3) { return ((4 * p1 + 5 * p2 - p3) >> 3) & 0xFF; }
    static __device__ __forceinline__ W adiff(W a, W b) { W d = a - b; return d < 0 ? -d : d; }
    static __device__ __forceinline__ W avg(W a, W b) { return (a + b + 1) >> 1; }
    static __device__ __forceinline__ W div16(W s) { return (s >> 4) & 0xFF; }  // s >= 0
    static __device__ __forceinline__ W sum3(W a, W b, W c) { return a + b + c; }
};

template <>
struct Px<uint16_t> {
    using W = int32_t;
    static __device__ __forceinline__ W narrow(W v) { return v & 0xFFFF; }
    static __device__ __forceinline__ W sg(W p1, W p2, W p3) { return ((4 * p1 + 5 * p2 - p3) >> 3) & 0xFFFF; }
    static __device__ __forceinline__ W adiff(W a, W b) { W d = a - b; return d < 0 ? -d : d; }
    static __device__ __forceinline__ W avg(W a, W b) { return (a + b + 1) >> 1; }
    static __device__ __forceinline__ W div16(W s) { return (s >> 4) & 0xFFFF; }
    static __device__ __forceinline__ W sum3(W a, W b, W c) { return a + b + c; }
};

template <>
struct Px<float> {
    using W = float;
    static __device__ __forceinline__ W narrow(W v) { return v; }
    static __device__ __forceinline__ W sg(W p1, W p2, W p3)
    {
        float s = (p1 * 4.0f + p2 * 5.0f) - p3;  // SangNom2.cpp:70
        return s * 0.125f;
    }
    static __device__ __forceinline__ W adiff(W a, W b) { return __builtin_fabsf(a - b); }
    static __device__ __forceinline__ W avg(W a, W b) { return (a + b) * 0.5f; }
    static __device__ __forceinline__ W div16(W s) { return s * 0.0625f; }  // == s / 16 exactly
    static __device__ __forceinline__ W sum3(W a, W b, W c) { return (a + b) + c; }
};

}  // namespace sn
