// Shared device helpers for the gfx950 kernels: storage types, bit-level
// f16 / bf16 / e5m2 conversions, wave64 reductions, and the host-side error slot.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>

#include <type_traits>
#include <utility>

#include "../../include/splitwise_hip.h"

namespace sw {

// ---------------------------------------------------------------- host errors
void set_error(const char* fmt, ...);

// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): loops whose index must be a
// compile-time constant (register arrays)
template <int... I, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, f);
}


#define SW_CHECK_ARG(cond, ...)                 \
  do {                                          \
    if (!(cond)) {                              \
      ::sw::set_error(__VA_ARGS__);             \
      return SW_ERR_INVALID;                    \
    }                                           \
  } while (0)

#define SW_CHECK_HIP(expr)                                                   \
  do {                                                                       \
    hipError_t e__ = (expr);                                                 \
    if (e__ != hipSuccess) {                                                 \
      ::sw::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                      __FILE__, __LINE__);                                   \
      return SW_ERR_HIP;                                                     \
    }                                                                        \
  } while (0)

#define SW_CHECK_LAUNCH()                                                    \
  do {                                                                       \
    hipError_t e__ = hipGetLastError();                                      \
    if (e__ != hipSuccess) {                                                 \
      ::sw::set_error("kernel launch failed: %s (%s:%d)",                    \
                      hipGetErrorString(e__), __FILE__, __LINE__);           \
      return SW_ERR_HIP;                                                     \
    }                                                                        \
  } while (0)

inline int dtype_size(int dt) {
  switch (dt) {
    case SW_F32: return 4;
    case SW_F16: return 2;
    case SW_BF16: return 2;
    case SW_FP8_E5M2: return 1;
    default: return 0;
  }
}

// ------------------------------------------------------------ storage types
// Activations and caches are handled as raw bit patterns; arithmetic is fp32.
struct f16_t { uint16_t bits; };
struct bf16_t { uint16_t bits; };
struct e5m2_t { uint8_t bits; };

__device__ __forceinline__ float bits_to_float(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t float_to_bits(float f) { return __float_as_uint(f); }

__device__ __forceinline__ float f16_bits_to_float(uint16_t h) {
  return __half2float(__ushort_as_half(h));
}
__device__ __forceinline__ uint16_t float_to_f16_bits(float f) {
  return __half_as_ushort(__float2half_rn(f));
}
__device__ __forceinline__ float bf16_bits_to_float(uint16_t h) {
  return __uint_as_float(static_cast<uint32_t>(h) << 16);
}
// Round-to-nearest-even; NaN stays a (quiet) NaN.
// (gfx950 has the conversion as an instruction, v_cvt_pk_bf16_f32, two values at a time: the integer form of the
// same rounding costs seven vector instructions per value)
__device__ __forceinline__ uint16_t float_to_bf16_bits(float f) {
  return __builtin_bit_cast(uint16_t, static_cast<__bf16>(f));
}

// e5m2 is the high byte of an IEEE half, so e5m2 -> half is exact.
__device__ __forceinline__ uint16_t e5m2_to_f16_bits(uint8_t b) {
  return static_cast<uint16_t>(b) << 8;
}
__device__ __forceinline__ float e5m2_to_float(uint8_t b) {
  return f16_bits_to_float(e5m2_to_f16_bits(b));
}

// half bits -> e5m2, round-to-nearest-even on the dropped 8 mantissa bits,
// saturating to the largest finite value (0x7B = 57344); Inf saturates too.
__device__ __forceinline__ uint8_t f16_bits_to_e5m2(uint16_t h) {
  const uint32_t sign = (h >> 8) & 0x80u;
  const uint32_t a = h & 0x7fffu;
  if (a > 0x7c00u) return static_cast<uint8_t>(sign | 0x7fu);
  uint32_t r = (a + 0x7fu + ((a >> 8) & 1u)) >> 8;
  if (r >= 0x7cu) r = 0x7bu;
  return static_cast<uint8_t>(sign | r);
}

// float -> e5m2 in ONE rounding step (never through half: that would round twice).
__device__ __forceinline__ uint8_t float_to_e5m2(float f) {
  const uint32_t u = __float_as_uint(f);
  const uint32_t sign = (u >> 24) & 0x80u;
  const uint32_t a = u & 0x7fffffffu;
  if (a > 0x7f800000u) return static_cast<uint8_t>(sign | 0x7fu);
  uint32_t r;
  if (a >= 0x38800000u) {                    // |f| >= 2^-14: normal e5m2 result
    const uint32_t t = a - 0x38000000u;      // rebias exponent 127 -> 15
    r = (t + 0xfffffu + ((t >> 21) & 1u)) >> 21;
    if (r >= 0x7cu) r = 0x7bu;
  } else {                                   // subnormal e5m2: multiples of 2^-16
    r = static_cast<uint32_t>(rintf(__uint_as_float(a) * 65536.0f));
  }
  return static_cast<uint8_t>(sign | r);
}

template <typename T> struct Conv;
template <> struct Conv<float> {
  using store_t = float;
  __device__ static __forceinline__ float to_float(float v) { return v; }
  __device__ static __forceinline__ float from_float(float v) { return v; }
  __device__ static __forceinline__ uint8_t to_e5m2(float v) { return float_to_e5m2(v); }
  __device__ static __forceinline__ float from_e5m2(uint8_t b) { return e5m2_to_float(b); }
};
template <> struct Conv<f16_t> {
  using store_t = uint16_t;
  __device__ static __forceinline__ float to_float(uint16_t v) { return f16_bits_to_float(v); }
  __device__ static __forceinline__ uint16_t from_float(float v) { return float_to_f16_bits(v); }
  __device__ static __forceinline__ uint8_t to_e5m2(uint16_t v) { return f16_bits_to_e5m2(v); }
  __device__ static __forceinline__ uint16_t from_e5m2(uint8_t b) { return e5m2_to_f16_bits(b); }
};
template <> struct Conv<bf16_t> {
  using store_t = uint16_t;
  __device__ static __forceinline__ float to_float(uint16_t v) { return bf16_bits_to_float(v); }
  __device__ static __forceinline__ uint16_t from_float(float v) { return float_to_bf16_bits(v); }
  __device__ static __forceinline__ uint8_t to_e5m2(uint16_t v) { return float_to_e5m2(bf16_bits_to_float(v)); }
  // every e5m2 value (2 mantissa bits, exponent >= 2^-16) is exact in bf16
  __device__ static __forceinline__ uint16_t from_e5m2(uint8_t b) { return float_to_bf16_bits(e5m2_to_float(b)); }
};

// ------------------------------------------------------------ wave64 helpers
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

}  // namespace sw
