// Internal helpers shared by the gfx950 kernels of libcsmri_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include "../../include/csmri_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

// Launch wrapper: clear any stale (sticky-free) error left by earlier, unrelated HIP
// calls of the host process so that CSMRI_LAUNCH_CHECK reports THIS launch only.
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kern_, grid_, block_, lds_, stream_, ...) \
  do { (void)hipGetLastError(); kern_<<<(grid_), (block_), (lds_), (stream_)>>>(__VA_ARGS__); } while (0)

#define CSMRI_CHECK_ARG(cond) do { if (!(cond)) return CSMRI_E_ARG; } while (0)
#define CSMRI_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

// Raise a kernel's dynamic-LDS limit to `bytes_` once per call site and size.  The cache is an atomic that only grows
// (hipFuncSetAttribute is idempotent, so two host threads racing here at worst both make the call): entry points stay
// re-entrant across host threads and no launch pays the attribute call twice.
#define CSMRI_SET_MAX_LDS(kern_, bytes_) \
  do { \
    static std::atomic<int> lds_set_{0}; \
    const int want_ = (int)(bytes_); \
    if (want_ > 48 * 1024 && lds_set_.load(std::memory_order_acquire) < want_) { \
      hipError_t e_ = hipFuncSetAttribute((const void*)(kern_), hipFuncAttributeMaxDynamicSharedMemorySize, want_); \
      if (e_ != hipSuccess) return (int)e_; \
      int cur_ = lds_set_.load(std::memory_order_relaxed); \
      while (cur_ < want_ && !lds_set_.compare_exchange_weak(cur_, want_, std::memory_order_release)) {} \
    } \
  } while (0)

template <int DT> struct DTraits;
template <> struct DTraits<CSMRI_F32> {
  typedef float T;
  static constexpr int ES = 4;    // element bytes
  static constexpr int VE = 4;    // elements per 16-byte vector
  static constexpr int BKE = 16;  // elements per 64-byte K chunk
};
template <> struct DTraits<CSMRI_BF16> {
  typedef __bf16 T;
  static constexpr int ES = 2;
  static constexpr int VE = 8;
  static constexpr int BKE = 32;
};

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
  return __uint_as_float(((unsigned int)b) << 16);
}
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
  __bf16 h = (__bf16)f;                    // v_cvt_pk_bf16_f32: RNE, NaN-preserving
  return __builtin_bit_cast(unsigned short, h);
}

// four floats -> four bf16 (RNE) with two v_cvt_pk_bf16_f32: the scalar form costs a conversion, a shift and an
// or per pair on top (PMC: 3-4 VALU instructions per MFMA in the conv kernels, much of it epilogue)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ u32x2_t pack4_bf16(f32x4_t v) {
  u32x2_t u;
  u[0] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){v[0], v[1]}, bf16x2_t));
  u[1] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){v[2], v[3]}, bf16x2_t));
  return u;
}

// typed scalar load/store by runtime dtype
__device__ __forceinline__ float load_elem(const void* p, long long idx, int dt) {
  if (dt == CSMRI_F32) return ((const float*)p)[idx];
  return bf16_bits_to_f32(((const unsigned short*)p)[idx]);
}
__device__ __forceinline__ void store_elem(void* p, long long idx, int dt, float v) {
  if (dt == CSMRI_F32) ((float*)p)[idx] = v;
  else ((unsigned short*)p)[idx] = f32_to_bf16_bits(v);
}

// 4 consecutive channels
__device__ __forceinline__ f32x4_t load4(const void* p, long long idx, int dt) {
  f32x4_t r;
  if (dt == CSMRI_F32) {
    r = *(const f32x4_t*)((const float*)p + idx);
  } else {
    u32x2_t u = *(const u32x2_t*)((const unsigned short*)p + idx);
    r[0] = __uint_as_float(u[0] << 16); r[1] = __uint_as_float(u[0] & 0xffff0000u);
    r[2] = __uint_as_float(u[1] << 16); r[3] = __uint_as_float(u[1] & 0xffff0000u);
  }
  return r;
}
__device__ __forceinline__ void store4(void* p, long long idx, int dt, f32x4_t v) {
  if (dt == CSMRI_F32) {
    *(f32x4_t*)((float*)p + idx) = v;
  } else {
    *(u32x2_t*)((unsigned short*)p + idx) = pack4_bf16(v);
  }
}

// raw 4-channel load (no conversion: the conversions of a batch come after ALL its loads were issued)
template <int DT> struct raw4 { typedef f32x4_t t; };
template <> struct raw4<CSMRI_BF16> { typedef u32x2_t t; };
template <int DT> __device__ __forceinline__ typename raw4<DT>::t ldraw(const void* p, long long idx) {
  if constexpr (DT == CSMRI_F32) return *(const f32x4_t*)((const float*)p + idx);
  else return *(const u32x2_t*)((const unsigned short*)p + idx);
}
template <int DT> __device__ __forceinline__ f32x4_t cvt4(typename raw4<DT>::t u) {
  if constexpr (DT == CSMRI_F32) return u;
  else return (f32x4_t){__uint_as_float(u[0] << 16), __uint_as_float(u[0] & 0xffff0000u),
                        __uint_as_float(u[1] << 16), __uint_as_float(u[1] & 0xffff0000u)};
}

__device__ __forceinline__ int reflect_idx(int u, int n) {
  // mirror without repeating the edge (nn.ReflectionPad2d); valid for |pad| < n
  u = u < 0 ? -u : u;
  return u >= n ? 2 * (n - 1) - u : u;
}

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline int dtype_size(int dt) { return dt == CSMRI_F32 ? 4 : 2; }

// XCD-aware tile id: blocks are dealt round-robin over the 8 XCDs, so give each
// XCD a contiguous chunk of the tile space (bijective for any grid size).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int nx = 8;
  int xcd = bid % nx, idx = bid / nx;
  int q = nwg / nx, r = nwg % nx;
  int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// Saturation in front of v_cvt_pk_fp8_f32: with DELAYED scaling (the scale of step t comes from the maximum seen at
// step t - 1, one bit of headroom) a tensor that grows more than ~3.5x between two forwards would leave the e4m3fn range
// (448), and the unclamped convert does not saturate.  Inside the range the clamp is the identity, so the bytes stay
// bit for bit csmri_quantize_fp8's.
__device__ __forceinline__ float sat_e4m3(float x) { return __builtin_amdgcn_fmed3f(x, -448.f, 448.f); }
