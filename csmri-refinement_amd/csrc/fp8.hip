// fp8 (OCP e4m3fn) operand preparation for the fp8 convolution variant (gconv_fp8.hip).
//
// Per-tensor scaling with a POWER-OF-TWO scale taken from the tensor's current absolute maximum:
//   e = floor(log2(amax)),  scale = 2^(7-e)   =>   |x * scale| < 256 <= 448 (e4m3fn finite maximum)
// so the scaling multiply is exact, nothing saturates, and the only rounding is the fp32 -> e4m3
// round-to-nearest-even of v_cvt_pk_fp8_f32 -- which is what the numpy restatement in
// oracle/csmri_lowprec.py reproduces bit for bit.  Both kernels are plain HBM streams.
#include "common.h"

// scale exponent for an absolute maximum (shared by the quantize kernel and the scale writer)
__device__ __forceinline__ int fp8_scale_exp(float amax) {
  if (!(amax > 0.f) || amax > 3.0e38f) return 0;            // all-zero / non-finite tensors: scale 1
  int e = (int)((__float_as_uint(amax) >> 23) & 0xff) - 127;  // floor(log2(amax)) for normal numbers
  if (e < -100) e = -100;                                    // denormal-range maxima: keep the scale finite
  return 7 - e;
}
__device__ __forceinline__ float pow2f(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }

template <int DT>
__global__ __launch_bounds__(256) void absmax_kernel(const void* __restrict__ x, long long nvec, unsigned* amax_bits) {
  // nvec: 16-byte vectors.  |x| as raw bits orders like an unsigned integer, so the maximum is exact and
  // order-independent: atomicMax on the bit pattern is deterministic.
  unsigned m = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
    const u32x4_t v = ((const u32x4_t*)x)[i];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (DT == CSMRI_F32) {
        const unsigned a = v[q] & 0x7fffffffu;
        m = (a <= 0x7f800000u && a > m) ? a : m;             // NaNs are skipped
      } else {
        const unsigned lo = (v[q] << 16) & 0x7fffffffu, hi = v[q] & 0x7fff0000u;
        m = (lo <= 0x7f800000u && lo > m) ? lo : m;
        m = (hi <= 0x7f800000u && hi > m) ? hi : m;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const unsigned t = __shfl_xor(m, o); m = t > m ? t : m; }
  if ((threadIdx.x & 63) == 0 && m) atomicMax(amax_bits, m);
}

extern "C" int csmri_absmax(int dtype, const void* x, long long n, float* amax, void* stream) {
  CSMRI_CHECK_ARG(x && amax && n > 0 && (dtype == CSMRI_F32 || dtype == CSMRI_BF16));
  const int per = dtype == CSMRI_F32 ? 4 : 8;
  CSMRI_CHECK_ARG(n % per == 0);
  if ((uintptr_t)x & 15) return CSMRI_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(amax, 0, sizeof(float), st);
  if (e != hipSuccess) return (int)e;
  const long long nvec = n / per;
  long long blocks = (nvec + 256 * 4 - 1) / (256 * 4);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  if (dtype == CSMRI_F32) hipLaunchKernelGGL(absmax_kernel<CSMRI_F32>, dim3((int)blocks), dim3(256), 0, st, x, nvec, (unsigned*)amax);
  else hipLaunchKernelGGL(absmax_kernel<CSMRI_BF16>, dim3((int)blocks), dim3(256), 0, st, x, nvec, (unsigned*)amax);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

__device__ __forceinline__ unsigned pack4_fp8(float a, float b, float c, float d) {
  int r = 0;
  r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, r, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
  return (unsigned)r;
}

template <int DT>
__global__ __launch_bounds__(256) void quantize_fp8_kernel(const void* __restrict__ x, void* __restrict__ q, long long n16,
                                                           const float* __restrict__ amax, float* scales) {
  // n16: groups of 16 elements (one 16-byte fp8 vector out)
  const int se = fp8_scale_exp(*amax);
  const float s = pow2f(se);
  if (blockIdx.x == 0 && threadIdx.x == 0 && scales) { scales[0] = s; scales[1] = pow2f(-se); }
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) {
    float f[16];
    if (DT == CSMRI_F32) {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const f32x4_t t = ((const f32x4_t*)x)[i * 4 + v];
#pragma unroll
        for (int k = 0; k < 4; ++k) f[v * 4 + k] = t[k];
      }
    } else {
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const u32x4_t t = ((const u32x4_t*)x)[i * 2 + v];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          f[v * 8 + 2 * k] = __uint_as_float(t[k] << 16);
          f[v * 8 + 2 * k + 1] = __uint_as_float(t[k] & 0xffff0000u);
        }
      }
    }
    u32x4_t o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = pack4_fp8(f[4 * k] * s, f[4 * k + 1] * s, f[4 * k + 2] * s, f[4 * k + 3] * s);
    ((u32x4_t*)q)[i] = o;
  }
}

extern "C" int csmri_quantize_fp8(int dtype, const void* x, void* q, long long n, const float* amax, float* scales,
                                  void* stream) {
  CSMRI_CHECK_ARG(x && q && amax && n > 0 && n % 16 == 0 && (dtype == CSMRI_F32 || dtype == CSMRI_BF16));
  if (((uintptr_t)x | (uintptr_t)q) & 15) return CSMRI_E_ALIGN;
  const long long n16 = n / 16;
  long long blocks = (n16 + 256 * 2 - 1) / (256 * 2);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CSMRI_F32) hipLaunchKernelGGL(quantize_fp8_kernel<CSMRI_F32>, dim3((int)blocks), dim3(256), 0, st, x, q, n16, amax, scales);
  else hipLaunchKernelGGL(quantize_fp8_kernel<CSMRI_BF16>, dim3((int)blocks), dim3(256), 0, st, x, q, n16, amax, scales);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

// Delayed scaling for a chain of fp8 tensors (the frozen VGG stack): n tensors, amax[j] = the |x| maximum the producers of
// tensor j accumulated during THIS step (atomicMax of bit patterns; csmri_gconv_desc.out_amax, csmri_maxpool2_q), scales[2j],
// scales[2j+1] = the quantisation / dequantisation scale pair the NEXT step uses: 2^(7 - floor(log2(amax)) - margin), so
// that values up to 2^margin x 1.75 times this step's maximum still fit e4m3fn's finite range.  amax[j] is cleared; a
// tensor that saw no data (amax 0) keeps its scales.
__global__ void fp8_scales_update_kernel(float* amax, float* scales, int n, int margin) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const float am = amax[j];
  if (am > 0.f) {
    const int se = fp8_scale_exp(am) - margin;
    scales[2 * j] = pow2f(se); scales[2 * j + 1] = pow2f(-se);
  }
  amax[j] = 0.f;
}
extern "C" int csmri_fp8_scales_update(float* amax, float* scales, int n, int margin, void* stream) {
  CSMRI_CHECK_ARG(amax && scales && n > 0 && margin >= 0 && margin <= 4);
  hipLaunchKernelGGL(fp8_scales_update_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, amax, scales, n, margin);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}
