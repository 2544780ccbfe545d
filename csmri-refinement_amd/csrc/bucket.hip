// Gradient-bucket transport kernels of the data-parallel step (training/distributed.py GradBucket).
// The reference reduces gradients inside nn.DataParallel (utils/custom_data_parallel.py:26-35,
// utils/__init__.py:59-68); here a sub-bucket of a model's flat fp32 gradient buffer travels as bf16:
//   pack   g[a:b] fp32 -> send bf16 [world][per] (RNE, zero tail)            csmri_bucket_pack_bf16
//   RCCL   all_to_all(send -> recv)
//   reduce mine[j] = bf16( sum_r float(recv[r][j]) ), fp32 sum in rank order  csmri_bucket_reduce
//   RCCL   all_gather(mine -> send)
//   unpack g[a:b] = float(send[0:b-a])                                        csmri_bucket_unpack_bf16
// All three are pure streaming passes (HBM-bound): 16-byte accesses, one thread per 8 elements.
#include "common.h"

static inline int bucket_grid(long long vecs) {
  long long b = (vecs + 255) / 256;
  if (b < 1) b = 1;
  if (b > 4096) b = 4096;
  return (int)b;
}

__device__ __forceinline__ u32x4_t pack8(f32x4_t a, f32x4_t b) {
  const u32x2_t lo = pack4_bf16(a), hi = pack4_bf16(b);
  return (u32x4_t){lo[0], lo[1], hi[0], hi[1]};
}
__device__ __forceinline__ void widen8(u32x4_t u, f32x4_t& a, f32x4_t& b) {
  a[0] = __uint_as_float(u[0] << 16); a[1] = __uint_as_float(u[0] & 0xffff0000u);
  a[2] = __uint_as_float(u[1] << 16); a[3] = __uint_as_float(u[1] & 0xffff0000u);
  b[0] = __uint_as_float(u[2] << 16); b[1] = __uint_as_float(u[2] & 0xffff0000u);
  b[2] = __uint_as_float(u[3] << 16); b[3] = __uint_as_float(u[3] & 0xffff0000u);
}

__global__ __launch_bounds__(256) void bucket_pack_kernel(const float* __restrict__ g, long long n,
                                                          unsigned short* __restrict__ out, long long n_pad) {
  const long long vecs = n_pad / 8;
  for (long long v = blockIdx.x * (long long)blockDim.x + threadIdx.x; v < vecs; v += (long long)gridDim.x * blockDim.x) {
    const long long i = v * 8;
    if (i + 8 <= n) {
      const f32x4_t a = *(const f32x4_t*)(g + i), b = *(const f32x4_t*)(g + i + 4);
      *(u32x4_t*)(out + i) = pack8(a, b);
    } else {
      for (int e = 0; e < 8; ++e) out[i + e] = i + e < n ? f32_to_bf16_bits(g[i + e]) : (unsigned short)0;
    }
  }
}

__global__ __launch_bounds__(256) void bucket_reduce_kernel(const unsigned short* __restrict__ recv, int world,
                                                            long long per, unsigned short* __restrict__ mine) {
  const long long vecs = per / 8;
  for (long long v = blockIdx.x * (long long)blockDim.x + threadIdx.x; v < vecs; v += (long long)gridDim.x * blockDim.x) {
    f32x4_t sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < world; ++r) {                     // rank order: the same sum on every rank count's owner
      f32x4_t a, b;
      widen8(*(const u32x4_t*)(recv + r * per + v * 8), a, b);
      sa += a; sb += b;
    }
    *(u32x4_t*)(mine + v * 8) = pack8(sa, sb);
  }
}

__global__ __launch_bounds__(256) void bucket_unpack_kernel(const unsigned short* __restrict__ src, long long n,
                                                            float* __restrict__ g) {
  const long long vecs = (n + 7) / 8;
  for (long long v = blockIdx.x * (long long)blockDim.x + threadIdx.x; v < vecs; v += (long long)gridDim.x * blockDim.x) {
    const long long i = v * 8;
    if (i + 8 <= n) {
      f32x4_t a, b;
      widen8(*(const u32x4_t*)(src + i), a, b);
      *(f32x4_t*)(g + i) = a; *(f32x4_t*)(g + i + 4) = b;
    } else {
      for (long long j = i; j < n; ++j) g[j] = bf16_bits_to_f32(src[j]);
    }
  }
}

extern "C" int csmri_bucket_pack_bf16(const float* g, long long n, void* send, long long n_padded, void* stream) {
  CSMRI_CHECK_ARG(g && send && n > 0 && n_padded >= n && n_padded % 8 == 0);
  if (((uintptr_t)g | (uintptr_t)send) & 15) return CSMRI_E_ALIGN;
  hipLaunchKernelGGL(bucket_pack_kernel, dim3(bucket_grid(n_padded / 8)), dim3(256), 0, (hipStream_t)stream, g, n,
                     (unsigned short*)send, n_padded);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

extern "C" int csmri_bucket_reduce(const void* recv, int world, long long per, void* mine, void* stream) {
  CSMRI_CHECK_ARG(recv && mine && world >= 1 && per > 0 && per % 8 == 0);
  if (((uintptr_t)recv | (uintptr_t)mine) & 15) return CSMRI_E_ALIGN;
  hipLaunchKernelGGL(bucket_reduce_kernel, dim3(bucket_grid(per / 8)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)recv, world, per, (unsigned short*)mine);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

extern "C" int csmri_bucket_unpack_bf16(const void* src, long long n, float* g, void* stream) {
  CSMRI_CHECK_ARG(src && g && n > 0);
  if (((uintptr_t)src | (uintptr_t)g) & 15) return CSMRI_E_ALIGN;
  hipLaunchKernelGGL(bucket_unpack_kernel, dim3(bucket_grid((n + 7) / 8)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)src, n, g);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}
