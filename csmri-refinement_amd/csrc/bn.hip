// BatchNorm2d (training) + LeakyReLU + Dropout2d mask on NHWC tensors: batch statistics,
// normalise+activate, and the two-pass backward.  HBM-bound: every kernel streams its
// tensors once with 8/16-byte channel vectors; per-channel constants live in registers
// (a thread owns a fixed group of 4 channels), reductions are deterministic two-stage
// (per-block partial rows -> fixed-order tree in the finalize kernels).
#include "common.h"

#define BN_MAX_ROWS 2048
// workgroups (= partial rows) for a [npix][C] tensor: about 16K elements each, at least one pixel per
// pixel lane of the 256-thread workgroup, so that wide-channel / few-pixel maps still fill the chip
extern "C" int csmri_bn_stats_rows(int npix, int C) {
  int ppb = 16384 / (C > 0 ? C : 1);
  const int lanes = C >= 4 ? 1024 / C : 256;
  if (ppb < lanes) ppb = lanes;
  if (ppb < 1) ppb = 1;
  int r = (npix + ppb - 1) / ppb;
  return r < 1 ? 1 : (r > BN_MAX_ROWS ? BN_MAX_ROWS : r);
}
static bool bn_channels_ok(int C) { return C >= 8 && C <= 1024 && (C & (C - 1)) == 0; }

template <int DT> __device__ __forceinline__ f32x4_t ld4(const void* p, long long idx) {
  f32x4_t r;
  if constexpr (DT == CSMRI_F32) {
    r = *(const f32x4_t*)((const float*)p + idx);
  } else {
    u32x2_t u = *(const u32x2_t*)((const unsigned short*)p + idx);
    r[0] = __uint_as_float(u[0] << 16); r[1] = __uint_as_float(u[0] & 0xffff0000u);
    r[2] = __uint_as_float(u[1] << 16); r[3] = __uint_as_float(u[1] & 0xffff0000u);
  }
  return r;
}
template <int DT> __device__ __forceinline__ void st4(void* p, long long idx, f32x4_t v) {
  store4(p, idx, DT, v);
}

// block-wide sum of two doubles (blockDim.x == 256); result valid in thread 0
__device__ __forceinline__ void block_sum2(double& a, double& b) {
  __shared__ double sa[4], sb[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sa[w] = a; sb[w] = b; }
  __syncthreads();
  if (threadIdx.x == 0) { a = sa[0] + sa[1] + sa[2] + sa[3]; b = sb[0] + sb[1] + sb[2] + sb[3]; }
}

// write one partial row: reduce the per-thread (a, b) over the pixel lanes of the block
__device__ __forceinline__ void write_partial_row(f32x4_t a, f32x4_t b, int nv, int lanes, int cv,
                                                  int pl, int C, float* partial, size_t R, size_t r) {
  __shared__ float red[2][256][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { red[0][threadIdx.x][q] = a[q]; red[1][threadIdx.x][q] = b[q]; }
  __syncthreads();
  if (pl == 0) {
    for (int l = 1; l < lanes; ++l)
#pragma unroll
      for (int q = 0; q < 4; ++q) { a[q] += red[0][l * nv + cv][q]; b[q] += red[1][l * nv + cv][q]; }
    // channel-major partials [2][C][R]: the finalize kernels read a channel's rows contiguously
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      partial[(size_t)(cv * 4 + q) * R + r] = a[q];
      partial[(size_t)(C + cv * 4 + q) * R + r] = b[q];
    }
  }
}

// ---- forward statistics (when the conv epilogue did not produce them) -------------
template <int DT>
__global__ __launch_bounds__(256) void bn_stats_kernel(const void* y, int ps, int npix, int C, int rows,
                                                       float* partial) {
  const int nv = C >> 2, lanes = 256 / nv, cv = threadIdx.x % nv, pl = threadIdx.x / nv;
  // npix, rows: per group; blockIdx.y = group (independent statistics per batch group)
  const int g = blockIdx.y, base = g * npix;
  const int chunk = (npix + rows - 1) / rows, p0 = blockIdx.x * chunk, p1 = min(npix, p0 + chunk);
  f32x4_t a = (f32x4_t){0, 0, 0, 0}, b = a;
#pragma unroll 4
  for (int p = p0 + pl; p < p1; p += lanes) {
    f32x4_t v = ld4<DT>(y, (long long)(base + p) * ps + cv * 4);
    a += v; b += v * v;
  }
  write_partial_row(a, b, nv, lanes, cv, pl, C, partial, (size_t)gridDim.y * rows, (size_t)g * rows + blockIdx.x);
}
extern "C" int csmri_bn_stats(int dtype, const void* y, int pix_stride, int npix, int C, float* partial,
                              int groups, void* stream) {
  CSMRI_CHECK_ARG(y && partial && npix > 0 && groups >= 1 && npix % groups == 0);
  if (!bn_channels_ok(C)) return CSMRI_E_UNSUPPORTED;
  const int npg = npix / groups, rows = csmri_bn_stats_rows(npg, C);
  if (dtype == CSMRI_BF16)
    hipLaunchKernelGGL(bn_stats_kernel<CSMRI_BF16>, dim3(rows, groups), dim3(256), 0, (hipStream_t)stream, y, pix_stride, npg, C, rows, partial);
  else
    hipLaunchKernelGGL(bn_stats_kernel<CSMRI_F32>, dim3(rows, groups), dim3(256), 0, (hipStream_t)stream, y, pix_stride, npg, C, rows, partial);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* partial, int rows, int C, int C_real,
                                                          double count, float eps, float momentum, float* mean,
                                                          float* invstd, float* rmean, float* rvar, int groups) {
  const int c = blockIdx.x;       // one block per channel, fixed-order tree over the rows
  // rows, count: per group.  Groups are finalized in order, so the running statistics see the
  // same sequence of momentum updates as `groups` separate forward calls would give.
  const size_t R = (size_t)rows * groups;             // partial: [2][C][R], group-major rows
  for (int g = 0; g < groups; ++g) {
    const float* p1 = partial + (size_t)c * R + (size_t)g * rows;
    const float* p2 = partial + ((size_t)C + c) * R + (size_t)g * rows;
    double s1 = 0, s2 = 0;
    for (int r = threadIdx.x; r < rows; r += 256) { s1 += p1[r]; s2 += p2[r]; }
    __syncthreads();
    block_sum2(s1, s2);
    if (threadIdx.x != 0) continue;
    const double m = s1 / count;
    double var = s2 / count - m * m;
    if (var < 0) var = 0;
    mean[g * C + c] = (float)m;
    invstd[g * C + c] = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean && c < C_real) {
      const double unbiased = count > 1 ? var * count / (count - 1) : var;
      rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)m;
      rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
    }
  }
}
extern "C" int csmri_bn_finalize(const float* partial, int rows, int C, int C_real, long long count, float eps,
                                 float momentum, float* mean, float* invstd, float* running_mean,
                                 float* running_var, int groups, void* stream) {
  CSMRI_CHECK_ARG(partial && mean && invstd && rows > 0 && count > 0 && groups >= 1 && rows % groups == 0 &&
                  count % groups == 0);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, partial, rows / groups, C,
                     C_real, (double)(count / groups), eps, momentum, mean, invstd, running_mean, running_var,
                     groups);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

// ---- z = drop[b,c] * lrelu((y-mean)*invstd*gamma + beta) ---------------------------
template <int DT>
__global__ __launch_bounds__(256) void bn_act_kernel(const void* __restrict__ y, int yps, void* __restrict__ z,
                                                     int zps, int B, int HW, int C, int C_real,
                                                     const float* __restrict__ mean,
                                                     const float* __restrict__ invstd,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float slope,
                                                     const float* __restrict__ drop, float* __restrict__ snap) {
  const int nv = C >> 2, lanes = 256 / nv, cv = threadIdx.x % nv, pl = threadIdx.x / nv, c = cv * 4;
  // B: images per group; blockIdx.y = group with its own mean/invstd
  const int grp = blockIdx.y, pbase = grp * B * HW;
  mean += grp * C; invstd += grp * C;
  float sc[4], mu[4], be[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const bool ok = c + q < C_real;
    sc[q] = ok ? invstd[c + q] * gamma[c + q] : 0.f;
    mu[q] = ok ? mean[c + q] : 0.f;
    be[q] = ok ? beta[c + q] : 0.f;
    if (snap && blockIdx.x == 0 && grp == 0 && pl == 0) {      // affine parameters as this forward saw them
      snap[c + q] = ok ? gamma[c + q] : 0.f;
      snap[C + c + q] = ok ? beta[c + q] : 0.f;
    }
  }
  const int npix = B * HW;
  // each workgroup streams one contiguous pixel range (same decomposition as the reductions)
  const int chunk = (npix + gridDim.x - 1) / gridDim.x, q0 = blockIdx.x * chunk, q1 = min(npix, q0 + chunk);
#pragma unroll 4
  for (int pp = q0 + pl; pp < q1; pp += lanes) {
    const int p = pbase + pp;
    f32x4_t v = ld4<DT>(y, (long long)p * yps + c), o;
    f32x4_t dm = (f32x4_t){1.f, 1.f, 1.f, 1.f};
    if (drop) dm = *(const f32x4_t*)(drop + (size_t)(p / HW) * C + c);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      // same operation order as (y-mean)*invstd*gamma+beta up to one rounding
      float t = v[q] - mu[q];
      t = t * sc[q] + be[q];
      t = t < 0.f ? t * slope : t;
      o[q] = t * dm[q];
    }
    st4<DT>(z, (long long)p * zps + c, o);
  }
}
extern "C" int csmri_bn_act(int dtype, const void* y, int y_pix_stride, void* z, int z_pix_stride, int B, int HW,
                            int C, int C_real, const float* mean, const float* invstd, const float* gamma,
                            const float* beta, float slope, const float* dropmask, float* affine_snap,
                            int groups, void* stream) {
  CSMRI_CHECK_ARG(y && z && mean && invstd && gamma && beta && groups >= 1 && B % groups == 0);
  if (!bn_channels_ok(C)) return CSMRI_E_UNSUPPORTED;
  const int lanes = 256 / (C / 4);
  B /= groups;                                   // images per group from here on
  const int blocks = csmri_bn_stats_rows(B * HW, C);
  (void)lanes;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CSMRI_BF16)
    hipLaunchKernelGGL(bn_act_kernel<CSMRI_BF16>, dim3(blocks, groups), dim3(256), 0, st, y, y_pix_stride, z, z_pix_stride, B, HW, C, C_real, mean, invstd, gamma, beta, slope, dropmask, affine_snap);
  else
    hipLaunchKernelGGL(bn_act_kernel<CSMRI_F32>, dim3(blocks, groups), dim3(256), 0, st, y, y_pix_stride, z, z_pix_stride, B, HW, C, C_real, mean, invstd, gamma, beta, slope, dropmask, affine_snap);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

// ---- backward pass 1: partials of dyh = dz*drop*lrelu'(z) and dyh*xhat -------------
// RECOMP: the sign of the activation is recomputed from y with the affine snapshot of the forward
// (exactly its arithmetic) instead of being read from z -- one tensor pass less
template <int DT, bool RECOMP>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const void* dz, int dzps, const void* y, int yps,
                                                            const void* z, int zps, int npix, int HW, int C,
                                                            const float* mean, const float* invstd, float slope,
                                                            const float* drop, int rows, float* partial,
                                                            const float* snap, const void* dz2, int dz2ps) {
  const int nv = C >> 2, lanes = 256 / nv, cv = threadIdx.x % nv, pl = threadIdx.x / nv, c = cv * 4;
  // npix, rows: per group; blockIdx.y = group
  const int grp = blockIdx.y, pbase = grp * npix;
  const f32x4_t mu = *(const f32x4_t*)(mean + grp * C + c), is = *(const f32x4_t*)(invstd + grp * C + c);
  f32x4_t fsc = is, fbe = is;
  if (RECOMP) { fsc = is * *(const f32x4_t*)(snap + c); fbe = *(const f32x4_t*)(snap + C + c); }
  const int chunk = (npix + rows - 1) / rows, p0 = blockIdx.x * chunk, p1 = min(npix, p0 + chunk);
  f32x4_t a = (f32x4_t){0, 0, 0, 0}, b = a;
#pragma unroll 4
  for (int pp = p0 + pl; pp < p1; pp += lanes) {
    const int p = pbase + pp;
    f32x4_t g = ld4<DT>(dz, (long long)p * dzps + c), yy = ld4<DT>(y, (long long)p * yps + c), zz;
    if (dz2) g += ld4<DT>(dz2, (long long)p * dz2ps + c);       // gradient fan-in of z: summed here in fp32
    if (RECOMP) zz = (yy - mu) * fsc + fbe; else zz = ld4<DT>(z, (long long)p * zps + c);
    f32x4_t dm = (f32x4_t){1.f, 1.f, 1.f, 1.f};
    if (drop) dm = *(const f32x4_t*)(drop + (size_t)(p / HW) * C + c);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float d = g[q] * (zz[q] > 0.f ? 1.f : slope) * dm[q];
      a[q] += d;
      b[q] += d * (yy[q] - mu[q]) * is[q];
    }
  }
  write_partial_row(a, b, nv, lanes, cv, pl, C, partial, (size_t)gridDim.y * rows + gridDim.y,
                    (size_t)grp * rows + blockIdx.x);
}
extern "C" int csmri_bn_bwd_reduce(int dtype, const void* dz, int dz_pix_stride, const void* y, int y_pix_stride,
                                   const void* z, int z_pix_stride, int B, int HW, int C, const float* mean,
                                   const float* invstd, float slope, const float* dropmask, float* partial,
                                   const float* affine_snap, int groups, const void* dz2, int dz2_pix_stride,
                                   void* stream) {
  CSMRI_CHECK_ARG(dz && y && partial && (z || affine_snap) && groups >= 1 && B % groups == 0);
  if (!bn_channels_ok(C)) return CSMRI_E_UNSUPPORTED;
  const int npix = B / groups * HW, rows = csmri_bn_stats_rows(npix, C);
  hipStream_t st = (hipStream_t)stream;
#define BN_RED(DT_, RC_) hipLaunchKernelGGL((bn_bwd_reduce_kernel<DT_, RC_>), dim3(rows, groups), dim3(256), 0, st, dz, dz_pix_stride, y, y_pix_stride, z, z_pix_stride, npix, HW, C, mean, invstd, slope, dropmask, rows, partial, affine_snap, dz2, dz2_pix_stride)
  if (dtype == CSMRI_BF16) { if (z) BN_RED(CSMRI_BF16, false); else BN_RED(CSMRI_BF16, true); }
  else { if (z) BN_RED(CSMRI_F32, false); else BN_RED(CSMRI_F32, true); }
#undef BN_RED
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(float* partial, int rows, int C, int C_real,
                                                              float* dgamma, float* dbeta, int accumulate,
                                                              int groups) {
  const int c = blockIdx.x;
  // rows: per group; the totals of group g go to row (groups*rows + g)
  float acc1 = 0.f, acc2 = 0.f;
  const size_t R = (size_t)rows * groups + groups;    // [2][C][R]: the last `groups` entries are the totals
  for (int g = 0; g < groups; ++g) {
    float* p1 = partial + (size_t)c * R;
    float* p2 = partial + ((size_t)C + c) * R;
    double s1 = 0, s2 = 0;
    for (int r = threadIdx.x; r < rows; r += 256) { s1 += p1[(size_t)g * rows + r]; s2 += p2[(size_t)g * rows + r]; }
    __syncthreads();
    block_sum2(s1, s2);
    if (threadIdx.x != 0) continue;
    p1[(size_t)groups * rows + g] = (float)s1;
    p2[(size_t)groups * rows + g] = (float)s2;
    acc1 += (float)s1; acc2 += (float)s2;
  }
  if (threadIdx.x == 0 && c < C_real) {
    if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + acc1;
    if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + acc2;
  }
}

// ---- backward pass 2: dy = gamma*invstd*(dyh - mean(dyh) - xhat*mean(dyh*xhat)) ----
template <int DT, bool RECOMP>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const void* __restrict__ dz, int dzps,
                                                           const void* __restrict__ y, int yps,
                                                           const void* __restrict__ z, int zps,
                                                           void* __restrict__ dy, int dyps, int npix, int HW, int C,
                                                           int C_real, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, float slope,
                                                           const float* __restrict__ drop,
                                                           const float* __restrict__ totals, float inv_count,
                                                           const float* __restrict__ snap, int totals_R,
                                                           const void* __restrict__ dz2, int dz2ps) {
  const int nv = C >> 2, lanes = 256 / nv, cv = threadIdx.x % nv, pl = threadIdx.x / nv, c = cv * 4;
  // npix: per group; blockIdx.y = group
  const int grp = blockIdx.y, pbase = grp * npix;
  const f32x4_t mu = *(const f32x4_t*)(mean + grp * C + c), is = *(const f32x4_t*)(invstd + grp * C + c);
  f32x4_t fsc = is, fbe = is;
  if (RECOMP) { fsc = is * *(const f32x4_t*)(snap + c); fbe = *(const f32x4_t*)(snap + C + c); }
  float gs[4], m1[4], m2[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    gs[q] = c + q < C_real ? gamma[c + q] * is[q] : 0.f;
    m1[q] = totals[(size_t)(c + q) * totals_R + grp] * inv_count;        // totals: &partial[rows_total], stride R
    m2[q] = totals[(size_t)(C + c + q) * totals_R + grp] * inv_count;
  }
  const int chunk = (npix + gridDim.x - 1) / gridDim.x, q0 = blockIdx.x * chunk, q1 = min(npix, q0 + chunk);
#pragma unroll 4
  for (int pp = q0 + pl; pp < q1; pp += lanes) {
    const int p = pbase + pp;
    f32x4_t g = ld4<DT>(dz, (long long)p * dzps + c), yy = ld4<DT>(y, (long long)p * yps + c), zz, o;
    if (dz2) g += ld4<DT>(dz2, (long long)p * dz2ps + c);
    if (RECOMP) zz = (yy - mu) * fsc + fbe; else zz = ld4<DT>(z, (long long)p * zps + c);
    f32x4_t dm = (f32x4_t){1.f, 1.f, 1.f, 1.f};
    if (drop) dm = *(const f32x4_t*)(drop + (size_t)(p / HW) * C + c);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float d = g[q] * (zz[q] > 0.f ? 1.f : slope) * dm[q];
      const float xh = (yy[q] - mu[q]) * is[q];
      o[q] = gs[q] * (d - m1[q] - xh * m2[q]);
    }
    st4<DT>(dy, (long long)p * dyps + c, o);
  }
}
// partial must hold (rows + groups) * 2 * C floats (the extra rows receive the per-group totals)
extern "C" int csmri_bn_bwd_apply(int dtype, const void* dz, int dz_pix_stride, const void* y, int y_pix_stride,
                                  const void* z, int z_pix_stride, void* dy, int dy_pix_stride, int B, int HW,
                                  int C, int C_real, const float* mean, const float* invstd, const float* gamma,
                                  float slope, const float* dropmask, const float* partial, int rows,
                                  float* dgamma, float* dbeta, int accumulate, const float* affine_snap,
                                  int groups, const void* dz2, int dz2_pix_stride, void* stream) {
  CSMRI_CHECK_ARG(dz && y && (z || affine_snap) && dy && partial && rows > 0 && groups >= 1 &&
                  rows % groups == 0 && B % groups == 0);
  if (!bn_channels_ok(C)) return CSMRI_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, st, (float*)partial, rows / groups, C, C_real,
                     dgamma, dbeta, accumulate, groups);
  CSMRI_LAUNCH_CHECK();
  B /= groups;                                   // images per group from here on
  const int npix = B * HW, lanes = 256 / (C / 4);
  const int blocks = csmri_bn_stats_rows(npix, C);
  (void)lanes;
  const float* totals = partial + rows;                // [2][C][rows + groups]: totals follow each channel's rows
  const int totals_R = rows + groups;
  const float inv = 1.0f / ((float)B * (float)HW);
#define BN_APP(DT_, RC_) hipLaunchKernelGGL((bn_bwd_apply_kernel<DT_, RC_>), dim3(blocks, groups), dim3(256), 0, st, dz, dz_pix_stride, y, y_pix_stride, z, z_pix_stride, dy, dy_pix_stride, npix, HW, C, C_real, mean, invstd, gamma, slope, dropmask, totals, inv, affine_snap, totals_R, dz2, dz2_pix_stride)
  if (dtype == CSMRI_BF16) { if (z) BN_APP(CSMRI_BF16, false); else BN_APP(CSMRI_BF16, true); }
  else { if (z) BN_APP(CSMRI_F32, false); else BN_APP(CSMRI_F32, true); }
#undef BN_APP
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}
