// BatchNorm2d (training) + LeakyReLU + Dropout2d mask on NHWC tensors: batch statistics,
// normalise+activate, and the two-pass backward.  HBM-bound: every kernel streams its
// tensors once with 8/16-byte channel vectors; per-channel constants live in registers
// (a thread owns a fixed group of 4 channels), reductions are deterministic two-stage
// (per-block partial rows -> fixed-order tree in the finalize kernels).
#include "common.h"

#define BN_MAX_ROWS 2048
#ifndef BN_EPB
#define BN_EPB 16384      // elements per workgroup (same-box A/B of the step: 8192 / 32768, profiles/r04_same_box_ab.json)
#endif
// workgroups (= partial rows) for a [npix][C] tensor: about 16K elements each, at least one pixel per
// pixel lane of the 256-thread workgroup, so that wide-channel / few-pixel maps still fill the chip
extern "C" int csmri_bn_stats_rows(int npix, int C) {
  int ppb = BN_EPB / (C > 0 ? C : 1);
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

// ---- how the pixel loops are written --------------------------------------------------------------------------------
// Every kernel streams a pixel range with `lanes` pixel lanes per workgroup.  The loops run in explicit batches of BU
// pixels per thread: all raw loads of a batch are issued (indices clamped to the range's last pixel, no predication),
// a scheduling barrier, then conversions and arithmetic.  No branch anywhere in a batch (the compiler sinks loads into
// the branch that uses them -- back to one round trip per pixel): tail pixels count with a 0 factor in the reductions
// and are stored AGAIN, same value, in the passes that write (which is why the output must not alias an input).  Rolled loops with optional loads under run-time branches compiled to one or two
// pixels in flight per thread (ISA: load, load, s_waitcnt vmcnt(0), use): 2-3.5 TB/s on the large maps and one L2 / HBM
// round trip per pixel on the small ones.  Optional operands (second gradient, dropout mask) are template parameters.
// BU: same-box A/B of the bench step (tools/ab_old_new.sh, slices/s resident): 1, 2, 3, 4 -> +0.8 % over the rolled loops,
// 8 -> -1 % (148-176 VGPRs: the co-running convolution kernels of the other graph branches lose occupancy)
#ifndef BU
#define BU 4
#endif
// four consecutive per-channel parameters of an array that holds C_real entries (zeros past the end): one 16-byte load
// for every channel vector but a ragged last one (predicated scalar loads under `c + q < C_real` compile to one
// serial round trip per channel at the head of every workgroup)
__device__ __forceinline__ f32x4_t ld4_real(const float* __restrict__ p, int c, int C_real) {
  // (a workgroup-uniform condition, i.e. a scalar branch: a per-lane one is if-converted and BOTH paths execute)
  if ((C_real & 3) == 0 && (((uintptr_t)p) & 15) == 0) return c < C_real ? *(const f32x4_t*)(p + c) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
  f32x4_t r;
#pragma unroll
  for (int q = 0; q < 4; ++q) r[q] = c + q < C_real ? p[c + q] : 0.f;
  return r;
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
  for (int q0 = p0 + pl; q0 < p1; q0 += lanes * BU) {
    typename raw4<DT>::t r[BU];
#pragma unroll
    for (int u = 0; u < BU; ++u) r[u] = ldraw<DT>(y, (long long)(base + min(q0 + u * lanes, p1 - 1)) * ps + cv * 4);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < BU; ++u) {
      const float m = q0 + u * lanes < p1 ? 1.f : 0.f;
      const f32x4_t v = cvt4<DT>(r[u]) * m;
      a += v; b += v * v;
    }
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
template <int DT, bool DROP>
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
  // (mean / invstd hold C entries, gamma / beta C_real)
  const f32x4_t ga = ld4_real(gamma, c, C_real), bt = ld4_real(beta, c, C_real);
  const f32x4_t mn = *(const f32x4_t*)(mean + c), iv = *(const f32x4_t*)(invstd + c);
  float sc[4], mu[4], be[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const bool ok = c + q < C_real;
    sc[q] = ok ? iv[q] * ga[q] : 0.f;
    mu[q] = ok ? mn[q] : 0.f;
    be[q] = bt[q];
  }
  if (snap && blockIdx.x == 0 && grp == 0 && pl == 0) {        // affine parameters as this forward saw them
    *(f32x4_t*)(snap + c) = ga;
    *(f32x4_t*)(snap + C + c) = bt;
  }
  const int npix = B * HW;
  // each workgroup streams one contiguous pixel range (same decomposition as the reductions)
  const int chunk = (npix + gridDim.x - 1) / gridDim.x, q0 = blockIdx.x * chunk, q1 = min(npix, q0 + chunk);
#ifdef BN_ROLLED
#pragma unroll 4
  for (int pp = q0 + pl; pp < q1; pp += lanes) {
    const int p = pbase + pp;
    f32x4_t v = ld4<DT>(y, (long long)p * yps + c), o;
    f32x4_t dm = (f32x4_t){1.f, 1.f, 1.f, 1.f};
    if constexpr (DROP) dm = *(const f32x4_t*)(drop + (size_t)(p / HW) * C + c);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float t = v[q] - mu[q];
      t = t * sc[q] + be[q];
      t = t < 0.f ? t * slope : t;
      o[q] = t * dm[q];
    }
    st4<DT>(z, (long long)p * zps + c, o);
  }
#else
  for (int b0 = q0 + pl; b0 < q1; b0 += lanes * BU) {
    typename raw4<DT>::t r[BU];
    f32x4_t dm[BU];
#pragma unroll
    for (int u = 0; u < BU; ++u) {
      const int p = pbase + min(b0 + u * lanes, q1 - 1);
      r[u] = ldraw<DT>(y, (long long)p * yps + c);
      if constexpr (DROP) dm[u] = *(const f32x4_t*)(drop + (size_t)(p / HW) * C + c);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < BU; ++u) {
      const int p = pbase + min(b0 + u * lanes, q1 - 1);     // past the end: the range's last pixel again
      const f32x4_t v = cvt4<DT>(r[u]);
      f32x4_t o;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // same operation order as (y-mean)*invstd*gamma+beta up to one rounding
        float t = v[q] - mu[q];
        t = t * sc[q] + be[q];
        t = t < 0.f ? t * slope : t;
        o[q] = DROP ? t * dm[u][q] : t;
      }
      st4<DT>(z, (long long)p * zps + c, o);
    }
  }
#endif
}
extern "C" int csmri_bn_act(int dtype, const void* y, int y_pix_stride, void* z, int z_pix_stride, int B, int HW,
                            int C, int C_real, const float* mean, const float* invstd, const float* gamma,
                            const float* beta, float slope, const float* dropmask, float* affine_snap,
                            int groups, void* stream) {
  CSMRI_CHECK_ARG(y && z && z != y && mean && invstd && gamma && beta && groups >= 1 && B % groups == 0);
  if (!bn_channels_ok(C)) return CSMRI_E_UNSUPPORTED;
  const int lanes = 256 / (C / 4);
  B /= groups;                                   // images per group from here on
  const int blocks = csmri_bn_stats_rows(B * HW, C);
  (void)lanes;
  hipStream_t st = (hipStream_t)stream;
#define BN_ACT2(DT_, DR_) hipLaunchKernelGGL((bn_act_kernel<DT_, DR_>), dim3(blocks, groups), dim3(256), 0, st, y, y_pix_stride, z, z_pix_stride, B, HW, C, C_real, mean, invstd, gamma, beta, slope, dropmask, affine_snap)
#define BN_ACT(DT_) do { if (dropmask) BN_ACT2(DT_, true); else BN_ACT2(DT_, false); } while (0)
  if (dtype == CSMRI_BF16) BN_ACT(CSMRI_BF16); else BN_ACT(CSMRI_F32);
#undef BN_ACT
#undef BN_ACT2
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

// ---- backward pass 1: partials of dyh = dz*drop*lrelu'(z) and dyh*xhat -------------
// RECOMP: the sign of the activation is recomputed from y with the affine snapshot of the forward
// (exactly its arithmetic) instead of being read from z -- one tensor pass less
// (HAS2 is a template parameter: a run-time `if (dz2)` around the extra load inside the pixel loop cost the loop its
// load pipelining -- 12.6 -> 18.2 us per launch on the bench step)
template <int DT, bool RECOMP, bool HAS2, bool DROP>
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
#ifdef BN_ROLLED
#pragma unroll 4
  for (int pp = p0 + pl; pp < p1; pp += lanes) {
    const int p = pbase + pp;
    f32x4_t g = ld4<DT>(dz, (long long)p * dzps + c), yy = ld4<DT>(y, (long long)p * yps + c), zz;
    if constexpr (HAS2) g += ld4<DT>(dz2, (long long)p * dz2ps + c);
    if (RECOMP) zz = (yy - mu) * fsc + fbe; else zz = ld4<DT>(z, (long long)p * zps + c);
    f32x4_t dm = (f32x4_t){1.f, 1.f, 1.f, 1.f};
    if constexpr (DROP) dm = *(const f32x4_t*)(drop + (size_t)(p / HW) * C + c);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float d = g[q] * (zz[q] > 0.f ? 1.f : slope) * dm[q];
      a[q] += d;
      b[q] += d * (yy[q] - mu[q]) * is[q];
    }
  }
#else
  for (int b0 = p0 + pl; b0 < p1; b0 += lanes * BU) {
    typename raw4<DT>::t rg[BU], ry[BU], rz[BU], rg2[BU];
    f32x4_t dm[BU];
#pragma unroll
    for (int u = 0; u < BU; ++u) {
      const int p = pbase + min(b0 + u * lanes, p1 - 1);
      rg[u] = ldraw<DT>(dz, (long long)p * dzps + c);
      ry[u] = ldraw<DT>(y, (long long)p * yps + c);
      if constexpr (!RECOMP) rz[u] = ldraw<DT>(z, (long long)p * zps + c);
      if constexpr (HAS2) rg2[u] = ldraw<DT>(dz2, (long long)p * dz2ps + c);     // gradient fan-in of z: summed here in fp32
      if constexpr (DROP) dm[u] = *(const f32x4_t*)(drop + (size_t)(p / HW) * C + c);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < BU; ++u) {
      const float m = b0 + u * lanes < p1 ? 1.f : 0.f;
      f32x4_t g = cvt4<DT>(rg[u]), zz;
      const f32x4_t yy = cvt4<DT>(ry[u]);
      if constexpr (HAS2) g += cvt4<DT>(rg2[u]);
      if constexpr (RECOMP) zz = (yy - mu) * fsc + fbe; else zz = cvt4<DT>(rz[u]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float d = g[q] * (zz[q] > 0.f ? m : slope * m);
        if constexpr (DROP) d *= dm[u][q];
        a[q] += d;
        b[q] += d * (yy[q] - mu[q]) * is[q];
      }
    }
  }
#endif
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
#define BN_RED2(DT_, RC_, H2_) do { if (dropmask) BN_RED3(DT_, RC_, H2_, true); else BN_RED3(DT_, RC_, H2_, false); } while (0)
#define BN_RED3(DT_, RC_, H2_, DR_) hipLaunchKernelGGL((bn_bwd_reduce_kernel<DT_, RC_, H2_, DR_>), dim3(rows, groups), dim3(256), 0, st, dz, dz_pix_stride, y, y_pix_stride, z, z_pix_stride, npix, HW, C, mean, invstd, slope, dropmask, rows, partial, affine_snap, dz2, dz2_pix_stride)
#define BN_RED(DT_, RC_) do { if (dz2) BN_RED2(DT_, RC_, true); else BN_RED2(DT_, RC_, false); } while (0)
  if (dtype == CSMRI_BF16) { if (z) BN_RED(CSMRI_BF16, false); else BN_RED(CSMRI_BF16, true); }
  else { if (z) BN_RED(CSMRI_F32, false); else BN_RED(CSMRI_F32, true); }
#undef BN_RED
#undef BN_RED2
#undef BN_RED3
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
template <int DT, bool RECOMP, bool HAS2, bool DROP>
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
  const f32x4_t ga = ld4_real(gamma, c, C_real);
  float gs[4], m1[4], m2[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    gs[q] = ga[q] * is[q];
    m1[q] = totals[(size_t)(c + q) * totals_R + grp] * inv_count;        // totals: &partial[rows_total], stride R
    m2[q] = totals[(size_t)(C + c + q) * totals_R + grp] * inv_count;
  }
  const int chunk = (npix + gridDim.x - 1) / gridDim.x, q0 = blockIdx.x * chunk, q1 = min(npix, q0 + chunk);
#ifdef BN_ROLLED
#pragma unroll 4
  for (int pp = q0 + pl; pp < q1; pp += lanes) {
    const int p = pbase + pp;
    f32x4_t g = ld4<DT>(dz, (long long)p * dzps + c), yy = ld4<DT>(y, (long long)p * yps + c), zz, o;
    if constexpr (HAS2) g += ld4<DT>(dz2, (long long)p * dz2ps + c);
    if (RECOMP) zz = (yy - mu) * fsc + fbe; else zz = ld4<DT>(z, (long long)p * zps + c);
    f32x4_t dm = (f32x4_t){1.f, 1.f, 1.f, 1.f};
    if constexpr (DROP) dm = *(const f32x4_t*)(drop + (size_t)(p / HW) * C + c);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float d = g[q] * (zz[q] > 0.f ? 1.f : slope) * dm[q];
      const float xh = (yy[q] - mu[q]) * is[q];
      o[q] = gs[q] * (d - m1[q] - xh * m2[q]);
    }
    st4<DT>(dy, (long long)p * dyps + c, o);
  }
#else
  for (int b0 = q0 + pl; b0 < q1; b0 += lanes * BU) {
    typename raw4<DT>::t rg[BU], ry[BU], rz[BU], rg2[BU];
    f32x4_t dm[BU];
#pragma unroll
    for (int u = 0; u < BU; ++u) {
      const int p = pbase + min(b0 + u * lanes, q1 - 1);
      rg[u] = ldraw<DT>(dz, (long long)p * dzps + c);
      ry[u] = ldraw<DT>(y, (long long)p * yps + c);
      if constexpr (!RECOMP) rz[u] = ldraw<DT>(z, (long long)p * zps + c);
      if constexpr (HAS2) rg2[u] = ldraw<DT>(dz2, (long long)p * dz2ps + c);
      if constexpr (DROP) dm[u] = *(const f32x4_t*)(drop + (size_t)(p / HW) * C + c);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < BU; ++u) {
      const int p = pbase + min(b0 + u * lanes, q1 - 1);
      f32x4_t g = cvt4<DT>(rg[u]), zz, o;
      const f32x4_t yy = cvt4<DT>(ry[u]);
      if constexpr (HAS2) g += cvt4<DT>(rg2[u]);
      if constexpr (RECOMP) zz = (yy - mu) * fsc + fbe; else zz = cvt4<DT>(rz[u]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float d = g[q] * (zz[q] > 0.f ? 1.f : slope);
        if constexpr (DROP) d *= dm[u][q];
        const float xh = (yy[q] - mu[q]) * is[q];
        o[q] = gs[q] * (d - m1[q] - xh * m2[q]);
      }
      st4<DT>(dy, (long long)p * dyps + c, o);
    }
  }
#endif
}
// partial must hold (rows + groups) * 2 * C floats (the extra rows receive the per-group totals)
extern "C" int csmri_bn_bwd_apply(int dtype, const void* dz, int dz_pix_stride, const void* y, int y_pix_stride,
                                  const void* z, int z_pix_stride, void* dy, int dy_pix_stride, int B, int HW,
                                  int C, int C_real, const float* mean, const float* invstd, const float* gamma,
                                  float slope, const float* dropmask, const float* partial, int rows,
                                  float* dgamma, float* dbeta, int accumulate, const float* affine_snap,
                                  int groups, const void* dz2, int dz2_pix_stride, void* stream) {
  CSMRI_CHECK_ARG(dz && y && (z || affine_snap) && dy && dy != dz && dy != y && dy != z && dy != dz2 && partial && rows > 0 && groups >= 1 &&
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
#define BN_APP(DT_, RC_) do { if (dz2) BN_APP2(DT_, RC_, true); else BN_APP2(DT_, RC_, false); } while (0)
#define BN_APP2(DT_, RC_, H2_) do { if (dropmask) BN_APP3(DT_, RC_, H2_, true); else BN_APP3(DT_, RC_, H2_, false); } while (0)
#define BN_APP3(DT_, RC_, H2_, DR_) hipLaunchKernelGGL((bn_bwd_apply_kernel<DT_, RC_, H2_, DR_>), dim3(blocks, groups), dim3(256), 0, st, dz, dz_pix_stride, y, y_pix_stride, z, z_pix_stride, dy, dy_pix_stride, npix, HW, C, C_real, mean, invstd, gamma, slope, dropmask, totals, inv, affine_snap, totals_R, dz2, dz2_pix_stride)
  if (dtype == CSMRI_BF16) { if (z) BN_APP(CSMRI_BF16, false); else BN_APP(CSMRI_BF16, true); }
  else { if (z) BN_APP(CSMRI_F32, false); else BN_APP(CSMRI_F32, true); }
#undef BN_APP
#undef BN_APP2
#undef BN_APP3
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

// ---------------------------------------------------------------------------------------------
// Small feature maps (<= 512 pixels: the discriminator's 8 x 8 x 1024 layers on the generator step's 8-image pass): the
// three-launch sequences above (statistics, finalize, normalise / reduce, finalize, apply) are three launch floors and
// six dependent round trips -- 18-19 us for a 1 MB tensor even with batched loads; one launch takes 12.  (At 2048
// pixels x 512 channels the one-launch form LOSES, 34 vs 19 us: 64 workgroups of 16-byte pixel segments.)  Here ONE launch does the whole pass: a
// workgroup owns 16 channels (a 32-byte bf16 segment of every pixel) and ALL pixels, so the per-channel sums need no
// second stage: pass 1 accumulates them (float per thread over <= 64 pixels, double across the 64 pixel lanes, fixed
// order), pass 2 re-reads the (L2-resident) tensor and writes the result.  One group only (the U-Net's calls).
// ---------------------------------------------------------------------------------------------
#define BN_SMALL_MAX_PIX 512
extern "C" int csmri_bn_small_ok(int npix, int C, int groups) {
  return groups == 1 && bn_channels_ok(C) && C >= 16 && npix >= 1 && npix <= BN_SMALL_MAX_PIX;
}

// sums of a[q], b[q] over the 256 / CVB pixel lanes of the workgroup (thread = CVB (channel vector) x pixel lane):
// every thread returns with the totals of its own channel vector
template <int CVB>
__device__ __forceinline__ void small_totals(const f32x4_t a, const f32x4_t b, double (&ta)[4], double (&tb)[4]) {
  __shared__ double sh[4][CVB][8];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, cv = threadIdx.x & (CVB - 1);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    double x = a[q], y = b[q];
#pragma unroll
    for (int o = CVB; o < 64; o <<= 1) { x += __shfl_xor(x, o); y += __shfl_xor(y, o); }
    if (lane < CVB) { sh[w][cv][q] = x; sh[w][cv][4 + q] = y; }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    ta[q] = ((sh[0][cv][q] + sh[1][cv][q]) + sh[2][cv][q]) + sh[3][cv][q];
    tb[q] = ((sh[0][cv][4 + q] + sh[1][cv][4 + q]) + sh[2][cv][4 + q]) + sh[3][cv][4 + q];
  }
}

template <int DT, int CVB, bool DROP>
__global__ __launch_bounds__(256) void bn_small_fwd_kernel(const void* __restrict__ y, int yps, void* __restrict__ z,
                                                           int zps, int npix, int HW, int C, int C_real,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float slope,
                                                           const float* __restrict__ drop, float eps, float momentum,
                                                           float* __restrict__ mean, float* __restrict__ invstd,
                                                           float* rmean, float* rvar, float* __restrict__ snap) {
  constexpr int PL = 256 / CVB, U = 8;          // pixel lanes; pixels per thread and batch
  const int hw_shift = (HW & (HW - 1)) ? -1 : 31 - __builtin_clz(HW);
  const int cv = threadIdx.x & (CVB - 1), pl = threadIdx.x / CVB, c = (blockIdx.x * CVB + cv) * 4;
  const f32x4_t ga = ld4_real(gamma, c, C_real), bt = ld4_real(beta, c, C_real);   // (in flight during pass 1)
  f32x4_t a = (f32x4_t){0, 0, 0, 0}, b = a;
  for (int p0 = pl; p0 < npix; p0 += PL * U) {
    f32x4_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld4<DT>(y, (long long)min(p0 + u * PL, npix - 1) * yps + c);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float m = p0 + u * PL < npix ? 1.f : 0.f;       // (a mask, not a branch: a branch makes the compiler sink
      a += v[u] * m; b += v[u] * v[u] * m;                   //  each load down to its use, one round trip per pixel)
    }
  }
  double ta[4], tb[4];
  small_totals<CVB>(a, b, ta, tb);
  float sc[4], mu[4], be[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const double count = (double)npix;
    const double m = ta[q] / count;
    double var = tb[q] / count - m * m;
    if (var < 0) var = 0;
    const float mf = (float)m, is = (float)(1.0 / sqrt(var + (double)eps));
    const bool ok = c + q < C_real;
    if (pl == 0) {
      mean[c + q] = mf; invstd[c + q] = is;
      if (rmean && ok) {
        const double unbiased = count > 1 ? var * count / (count - 1) : var;
        rmean[c + q] = (1.f - momentum) * rmean[c + q] + momentum * mf;
        rvar[c + q] = (1.f - momentum) * rvar[c + q] + momentum * (float)unbiased;
      }
      if (snap) { snap[c + q] = ga[q]; snap[C + c + q] = bt[q]; }
    }
    sc[q] = ok ? is * ga[q] : 0.f;
    mu[q] = ok ? mf : 0.f;
    be[q] = bt[q];
  }
  for (int p0 = pl; p0 < npix; p0 += PL * U) {
    f32x4_t v[U], dm[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = min(p0 + u * PL, npix - 1);
      v[u] = ld4<DT>(y, (long long)p * yps + c);
      if constexpr (DROP) dm[u] = *(const f32x4_t*)(drop + (size_t)(hw_shift >= 0 ? p >> hw_shift : p / HW) * C + c);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = min(p0 + u * PL, npix - 1);     // past the end: pixel npix - 1 again
      f32x4_t o;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float t = v[u][q] - mu[q];                  // the arithmetic of bn_act_kernel
        t = t * sc[q] + be[q];
        t = t < 0.f ? t * slope : t;
        o[q] = DROP ? t * dm[u][q] : t;
      }
      st4<DT>(z, (long long)p * zps + c, o);
    }
  }
}
extern "C" int csmri_bn_small_fwd(int dtype, const void* y, int y_pix_stride, void* z, int z_pix_stride, int B, int HW,
                                  int C, int C_real, const float* gamma, const float* beta, float slope,
                                  const float* dropmask, float eps, float momentum, float* mean, float* invstd,
                                  float* running_mean, float* running_var, float* affine_snap, void* stream) {
  CSMRI_CHECK_ARG(y && z && z != y && gamma && beta && mean && invstd && (!running_mean == !running_var));
  if (!csmri_bn_small_ok(B * HW, C, 1)) return CSMRI_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  // 16 channels per workgroup for the tiny maps, 8 (twice the workgroups, half the serial pixel loop) from 512 pixels on
#define BN_SF3(DT_, CVB_, DR_) hipLaunchKernelGGL((bn_small_fwd_kernel<DT_, CVB_, DR_>), dim3(C / (4 * CVB_)), dim3(256), 0, st, y, y_pix_stride, z, z_pix_stride, B * HW, HW, C, C_real, gamma, beta, slope, dropmask, eps, momentum, mean, invstd, running_mean, running_var, affine_snap)
#define BN_SF2(DT_, CVB_) do { if (dropmask) BN_SF3(DT_, CVB_, true); else BN_SF3(DT_, CVB_, false); } while (0)
#define BN_SF(DT_) do { if (B * HW >= 512) BN_SF2(DT_, 2); else BN_SF2(DT_, 4); } while (0)
  if (dtype == CSMRI_BF16) BN_SF(CSMRI_BF16); else BN_SF(CSMRI_F32);
#undef BN_SF
#undef BN_SF2
#undef BN_SF3
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

template <int DT, int CVB, bool HAS2, bool DROP>
__global__ __launch_bounds__(256) void bn_small_bwd_kernel(const void* __restrict__ dz, int dzps,
                                                           const void* __restrict__ dz2, int dz2ps,
                                                           const void* __restrict__ y, int yps, void* __restrict__ dy,
                                                           int dyps, int npix, int HW, int C, int C_real,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, float slope,
                                                           const float* __restrict__ drop,
                                                           const float* __restrict__ snap, float* dgamma, float* dbeta,
                                                           int accumulate) {
  constexpr int PL = 256 / CVB, U = 8;
  const int hw_shift = (HW & (HW - 1)) ? -1 : 31 - __builtin_clz(HW);
  const int cv = threadIdx.x & (CVB - 1), pl = threadIdx.x / CVB, c = (blockIdx.x * CVB + cv) * 4;
  const f32x4_t mu = *(const f32x4_t*)(mean + c), is = *(const f32x4_t*)(invstd + c);
  const f32x4_t fsc = is * *(const f32x4_t*)(snap + c), fbe = *(const f32x4_t*)(snap + C + c);
  const f32x4_t ga = ld4_real(gamma, c, C_real);
  f32x4_t a = (f32x4_t){0, 0, 0, 0}, b = a;
  for (int p0 = pl; p0 < npix; p0 += PL * U) {
    typename raw4<DT>::t rg[U], ry[U], rg2[U];
    f32x4_t g[U], yy[U], dm[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = min(p0 + u * PL, npix - 1);
      rg[u] = ldraw<DT>(dz, (long long)p * dzps + c);
      ry[u] = ldraw<DT>(y, (long long)p * yps + c);
      if constexpr (HAS2) rg2[u] = ldraw<DT>(dz2, (long long)p * dz2ps + c);
      if constexpr (DROP) dm[u] = *(const f32x4_t*)(drop + (size_t)(hw_shift >= 0 ? p >> hw_shift : p / HW) * C + c);
    }
    __builtin_amdgcn_sched_barrier(0);        // every load of the batch is issued before the first conversion
#pragma unroll
    for (int u = 0; u < U; ++u) {
      g[u] = cvt4<DT>(rg[u]); yy[u] = cvt4<DT>(ry[u]);
      if constexpr (HAS2) g[u] += cvt4<DT>(rg2[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float m = p0 + u * PL < npix ? 1.f : 0.f;
      const f32x4_t zz = (yy[u] - mu) * fsc + fbe;            // the activation's sign, recomputed as the forward computed it
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float d = g[u][q] * (zz[q] > 0.f ? m : slope * m);
        if constexpr (DROP) d *= dm[u][q];
        a[q] += d;
        b[q] += d * (yy[u][q] - mu[q]) * is[q];
      }
    }
  }
  double ta[4], tb[4];
  small_totals<CVB>(a, b, ta, tb);
  float gs[4], m1[4], m2[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const bool ok = c + q < C_real;
    if (pl == 0 && ok) {
      if (dbeta) dbeta[c + q] = (accumulate ? dbeta[c + q] : 0.f) + (float)ta[q];
      if (dgamma) dgamma[c + q] = (accumulate ? dgamma[c + q] : 0.f) + (float)tb[q];
    }
    gs[q] = ga[q] * is[q];
    const float inv = 1.0f / (float)npix;
    m1[q] = (float)ta[q] * inv;
    m2[q] = (float)tb[q] * inv;
  }
  for (int p0 = pl; p0 < npix; p0 += PL * U) {
    typename raw4<DT>::t rg[U], ry[U], rg2[U];
    f32x4_t g[U], yy[U], dm[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = min(p0 + u * PL, npix - 1);
      rg[u] = ldraw<DT>(dz, (long long)p * dzps + c);
      ry[u] = ldraw<DT>(y, (long long)p * yps + c);
      if constexpr (HAS2) rg2[u] = ldraw<DT>(dz2, (long long)p * dz2ps + c);
      if constexpr (DROP) dm[u] = *(const f32x4_t*)(drop + (size_t)(hw_shift >= 0 ? p >> hw_shift : p / HW) * C + c);
    }
    __builtin_amdgcn_sched_barrier(0);        // every load of the batch is issued before the first conversion
#pragma unroll
    for (int u = 0; u < U; ++u) {
      g[u] = cvt4<DT>(rg[u]); yy[u] = cvt4<DT>(ry[u]);
      if constexpr (HAS2) g[u] += cvt4<DT>(rg2[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = min(p0 + u * PL, npix - 1);
      const f32x4_t zz = (yy[u] - mu) * fsc + fbe;
      f32x4_t o;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float d = g[u][q] * (zz[q] > 0.f ? 1.f : slope);
        if constexpr (DROP) d *= dm[u][q];
        const float xh = (yy[u][q] - mu[q]) * is[q];
        o[q] = gs[q] * (d - m1[q] - xh * m2[q]);
      }
      st4<DT>(dy, (long long)p * dyps + c, o);
    }
  }
}
extern "C" int csmri_bn_small_bwd(int dtype, const void* dz, int dz_pix_stride, const void* dz2, int dz2_pix_stride,
                                  const void* y, int y_pix_stride, void* dy, int dy_pix_stride, int B, int HW, int C,
                                  int C_real, const float* mean, const float* invstd, const float* gamma, float slope,
                                  const float* dropmask, const float* affine_snap, float* dgamma, float* dbeta,
                                  int accumulate, void* stream) {
  CSMRI_CHECK_ARG(dz && y && dy && dy != dz && dy != y && dy != dz2 && mean && invstd && gamma && affine_snap);
  if (!csmri_bn_small_ok(B * HW, C, 1)) return CSMRI_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
#define BN_SB4(DT_, CVB_, H2_, DR_) hipLaunchKernelGGL((bn_small_bwd_kernel<DT_, CVB_, H2_, DR_>), dim3(C / (4 * CVB_)), dim3(256), 0, st, dz, dz_pix_stride, dz2, dz2_pix_stride, y, y_pix_stride, dy, dy_pix_stride, B * HW, HW, C, C_real, mean, invstd, gamma, slope, dropmask, affine_snap, dgamma, dbeta, accumulate)
#define BN_SB3(DT_, CVB_, H2_) do { if (dropmask) BN_SB4(DT_, CVB_, H2_, true); else BN_SB4(DT_, CVB_, H2_, false); } while (0)
#define BN_SB2(DT_, CVB_) do { if (dz2) BN_SB3(DT_, CVB_, true); else BN_SB3(DT_, CVB_, false); } while (0)
#define BN_SB(DT_) do { if (B * HW >= 512) BN_SB2(DT_, 2); else BN_SB2(DT_, 4); } while (0)
  if (dtype == CSMRI_BF16) BN_SB(CSMRI_BF16); else BN_SB(CSMRI_F32);
#undef BN_SB
#undef BN_SB2
#undef BN_SB3
#undef BN_SB4
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}
