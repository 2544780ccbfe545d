// Small fused ops around the networks: complex magnitude, per-sample min/max
// scaling of the refinement wrapper, loss reductions (deterministic two-stage),
// BCE on logits, PSNR, Adam.  All HBM- or latency-bound.
#include "common.h"

static inline int grid_for(long long work, int threads = 256) {
  long long b = (work + threads - 1) / threads;
  if (b < 1) b = 1;
  if (b > 1024) b = 1024;
  return (int)b;
}
#define GRID_STRIDE(i, n) \
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// block-wide sum of a double (256 threads); result valid in thread 0
__device__ double block_sum(double v) {
  __shared__ double sh[256];
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  double r = sh[0];
  __syncthreads();
  return r;
}

// ------------------------------------------------------------ complex_abs ----
__constant__ float c_vgg_mean[3] = {0.485f, 0.456f, 0.406f};
__constant__ float c_vgg_std[3] = {0.229f, 0.224f, 0.225f};

__global__ void complex_abs_kernel(const float2* x, long long npix, void* out, int dt, int ps, int Cpad,
                                   int mode) {
  GRID_STRIDE(p, npix) {
    const float2 v = x[p];
    const float a = sqrtf(v.x * v.x + v.y * v.y);
    if (Cpad == 8 && (ps & 3) == 0) {          // the channel-padded conv input: the pixel's 8 channels as two 4-wide stores
      f32x4_t lo = (f32x4_t){a, 0.f, 0.f, 0.f};
      if (mode != 0) lo = (f32x4_t){(a - c_vgg_mean[0]) / c_vgg_std[0], (a - c_vgg_mean[1]) / c_vgg_std[1],
                                    (a - c_vgg_mean[2]) / c_vgg_std[2], 0.f};
      store4(out, p * ps, dt, lo);
      store4(out, p * ps + 4, dt, (f32x4_t){0.f, 0.f, 0.f, 0.f});
      continue;
    }
    for (int c = 0; c < Cpad; ++c) {
      float o = 0.f;
      if (mode == 0) o = c == 0 ? a : 0.f;
      else if (c < 3) o = (a - c_vgg_mean[c]) / c_vgg_std[c];
      store_elem(out, p * ps + c, dt, o);
    }
  }
}
extern "C" int csmri_complex_abs(const float* x, long long npix, void* out, int out_dtype,
                                 int out_pix_stride, int Cpad, int mode, void* stream) {
  CSMRI_CHECK_ARG(x && out && (mode == 0 || mode == 3) && Cpad >= (mode == 3 ? 3 : 1));
  hipLaunchKernelGGL(complex_abs_kernel, dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)stream,
                     (const float2*)x, npix, out, out_dtype, out_pix_stride, Cpad, mode);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}
// d|x|/dx = x/|x| (NaN at exactly 0, as in the reference's **0.5)
__global__ void complex_abs_bwd_kernel(const float2* x, long long npix, const void* g, int gdt, int gps,
                                       int nch, int mode, float2* dx, int accumulate) {
  GRID_STRIDE(p, npix) {
    const float2 v = x[p];
    const float a = sqrtf(v.x * v.x + v.y * v.y);
    float s = 0.f;
    for (int c = 0; c < nch; ++c) {
      float gv = load_elem(g, p * gps + c, gdt);
      s += mode == 3 ? gv / c_vgg_std[c] : gv;
    }
    float2 r = make_float2(s * v.x / a, s * v.y / a);
    if (accumulate) { float2 o = dx[p]; r.x += o.x; r.y += o.y; }
    dx[p] = r;
  }
}
extern "C" int csmri_complex_abs_bwd(const float* x, long long npix, const void* g, int g_dtype,
                                     int g_pix_stride, int nch, int mode, float* dx, int accumulate,
                                     void* stream) {
  CSMRI_CHECK_ARG(x && g && dx && nch >= 1 && nch <= 3);
  hipLaunchKernelGGL(complex_abs_bwd_kernel, dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)stream,
                     (const float2*)x, npix, g, g_dtype, g_pix_stride, nch, mode, (float2*)dx, accumulate);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

// ---------------------------------------------------- refinement wrapper ----
// min of the real channel and max of (x - min) per sample.  max_i fl(x_i - mn) == fl(max_i x_i - mn)
// (rounding is monotone), so one pass computes min and max; two stages keep it deterministic and
// spread a sample over MM_SPLIT workgroups.
#define MM_SPLIT 32
__global__ __launch_bounds__(256) void minmax_partial_kernel(const float2* x, long long HW, float* part) {
  __shared__ float smn[256], smx[256];
  const float2* xb = x + (size_t)blockIdx.y * HW;
  const long long chunk = (HW + MM_SPLIT - 1) / MM_SPLIT, i0 = blockIdx.x * chunk, i1 = min(HW, i0 + chunk);
  float mn = INFINITY, mx = -INFINITY;
  for (long long i = i0 + threadIdx.x; i < i1; i += 256) { const float v = xb[i].x; mn = fminf(mn, v); mx = fmaxf(mx, v); }
  smn[threadIdx.x] = mn; smx[threadIdx.x] = mx;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      smn[threadIdx.x] = fminf(smn[threadIdx.x], smn[threadIdx.x + s]);
      smx[threadIdx.x] = fmaxf(smx[threadIdx.x], smx[threadIdx.x + s]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    part[(blockIdx.y * MM_SPLIT + blockIdx.x) * 2] = smn[0];
    part[(blockIdx.y * MM_SPLIT + blockIdx.x) * 2 + 1] = smx[0];
  }
}
__global__ void minmax_final_kernel(const float* part, int B, float* mm) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float mn = INFINITY, mx = -INFINITY;
  for (int k = 0; k < MM_SPLIT; ++k) { mn = fminf(mn, part[(b * MM_SPLIT + k) * 2]); mx = fmaxf(mx, part[(b * MM_SPLIT + k) * 2 + 1]); }
  mm[2 * b] = mn; mm[2 * b + 1] = mx - mn;
}
// minmax: [B][2] results followed by B*MM_SPLIT*2 floats of scratch (csmri_minmax_floats(B) in all)
extern "C" size_t csmri_minmax_floats(int B) { return (size_t)B * 2 * (1 + MM_SPLIT); }
extern "C" int csmri_minmax_real(const float* x, int B, long long HW, float* minmax, void* stream) {
  CSMRI_CHECK_ARG(x && minmax && B > 0);
  float* part = minmax + (size_t)B * 2;
  hipLaunchKernelGGL(minmax_partial_kernel, dim3(MM_SPLIT, B), dim3(256), 0, (hipStream_t)stream, (const float2*)x,
                     HW, part);
  CSMRI_LAUNCH_CHECK();
  hipLaunchKernelGGL(minmax_final_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, part, B, minmax);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}
// same operation order as _scale/_unscale: ((x-min)/max*2-1 + s*u + 1)/2*max + min
__global__ void refine_combine_kernel(const float2* pre, const void* u, int udt, int ups,
                                      const float* sp, const float* mm, int B, long long HW,
                                      float2* pred, float* scaled) {
  const float s = sp[0];
  GRID_STRIDE(i, (long long)B * HW) {
    const int b = (int)(i / HW);
    const float mn = mm[2 * b], mx = mm[2 * b + 1];
    const float2 p = pre[i];
    const float uv = load_elem(u, i * ups, udt);
    float t = (p.x - mn) / mx;
    t = t * 2.f - 1.f;
    const float su = s * uv;
    t = t + su;
    t = (t + 1.f) / 2.f;
    t = t * mx + mn;
    pred[i] = make_float2(t, p.y);
    if (scaled) scaled[i] = su;
  }
}
extern "C" int csmri_refine_combine(const float* pre, const void* u, int u_dtype, int u_pix_stride,
                                    const float* scale_param, const float* minmax, int B, long long HW,
                                    float* pred, float* scaled, void* stream) {
  CSMRI_CHECK_ARG(pre && u && scale_param && minmax && pred);
  hipLaunchKernelGGL(refine_combine_kernel, dim3(grid_for((long long)B * HW)), dim3(256), 0,
                     (hipStream_t)stream, (const float2*)pre, u, u_dtype, u_pix_stride, scale_param,
                     minmax, B, HW, (float2*)pred, scaled);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}
// d pred_real / d u = s*max/2 ; d pred_real / d s = u*max/2.  dscale_partial: [blocks] partials,
// element 0 receives the total (second stage in the same launch via last-block-free 2 kernels).
__global__ __launch_bounds__(256) void refine_combine_bwd_kernel(
    const float2* gpred, const void* u, int udt, int ups, const float* sp, const float* mm, int B,
    long long HW, void* du, int dudt, int dups, float* partial, const float2* gpred2, const void* du2, int du2ps) {
  const float s = sp[0];
  double acc = 0.0;
  GRID_STRIDE(i, (long long)B * HW) {
    const int b = (int)(i / HW);
    const float half_mx = mm[2 * b + 1] * 0.5f;
    float g = gpred[i].x;
    if (gpred2) g += gpred2[i].x;              // pred has two consumers (discriminator input, VGG loss)
    const float uv = load_elem(u, i * ups, udt);
    // a pixel of exactly 8 channels is written whole (the value + 7 zero pad channels): the caller need not clear du
    float dv = g * s * half_mx;
    if (du2) dv += load_elem(du2, i * du2ps, dudt);   // u's second consumer (the feature penalty on the raw refinement)
    if (dups == 8 && dudt == CSMRI_BF16) ((u32x4_t*)du)[i] = (u32x4_t){(unsigned)f32_to_bf16_bits(dv), 0u, 0u, 0u};
    else if (dups == 8) { ((f32x4_t*)du)[2 * i] = (f32x4_t){dv, 0.f, 0.f, 0.f}; ((f32x4_t*)du)[2 * i + 1] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
    else store_elem(du, i * dups, dudt, dv);
    acc += (double)(g * uv * half_mx);
  }
  double tot = block_sum(acc);
  if (threadIdx.x == 0) partial[1 + blockIdx.x] = (float)tot;
}
__global__ __launch_bounds__(256) void sum_partials_kernel(float* partial, int n, float* total, int accumulate) {
  // one workgroup, fixed-order strided sums + tree: deterministic
  __shared__ double sh[256];
  double t = 0;
  for (int i = threadIdx.x; i < n; i += 256) t += partial[1 + i];
  sh[threadIdx.x] = t;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s]; __syncthreads(); }
  if (threadIdx.x == 0) {
    partial[0] = (float)sh[0];
    if (total) total[0] = (accumulate ? total[0] : 0.f) + (float)sh[0];
  }
}
extern "C" int csmri_refine_combine_bwd(const float* gpred, const void* u, int u_dtype, int u_pix_stride,
                                        const float* scale_param, const float* minmax, int B,
                                        long long HW, void* du, int du_dtype, int du_pix_stride,
                                        float* dscale_partial, const float* gpred2, const void* du2,
                                        int du2_pix_stride, float* dscale, int accumulate, void* stream) {
  CSMRI_CHECK_ARG(gpred && u && scale_param && minmax && du && dscale_partial);
  if (du_pix_stride == 8 && ((uintptr_t)du & 15)) return CSMRI_E_ALIGN;
  const int blocks = grid_for((long long)B * HW);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(refine_combine_bwd_kernel, dim3(blocks), dim3(256), 0, st, (const float2*)gpred, u,
                     u_dtype, u_pix_stride, scale_param, minmax, B, HW, du, du_dtype, du_pix_stride,
                     dscale_partial, (const float2*)gpred2, du2, du2_pix_stride);
  CSMRI_LAUNCH_CHECK();
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, dscale_partial, blocks, dscale, accumulate);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

// ------------------------------------------------------------------ losses ----
#define LOSS_BLOCKS 512
extern "C" size_t csmri_loss_work_bytes(void) { return (LOSS_BLOCKS + 1) * sizeof(float) * 2; }

__global__ __launch_bounds__(256) void loss_partial_kernel(int kind, int dt, const void* a, int aps,
                                                           const void* b, int bps, long long npix,
                                                           int C_real, float* work) {
  const int nv = (C_real + 3) >> 2;
  double acc = 0.0;
  GRID_STRIDE(i, npix * nv) {
    const int c = (int)(i % nv) * 4;
    const long long p = i / nv;
    if (c + 4 <= C_real) {
      f32x4_t x = load4(a, p * aps + c, dt);
      f32x4_t y = b ? load4(b, p * bps + c, dt) : (f32x4_t){0, 0, 0, 0};
      for (int q = 0; q < 4; ++q) { float d = x[q] - y[q]; acc += kind == 0 ? fabsf(d) : d * d; }
    } else {
      for (int q = 0; c + q < C_real; ++q) {
        float d = load_elem(a, p * aps + c + q, dt) - (b ? load_elem(b, p * bps + c + q, dt) : 0.f);
        acc += kind == 0 ? fabsf(d) : d * d;
      }
    }
  }
  double tot = block_sum(acc);
  if (threadIdx.x == 0) ((double*)work)[1 + blockIdx.x] = tot;
}
__global__ __launch_bounds__(256) void loss_final_kernel(float* work, int n, double inv_count, float* result) {
  // fixed-order tree over the per-block partials (deterministic)
  double t = 0;
  for (int i = threadIdx.x; i < n; i += 256) t += ((double*)work)[1 + i];
  t = block_sum(t);
  if (threadIdx.x == 0) result[0] = (float)(t * inv_count);
}
extern "C" int csmri_loss(int kind, int dtype, const void* a, int a_pix_stride, const void* b,
                          int b_pix_stride, long long npix, int C_real, float* result, float* work,
                          void* stream) {
  CSMRI_CHECK_ARG(a && result && work && (kind == 0 || kind == 1) && npix > 0 && C_real > 0);
  hipStream_t st = (hipStream_t)stream;
  int blocks = grid_for(npix * ((C_real + 3) / 4));
  if (blocks > LOSS_BLOCKS) blocks = LOSS_BLOCKS;
  hipLaunchKernelGGL(loss_partial_kernel, dim3(blocks), dim3(256), 0, st, kind, dtype, a, a_pix_stride,
                     b, b_pix_stride, npix, C_real, work);
  CSMRI_LAUNCH_CHECK();
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, st, work, blocks,
                     1.0 / ((double)npix * (double)C_real), result);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}
// d mean-loss / d a over vector range [first, total) with stride `stride` (a grid-stride walk), U vectors in flight per
// thread: raw loads with clamped indices first, a scheduling barrier, then arithmetic and stores (a clamped duplicate
// stores the same value again; ga is a fresh buffer, never an input).  The rolled form -- one dependent load / store
// pair per iteration behind 64-bit index divisions -- ran the 8 MB feature-penalty gradient in 58-78 us.
template <int DT, bool HASB>
__device__ __forceinline__ void loss_bwd_range(int kind, const void* __restrict__ a, int aps,
                                               const void* __restrict__ b, int bps, unsigned total, unsigned nv,
                                               int C_real, float k, void* __restrict__ ga, int gaps, unsigned first,
                                               unsigned stride) {
  constexpr int U = 4;
  for (unsigned i0 = first; i0 < total; i0 += stride * U) {
    typename raw4<DT>::t ra[U], rb[U];
    unsigned pp[U];
    int cc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned i = min(i0 + u * stride, total - 1);
      pp[u] = i / nv;
      cc[u] = (int)(i - pp[u] * nv) * 4;
      ra[u] = ldraw<DT>(a, (long long)pp[u] * aps + cc[u]);
      if constexpr (HASB) rb[u] = ldraw<DT>(b, (long long)pp[u] * bps + cc[u]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const f32x4_t x = cvt4<DT>(ra[u]);
      f32x4_t y = (f32x4_t){0.f, 0.f, 0.f, 0.f}, g;
      if constexpr (HASB) y = cvt4<DT>(rb[u]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float d = x[q] - y[q];
        const float v = kind == 0 ? (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : 2.f * d;
        g[q] = (cc[u] + q < C_real) ? v * k : 0.f;
      }
      store4(ga, (long long)pp[u] * gaps + cc[u], DT, g);
    }
  }
}
template <int DT, bool HASB>
__global__ __launch_bounds__(256) void loss_bwd_kernel(int kind, const void* a, int aps, const void* b, int bps,
                                                       unsigned total, unsigned nv, int C_real, const float* coeff,
                                                       float scale, void* ga, int gaps) {
  const float k = (coeff ? coeff[0] : 1.f) * scale;
  loss_bwd_range<DT, HASB>(kind, a, aps, b, bps, total, nv, C_real, k, ga, gaps, blockIdx.x * 256 + threadIdx.x,
                           gridDim.x * 256);
}
// accumulating form (read-modify-write of ga: no duplicate stores, plain loop)
__global__ void loss_bwd_acc_kernel(int kind, int dt, const void* a, int aps, const void* b, int bps,
                                    long long npix, int C, int C_real, const float* coeff, float scale,
                                    void* ga, int gaps) {
  const int nv = C >> 2;
  const float k = (coeff ? coeff[0] : 1.f) * scale;
  GRID_STRIDE(i, npix * nv) {
    const int c = (int)(i % nv) * 4;
    const long long p = i / nv;
    f32x4_t x = load4(a, p * aps + c, dt);
    f32x4_t y = b ? load4(b, p * bps + c, dt) : (f32x4_t){0, 0, 0, 0};
    f32x4_t g;
    for (int q = 0; q < 4; ++q) {
      const float d = x[q] - y[q];
      float v = kind == 0 ? (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : 2.f * d;
      g[q] = (c + q < C_real) ? v * k : 0.f;
    }
    g += load4(ga, p * gaps + c, dt);
    store4(ga, p * gaps + c, dt, g);
  }
}
extern "C" int csmri_loss_bwd(int kind, int dtype, const void* a, int a_pix_stride, const void* b,
                              int b_pix_stride, long long npix, int C, int C_real, const float* coeff,
                              float weight, void* ga, int ga_pix_stride, int accumulate, void* stream) {
  CSMRI_CHECK_ARG(a && ga && C % 4 == 0 && ga != a && ga != b);
  const float scale = weight / ((float)npix * (float)C_real);
  hipStream_t st = (hipStream_t)stream;
  const long long total = npix * (C / 4);
  if (accumulate) {
    hipLaunchKernelGGL(loss_bwd_acc_kernel, dim3(grid_for(total)), dim3(256), 0, st, kind, dtype, a, a_pix_stride, b,
                       b_pix_stride, npix, C, C_real, coeff, scale, ga, ga_pix_stride);
  } else {
    if (total >= (1ll << 31)) return CSMRI_E_UNSUPPORTED;
    long long blocks = (total + 256 * 4 - 1) / (256 * 4);
    if (blocks > 4096) blocks = 4096;
#define LB2(DT_, HB_) hipLaunchKernelGGL((loss_bwd_kernel<DT_, HB_>), dim3((int)blocks), dim3(256), 0, st, kind, a, a_pix_stride, b, b_pix_stride, (unsigned)total, (unsigned)(C / 4), C_real, coeff, scale, ga, ga_pix_stride)
#define LB(DT_) do { if (b) LB2(DT_, true); else LB2(DT_, false); } while (0)
    if (dtype == CSMRI_BF16) LB(CSMRI_BF16); else LB(CSMRI_F32);
#undef LB
#undef LB2
  }
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

// ---- several mean losses in one launch (feature matching: one L1 per discriminator layer) ----
#define LOSS_MULTI_BLOCKS 256
struct LossItems { csmri_loss_item it[CSMRI_LOSS_MAX_ITEMS]; };
extern "C" size_t csmri_loss_multi_work_bytes(int n) { return (size_t)n * LOSS_MULTI_BLOCKS * sizeof(double); }

__global__ __launch_bounds__(256) void loss_multi_partial_kernel(int kind, int dt, const LossItems L, double* work) {
  const csmri_loss_item& t = L.it[blockIdx.y];
  if (t.dtype_plus1) dt = t.dtype_plus1 - 1;
  const int nv = (t.C_real + 3) >> 2;
  const unsigned total = (unsigned)(t.npix * nv);            // host: < 2^31
  // four vectors in flight per thread, four independent double accumulators (the round-2 loop -- one 8-byte load,
  // four dependent double adds per iteration, 64 workgroups per tensor -- ran at 0.9 TB/s: 74 us per launch)
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  auto one = [&](unsigned i) -> float {
    const unsigned p = i / nv;
    const int c = (int)(i - p * nv) * 4;
    float s = 0.f;
    if (c + 4 <= t.C_real) {
      f32x4_t x = load4(t.a, (long long)p * t.a_pix_stride + c, dt);
      f32x4_t y = t.b ? load4(t.b, (long long)p * t.b_pix_stride + c, dt) : (f32x4_t){0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < 4; ++q) { float d = x[q] - y[q]; s += kind == 0 ? fabsf(d) : d * d; }
    } else {
      for (int q = 0; c + q < t.C_real; ++q) {
        float d = load_elem(t.a, (long long)p * t.a_pix_stride + c + q, dt) -
                  (t.b ? load_elem(t.b, (long long)p * t.b_pix_stride + c + q, dt) : 0.f);
        s += kind == 0 ? fabsf(d) : d * d;
      }
    }
    return s;
  };
  const unsigned stride = LOSS_MULTI_BLOCKS * 256;
  unsigned i = blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < total; i += 4 * stride) {
    const float s0 = one(i), s1 = one(i + stride), s2 = one(i + 2 * stride), s3 = one(i + 3 * stride);
    acc[0] += (double)s0; acc[1] += (double)s1; acc[2] += (double)s2; acc[3] += (double)s3;
  }
  for (; i < total; i += stride) acc[0] += (double)one(i);
  double tot = block_sum((acc[0] + acc[1]) + (acc[2] + acc[3]));
  if (threadIdx.x == 0) work[blockIdx.y * LOSS_MULTI_BLOCKS + blockIdx.x] = tot;
}
__global__ __launch_bounds__(64) void loss_multi_final_kernel(const LossItems L, int n, const double* work, float* result) {
  // lane i < n sums item i's partials in fixed order; lane 0 combines the weighted means in order
  __shared__ double means[CSMRI_LOSS_MAX_ITEMS];
  const int i = threadIdx.x;
  if (i < n) {
    double t = 0;
    for (int k = 0; k < LOSS_MULTI_BLOCKS; ++k) t += work[i * LOSS_MULTI_BLOCKS + k];     // fixed order
    means[i] = t / ((double)L.it[i].npix * (double)L.it[i].C_real);
    result[1 + i] = (float)means[i];
  }
  __syncthreads();
  if (i == 0) {
    double tot = 0;
    for (int k = 0; k < n; ++k) tot += (double)L.it[k].weight * means[k];
    result[0] = (float)tot;
  }
}
static int loss_items_ok(const csmri_loss_item* items, int n) {
  if (!items || n < 1 || n > CSMRI_LOSS_MAX_ITEMS) return 0;
  for (int i = 0; i < n; ++i)
    if (!items[i].a || items[i].npix <= 0 || items[i].C_real <= 0 || items[i].dtype_plus1 < 0 || items[i].dtype_plus1 > 2 ||
        items[i].npix * ((items[i].C > items[i].C_real ? items[i].C : items[i].C_real) + 3) >= (1ll << 31)) return 0;
  return 1;
}
extern "C" int csmri_loss_multi(int kind, int dtype, const csmri_loss_item* items, int n, float* result,
                                float* work, void* stream) {
  CSMRI_CHECK_ARG(result && work && (kind == 0 || kind == 1) && loss_items_ok(items, n));
  LossItems L;
  for (int i = 0; i < n; ++i) L.it[i] = items[i];
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(loss_multi_partial_kernel, dim3(LOSS_MULTI_BLOCKS, n), dim3(256), 0, st, kind, dtype, L,
                     (double*)work);
  CSMRI_LAUNCH_CHECK();
  hipLaunchKernelGGL(loss_multi_final_kernel, dim3(1), dim3(64), 0, st, L, n, (const double*)work, result);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}
__global__ __launch_bounds__(256) void loss_multi_bwd_kernel(int kind, int dt, const LossItems L, const float* coeff) {
  const csmri_loss_item& t = L.it[blockIdx.y];
  if (t.dtype_plus1) dt = t.dtype_plus1 - 1;
  const unsigned nv = t.C >> 2;
  const float k = (coeff ? coeff[0] : 1.f) * (t.weight / ((float)t.npix * (float)t.C_real));
  const unsigned total = (unsigned)(t.npix * nv), first = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
  // (workgroup-uniform dispatch to the batched walk of loss_bwd_range)
  if (dt == CSMRI_BF16) {
    if (t.b) loss_bwd_range<CSMRI_BF16, true>(kind, t.a, t.a_pix_stride, t.b, t.b_pix_stride, total, nv, t.C_real, k, t.ga, t.ga_pix_stride, first, stride);
    else loss_bwd_range<CSMRI_BF16, false>(kind, t.a, t.a_pix_stride, t.b, t.b_pix_stride, total, nv, t.C_real, k, t.ga, t.ga_pix_stride, first, stride);
  } else {
    if (t.b) loss_bwd_range<CSMRI_F32, true>(kind, t.a, t.a_pix_stride, t.b, t.b_pix_stride, total, nv, t.C_real, k, t.ga, t.ga_pix_stride, first, stride);
    else loss_bwd_range<CSMRI_F32, false>(kind, t.a, t.a_pix_stride, t.b, t.b_pix_stride, total, nv, t.C_real, k, t.ga, t.ga_pix_stride, first, stride);
  }
}
extern "C" int csmri_loss_multi_bwd(int kind, int dtype, const csmri_loss_item* items, int n, const float* coeff,
                                    void* stream) {
  CSMRI_CHECK_ARG((kind == 0 || kind == 1) && loss_items_ok(items, n));
  LossItems L;
  for (int i = 0; i < n; ++i) {
    CSMRI_CHECK_ARG(items[i].ga && items[i].C % 4 == 0 && items[i].ga != items[i].a && items[i].ga != items[i].b);
    L.it[i] = items[i];
  }
  hipLaunchKernelGGL(loss_multi_bwd_kernel, dim3(256, n), dim3(256), 0, (hipStream_t)stream, kind, dtype, L, coeff);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

// BCE(sigmoid(l), t) with torch's log clamp at -100; one block (n is tiny: B*5*5)
__global__ __launch_bounds__(256) void bce_logits_kernel(const float* l, long long n, float t, float* prob,
                                                         float* result) {
  double acc = 0;
  for (long long i = threadIdx.x; i < n; i += 256) {
    const float p = 1.f / (1.f + expf(-l[i]));
    if (prob) prob[i] = p;
    const float lp = fmaxf(logf(p), -100.f), l1p = fmaxf(logf(1.f - p), -100.f);
    acc += (double)(-(t * lp + (1.f - t) * l1p));
  }
  double tot = block_sum(acc);
  if (threadIdx.x == 0) result[0] = (float)(tot / (double)n);
}
extern "C" int csmri_bce_logits(const float* logits, long long n, float target, float* prob,
                                float* result, void* stream) {
  CSMRI_CHECK_ARG(logits && result && n > 0);
  hipLaunchKernelGGL(bce_logits_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, n, target, prob, result);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}
// d/dl mean BCE = (p - t)/n (un-clamped region); matches autograd of BCE(sigmoid)
// = (p-t)/(p(1-p)) * p(1-p)
__global__ void bce_logits_bwd_kernel(const float* l, long long n, float t, const float* coeff, float w,
                                      float* g, int accumulate) {
  const float k = (coeff ? coeff[0] : 1.f) * w / (float)n;
  GRID_STRIDE(i, n) {
    const float p = 1.f / (1.f + expf(-l[i]));
    const float v = (p - t) * k;
    g[i] = accumulate ? g[i] + v : v;
  }
}
extern "C" int csmri_bce_logits_bwd(const float* logits, long long n, float target, const float* coeff,
                                    float weight, float* glogits, int accumulate, void* stream) {
  CSMRI_CHECK_ARG(logits && glogits && n > 0);
  hipLaunchKernelGGL(bce_logits_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, logits,
                     n, target, coeff, weight, glogits, accumulate);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

// per-image MSE between clamp(|pred|,0,1) and clamp(|target|,0,1); one block per image
__global__ __launch_bounds__(256) void psnr_mse_kernel(const float2* pred, const float2* tgt, long long HW,
                                                       float* mse) {
  // grid = (image, PSNR_SPLIT): partial sums per slice of the image, combined below
  const float2* p = pred + (size_t)blockIdx.x * HW;
  const float2* t = tgt + (size_t)blockIdx.x * HW;
  const long long chunk = (HW + gridDim.y - 1) / gridDim.y;
  const long long i0 = blockIdx.y * chunk, i1 = min(HW, i0 + chunk);
  double acc = 0;
  for (long long i = i0 + threadIdx.x; i < i1; i += 256) {
    float a = fminf(fmaxf(sqrtf(p[i].x * p[i].x + p[i].y * p[i].y), 0.f), 1.f);
    float b = fminf(fmaxf(sqrtf(t[i].x * t[i].x + t[i].y * t[i].y), 0.f), 1.f);
    float d = a - b;
    acc += (double)(d * d);
  }
  double tot = block_sum(acc);
  if (threadIdx.x == 0) mse[blockIdx.x * gridDim.y + blockIdx.y] = (float)tot;
}
#define PSNR_SPLIT 32
__global__ void psnr_final_kernel(const float* part, int B, double inv_hw, float* mse) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double t = 0;
  for (int j = 0; j < PSNR_SPLIT; ++j) t += part[b * PSNR_SPLIT + j];
  mse[b] = (float)(t * inv_hw);
}
extern "C" int csmri_psnr_mse(const float* pred, const float* target, int B, long long HW, float* mse,
                              void* stream) {
  CSMRI_CHECK_ARG(pred && target && mse && B > 0);
  // mse must hold B * (1 + PSNR_SPLIT) floats: [0,B) results, then the partials
  float* part = mse + B;
  hipLaunchKernelGGL(psnr_mse_kernel, dim3(B, PSNR_SPLIT), dim3(256), 0, (hipStream_t)stream,
                     (const float2*)pred, (const float2*)target, HW, part);
  CSMRI_LAUNCH_CHECK();
  hipLaunchKernelGGL(psnr_final_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, part, B,
                     1.0 / (double)HW, mse);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

// ---------------------------------------------------------------------- SSIM ----
// SSIM of clamp(|pred|,0,1) vs clamp(|target|,0,1) per image (metrics/pytorch_ssim/__init__.py:22-42
// through metrics/image_metrics.py:22-42): 11x11 gaussian window (sigma 1.5), zero padding,
// C1 = 0.01^2, C2 = 0.03^2, mean of the SSIM map.  One workgroup per 16x16 tile: the magnitudes of
// the tile plus a 5-pixel halo go to LDS once, the five windowed moments are computed separably
// (row pass into LDS, column pass in registers), the tile's map values are summed in fixed order.
#define SSIM_R 5
#define SSIM_T 16
#define SSIM_P (SSIM_T + 2 * SSIM_R)
struct SsimWin { float g[2 * SSIM_R + 1]; };
__global__ __launch_bounds__(256) void ssim_tile_kernel(const float2* pred, const float2* tgt, int H, int W,
                                                        int tiles_x, int tiles_y, const SsimWin win, double* part) {
  __shared__ float a[SSIM_P][SSIM_P + 1], b[SSIM_P][SSIM_P + 1];
  __shared__ float hrow[5][SSIM_P][SSIM_T + 1];
  const int img = blockIdx.y, tile = blockIdx.x;
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int y0 = ty * SSIM_T - SSIM_R, x0 = tx * SSIM_T - SSIM_R;
  const float2* pi = pred + (size_t)img * H * W;
  const float2* ti = tgt + (size_t)img * H * W;
  for (int e = threadIdx.x; e < SSIM_P * SSIM_P; e += 256) {
    const int r = e / SSIM_P, c = e - r * SSIM_P, y = y0 + r, x = x0 + c;
    float va = 0.f, vb = 0.f;
    if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
      const float2 p = pi[(size_t)y * W + x], t = ti[(size_t)y * W + x];
      va = fminf(fmaxf(sqrtf(p.x * p.x + p.y * p.y), 0.f), 1.f);
      vb = fminf(fmaxf(sqrtf(t.x * t.x + t.y * t.y), 0.f), 1.f);
    }
    a[r][c] = va; b[r][c] = vb;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < SSIM_P * SSIM_T; e += 256) {      // row pass
    const int r = e / SSIM_T, c = e - r * SSIM_T;
    float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
    for (int k = 0; k <= 2 * SSIM_R; ++k) {
      const float g = win.g[k], va = a[r][c + k], vb = b[r][c + k];
      m1 += g * va; m2 += g * vb; e11 += g * (va * va); e22 += g * (vb * vb); e12 += g * (va * vb);
    }
    hrow[0][r][c] = m1; hrow[1][r][c] = m2; hrow[2][r][c] = e11; hrow[3][r][c] = e22; hrow[4][r][c] = e12;
  }
  __syncthreads();
  const int r = threadIdx.x / SSIM_T, c = threadIdx.x - r * SSIM_T;     // column pass: one pixel per thread
  float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
  for (int k = 0; k <= 2 * SSIM_R; ++k) {
    const float g = win.g[k];
    m1 += g * hrow[0][r + k][c]; m2 += g * hrow[1][r + k][c]; e11 += g * hrow[2][r + k][c];
    e22 += g * hrow[3][r + k][c]; e12 += g * hrow[4][r + k][c];
  }
  double v = 0.0;
  if (ty * SSIM_T + r < H && tx * SSIM_T + c < W) {
    const float mu1_sq = m1 * m1, mu2_sq = m2 * m2, mu12 = m1 * m2;
    const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu12;
    const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
    v = (double)(((2.f * mu12 + c1) * (2.f * s12 + c2)) / ((mu1_sq + mu2_sq + c1) * (s1 + s2 + c2)));
  }
  const double tot = block_sum(v);
  if (threadIdx.x == 0) part[(size_t)img * tiles_x * tiles_y + tile] = tot;
}
__global__ __launch_bounds__(256) void ssim_final_kernel(const double* part, int ntiles, double inv_hw, float* out) {
  double t = 0;
  for (int i = threadIdx.x; i < ntiles; i += 256) t += part[(size_t)blockIdx.x * ntiles + i];
  t = block_sum(t);
  if (threadIdx.x == 0) out[blockIdx.x] = (float)(t * inv_hw);
}
extern "C" size_t csmri_ssim_work_bytes(int B, int H, int W) {
  return (size_t)B * ((H + SSIM_T - 1) / SSIM_T) * ((W + SSIM_T - 1) / SSIM_T) * sizeof(double);
}
extern "C" int csmri_ssim(const float* pred, const float* target, int B, int H, int W, float* ssim, void* work,
                          void* stream) {
  CSMRI_CHECK_ARG(pred && target && ssim && work && B > 0 && H > 0 && W > 0);
  SsimWin win;
  float sum = 0.f;                       // the reference builds the window in float32 (torch.Tensor)
  for (int k = 0; k <= 2 * SSIM_R; ++k) {
    win.g[k] = (float)exp(-(double)((k - SSIM_R) * (k - SSIM_R)) / (2.0 * 1.5 * 1.5));
    sum += win.g[k];
  }
  for (int k = 0; k <= 2 * SSIM_R; ++k) win.g[k] /= sum;
  const int tx = (W + SSIM_T - 1) / SSIM_T, ty = (H + SSIM_T - 1) / SSIM_T;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ssim_tile_kernel, dim3(tx * ty, B), dim3(256), 0, st, (const float2*)pred, (const float2*)target,
                     H, W, tx, ty, win, (double*)work);
  CSMRI_LAUNCH_CHECK();
  hipLaunchKernelGGL(ssim_final_kernel, dim3(B), dim3(256), 0, st, (const double*)work, tx * ty,
                     1.0 / ((double)H * (double)W), ssim);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

// -------------------------------------------------------------------- Adam ----
// torch.optim.Adam (2.x): denom = sqrt(v)/sqrt(1-b2^t) + eps; p -= lr/(1-b1^t) * m/denom
__global__ void adam_kernel(float* p, const float* g, float* m, float* v, long long n, float lr,
                            float b1, float b2, float eps, float bc1, float bc2_sqrt, float gscale) {
  const float step_size = lr / bc1;
  GRID_STRIDE(i4, (n + 3) / 4) {
    const long long i = i4 * 4;
    if (i + 4 <= n) {
      f32x4_t pp = *(f32x4_t*)(p + i), gg = *(const f32x4_t*)(g + i), mm = *(f32x4_t*)(m + i), vv = *(f32x4_t*)(v + i);
      for (int q = 0; q < 4; ++q) {
        const float gq = gg[q] * gscale;
        mm[q] = b1 * mm[q] + (1.f - b1) * gq;
        vv[q] = b2 * vv[q] + (1.f - b2) * gq * gq;
        pp[q] -= step_size * (mm[q] / (sqrtf(vv[q]) / bc2_sqrt + eps));
      }
      *(f32x4_t*)(p + i) = pp; *(f32x4_t*)(m + i) = mm; *(f32x4_t*)(v + i) = vv;
    } else {
      for (long long j = i; j < n; ++j) {
        const float gq = g[j] * gscale;
        m[j] = b1 * m[j] + (1.f - b1) * gq;
        v[j] = b2 * v[j] + (1.f - b2) * gq * gq;
        p[j] -= step_size * (m[j] / (sqrtf(v[j]) / bc2_sqrt + eps));
      }
    }
  }
}
extern "C" int csmri_adam(float* p, const float* g, float* m, float* v, long long n, float lr,
                          float beta1, float beta2, float eps, int step, float grad_scale, void* stream) {
  CSMRI_CHECK_ARG(p && g && m && v && n > 0 && step >= 1);
  if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return CSMRI_E_ALIGN;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p, g, m,
                     v, n, lr, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2), grad_scale);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

// Same update with the step count held in device memory (*step_dev = number of steps already
// taken; this launch performs step *step_dev + 1 and a trailing 1-thread kernel increments the
// counter) -- nothing step-dependent is baked into the launch, so the optimizer can live inside
// a captured hipGraph and be replayed.
__global__ void adam_dev_kernel(float* p, const float* g, float* m, float* v, long long n, float lr_arg,
                                float b1, float b2, float eps, const int* step_dev, float gscale, const float* lr_dev) {
  const int step = *step_dev + 1;
  const float lr = lr_dev ? *lr_dev : lr_arg;         // (csmri_adam_dev_lr: the rate lives in device memory)
  const float bc1 = (float)(1.0 - pow((double)b1, (double)step));
  const float bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, (double)step));
  const float step_size = lr / bc1;
  GRID_STRIDE(i4, (n + 3) / 4) {
    const long long i = i4 * 4;
    if (i + 4 <= n) {
      f32x4_t pp = *(f32x4_t*)(p + i), gg = *(const f32x4_t*)(g + i), mm = *(f32x4_t*)(m + i), vv = *(f32x4_t*)(v + i);
      for (int q = 0; q < 4; ++q) {
        const float gq = gg[q] * gscale;
        mm[q] = b1 * mm[q] + (1.f - b1) * gq;
        vv[q] = b2 * vv[q] + (1.f - b2) * gq * gq;
        pp[q] -= step_size * (mm[q] / (sqrtf(vv[q]) / bc2_sqrt + eps));
      }
      *(f32x4_t*)(p + i) = pp; *(f32x4_t*)(m + i) = mm; *(f32x4_t*)(v + i) = vv;
    } else {
      for (long long j = i; j < n; ++j) {
        const float gq = g[j] * gscale;
        m[j] = b1 * m[j] + (1.f - b1) * gq;
        v[j] = b2 * v[j] + (1.f - b2) * gq * gq;
        p[j] -= step_size * (m[j] / (sqrtf(v[j]) / bc2_sqrt + eps));
      }
    }
  }
}
__global__ void incr_kernel(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) *p += 1; }
extern "C" int csmri_adam_dev(float* p, const float* g, float* m, float* v, long long n, float lr,
                              float beta1, float beta2, float eps, int* step_dev, float grad_scale,
                              void* stream) {
  CSMRI_CHECK_ARG(p && g && m && v && n > 0 && step_dev);
  if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return CSMRI_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(adam_dev_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, st, p, g, m, v, n, lr, beta1,
                     beta2, eps, (const int*)step_dev, grad_scale, (const float*)nullptr);
  CSMRI_LAUNCH_CHECK();
  hipLaunchKernelGGL(incr_kernel, dim3(1), dim3(64), 0, st, step_dev);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}
// The same update with the learning rate read from device memory: NOTHING a scheduler changes is baked into the launch,
// so a captured hipGraph follows an LR schedule without being captured again (the host writes *lr_dev before a replay).
extern "C" int csmri_adam_dev_lr(float* p, const float* g, float* m, float* v, long long n, const float* lr_dev,
                                 float beta1, float beta2, float eps, int* step_dev, float grad_scale,
                                 void* stream) {
  CSMRI_CHECK_ARG(p && g && m && v && n > 0 && step_dev && lr_dev);
  if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return CSMRI_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(adam_dev_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, st, p, g, m, v, n, 0.f, beta1,
                     beta2, eps, (const int*)step_dev, grad_scale, lr_dev);
  CSMRI_LAUNCH_CHECK();
  hipLaunchKernelGGL(incr_kernel, dim3(1), dim3(64), 0, st, step_dev);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

// --------------------------------------------------------- image pool ----
// One query of the history pool of generated images (reference utils/image_pool.py:8-60) on the device, given
// the host's plan (int64 [5][n]: kind, pool_idx, x_idx, write_slot, write_src):
//   out[i]              = kind[i] == 1 ? pool[pool_idx[i]] : kind[i] == 2 ? x[x_idx[i]] : x[i]
//   pool[write_slot[j]] = x[write_src[j]]                     (slot == pool_size: the write-only dummy)
// A thread owns one 16-byte column of EVERY image: it performs all n reads of the old pool content before its
// n writes, so the gather-before-scatter order of the sequential reference holds without a grid-wide barrier.
#define POOL_MAX_N 64
__global__ __launch_bounds__(256) void image_pool_exchange_kernel(const uint4* __restrict__ x, uint4* pool, uint4* __restrict__ out,
                                                                  const long long* __restrict__ plan, int n, long long vecs) {
  __shared__ int s_plan[5 * POOL_MAX_N];
  for (int t = threadIdx.x; t < 5 * n; t += blockDim.x) s_plan[t] = (int)plan[t];
  __syncthreads();
  GRID_STRIDE(v, vecs) {
    for (int i = 0; i < n; ++i) {
      const int kind = s_plan[i];
      const uint4* src = kind == 1 ? pool + (long long)s_plan[n + i] * vecs : x + (long long)(kind == 2 ? s_plan[2 * n + i] : i) * vecs;
      out[(long long)i * vecs + v] = src[v];
    }
    for (int j = 0; j < n; ++j)
      pool[(long long)s_plan[3 * n + j] * vecs + v] = x[(long long)s_plan[4 * n + j] * vecs + v];
  }
}
extern "C" int csmri_image_pool_exchange(const void* x, void* pool, void* out, const long long* plan, int n,
                                         long long bytes_per_image, void* stream) {
  CSMRI_CHECK_ARG(x && pool && out && plan && n > 0 && n <= POOL_MAX_N && bytes_per_image > 0 &&
                  bytes_per_image % 16 == 0);
  if (((uintptr_t)x | (uintptr_t)pool | (uintptr_t)out) & 15) return CSMRI_E_ALIGN;
  const long long vecs = bytes_per_image / 16;
  hipLaunchKernelGGL(image_pool_exchange_kernel, dim3(grid_for(vecs)), dim3(256), 0, (hipStream_t)stream,
                     (const uint4*)x, (uint4*)pool, (uint4*)out, plan, n, vecs);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

// ------------------------------------------------------------ Dropout2d masks ----
// nn.Dropout2d(p) of the discriminator (reference models/discriminators.py:150-152) draws one Bernoulli(1 - p) per
// (image, channel).  All masks of one forward pass are ONE launch: element i of the flat buffer is
//   keep_i / (1 - p),  keep_i = [ uniform_i < 1 - p ],  uniform_i = (Philox4x32-10(counter = (i / 4, 0, call, call >> 32),
//                                                                    key = seed)[i % 4] >> 8) * 2^-24
// with (seed, call) in DEVICE memory (state[0], state[1]); the launch increments `call`, so a replayed hipGraph
// draws fresh masks every replay and an eager run draws the very same sequence.  One workgroup: the state is read
// by every thread before thread 0 bumps it (a barrier apart), no second launch.
__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned (&k)[2]) {
  const unsigned hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
  const unsigned hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
  const unsigned n0 = hi1 ^ c[1] ^ k[0], n2 = hi0 ^ c[3] ^ k[1];
  c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
  k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
}
// state = {seed, call, arrivals}: every workgroup reads `call` when it starts and takes a ticket when it is done; the
// one that draws the last ticket advances `call` and clears the tickets -- after every workgroup has read the old value,
// so the launch may span the chip (one workgroup of 1024 threads took 33 us for the 43,008 values of the 24-image
// discriminator pass; it sits at the head of the discriminator chain).
__global__ __launch_bounds__(256) void dropout2d_mask_kernel(float* __restrict__ out, long long n, float p,
                                                             unsigned long long* state) {
  const unsigned long long seed = state[0], call = state[1];
  const float keep_p = 1.f - p, scale = 1.f / (1.f - p);
  for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g * 4 < n; g += (long long)gridDim.x * 256) {
    unsigned c[4] = {(unsigned)g, (unsigned)(g >> 32), (unsigned)call, (unsigned)(call >> 32)};
    unsigned k[2] = {(unsigned)seed, (unsigned)(seed >> 32)};
#pragma unroll
    for (int r = 0; r < 10; ++r) philox_round(c, k);
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (float)(c[e] >> 8) * (1.f / 16777216.f) < keep_p ? scale : 0.f;
    if (g * 4 + 3 < n) *(f32x4_t*)(out + g * 4) = (f32x4_t){v[0], v[1], v[2], v[3]};
    else
      for (int e = 0; e < 4; ++e) if (g * 4 + e < n) out[g * 4 + e] = v[e];
  }
  __syncthreads();                                  // every thread of the workgroup has read `call`
  if (threadIdx.x == 0) {
    const unsigned long long t = atomicAdd(&state[2], 1ull);
    if (t == (unsigned long long)gridDim.x - 1) { state[2] = 0ull; state[1] = call + 1; }
  }
}
extern "C" int csmri_dropout2d_mask(float* mask, long long n, float p, unsigned long long* state, void* stream) {
  CSMRI_CHECK_ARG(mask && state && n > 0 && n <= (1ll << 24) && p >= 0.f && p < 1.f);
  if (((uintptr_t)state & 7) || ((uintptr_t)mask & 15)) return CSMRI_E_ALIGN;
  long long blocks = (n / 4 + 255) / 256;
  if (blocks > 256) blocks = 256;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(dropout2d_mask_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, mask, n, p, state);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

// ------------------------------------------------------------------ scalar glue of the step ----
// The runner's scalar arithmetic as single launches (each was 3-9 framework launches inside the captured step):
//   csmri_weighted_sum      total = sum_i w_i * loss_i            (reference base_runner / adversarial_runner.py:314-320:
//                           `torch.sum(torch.cat(losses) * weights)`), fp32, in list order; backward g_i = g * w_i
//   csmri_bce_logits_pair   mean BCE(sigmoid(l[0:n]), t0) + mean BCE(sigmoid(l[n:2n]), t1): the discriminator's GAN loss
//                           on the [fake; real] logits of one batched pass (models/adversarial_loss.py:71-85)
//   csmri_psnr_mean         mean_b 10 log10(1 / mse_b)            (metrics/image_metrics.py:7-19, metrics/__init__.py:38-72)
//   csmri_disc_accuracy     binary accuracy of per-image mean probabilities against label 0 (fake) / 1 (real)
//                           (metrics/scalar_metrics.py:26-53)
__global__ void weighted_sum_kernel(const csmri_scalar_list L, float* out) {
  if (threadIdx.x || blockIdx.x) return;
  float s = 0.f;
  for (int i = 0; i < L.n; ++i) s += L.w[i] * L.v[i][0];
  out[0] = s;
}
__global__ void weighted_sum_bwd_kernel(const csmri_scalar_list L, const float* g, float* out) {
  const int i = threadIdx.x;
  if (i < L.n) out[i] = g[0] * L.w[i];
}
extern "C" int csmri_weighted_sum(const csmri_scalar_list* items, float* out, void* stream) {
  CSMRI_CHECK_ARG(items && out && items->n > 0 && items->n <= CSMRI_SCALAR_LIST_MAX);
  for (int i = 0; i < items->n; ++i) CSMRI_CHECK_ARG(items->v[i]);
  hipLaunchKernelGGL(weighted_sum_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, *items, out);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}
extern "C" int csmri_weighted_sum_bwd(const csmri_scalar_list* items, const float* g, float* out, void* stream) {
  CSMRI_CHECK_ARG(items && g && out && items->n > 0 && items->n <= CSMRI_SCALAR_LIST_MAX);
  hipLaunchKernelGGL(weighted_sum_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, *items, g, out);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

__global__ __launch_bounds__(256) void bce_logits_pair_kernel(const float* l, long long n, float t0, float t1, float* result) {
  double a0 = 0, a1 = 0;
  for (long long i = threadIdx.x; i < 2 * n; i += 256) {
    const float t = i < n ? t0 : t1;
    const float p = 1.f / (1.f + expf(-l[i]));
    const float lp = fmaxf(logf(p), -100.f), l1p = fmaxf(logf(1.f - p), -100.f);
    const double v = (double)(-(t * lp + (1.f - t) * l1p));
    if (i < n) a0 += v; else a1 += v;
  }
  const double s0 = block_sum(a0);
  __syncthreads();
  const double s1 = block_sum(a1);
  if (threadIdx.x == 0) {
    const float m0 = (float)(s0 / (double)n), m1 = (float)(s1 / (double)n);
    result[0] = m0 + m1; result[1] = m0; result[2] = m1;
  }
}
extern "C" int csmri_bce_logits_pair(const float* logits, long long n_half, float t_first, float t_second, float* result,
                                     void* stream) {
  CSMRI_CHECK_ARG(logits && result && n_half > 0);
  hipLaunchKernelGGL(bce_logits_pair_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, n_half, t_first,
                     t_second, result);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}
__global__ void bce_logits_pair_bwd_kernel(const float* l, long long n, float t0, float t1, const float* coeff, float* g) {
  const float k = (coeff ? coeff[0] : 1.f) / (float)n;
  GRID_STRIDE(i, 2 * n) {
    const float p = 1.f / (1.f + expf(-l[i]));
    g[i] = (p - (i < n ? t0 : t1)) * k;
  }
}
extern "C" int csmri_bce_logits_pair_bwd(const float* logits, long long n_half, float t_first, float t_second,
                                         const float* coeff, float* glogits, void* stream) {
  CSMRI_CHECK_ARG(logits && glogits && n_half > 0);
  hipLaunchKernelGGL(bce_logits_pair_bwd_kernel, dim3(grid_for(2 * n_half)), dim3(256), 0, (hipStream_t)stream, logits,
                     n_half, t_first, t_second, coeff, glogits);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

__global__ void psnr_mean_kernel(const float* mse, int B, float* out) {
  if (threadIdx.x || blockIdx.x) return;
  double s = 0;
  for (int b = 0; b < B; ++b) s += (10.0 / log(10.0)) * log(1.0 / (double)mse[b]);
  out[0] = (float)(s / (double)B);
}
extern "C" int csmri_psnr_mean(const float* mse, int B, float* out, void* stream) {
  CSMRI_CHECK_ARG(mse && out && B > 0);
  hipLaunchKernelGGL(psnr_mean_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, mse, B, out);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

__global__ __launch_bounds__(64) void disc_accuracy_kernel(const float* pf, const float* pr, int B, int n, float* out) {
  // lane b: image b of the fake and/or the real batch (B <= 64 images each); per-image mean in fp32, in element order
  const int b = threadIdx.x;
  float hits = 0.f;
  if (b < B) {
    if (pf) { float s = 0.f; for (int i = 0; i < n; ++i) s += pf[(size_t)b * n + i]; hits += (s / (float)n > 0.5f) ? 0.f : 1.f; }
    if (pr) { float s = 0.f; for (int i = 0; i < n; ++i) s += pr[(size_t)b * n + i]; hits += (s / (float)n > 0.5f) ? 1.f : 0.f; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) hits += __shfl_xor(hits, o);
  if (b == 0) out[0] = hits / (float)(B * ((pf ? 1 : 0) + (pr ? 1 : 0)));
}
extern "C" int csmri_disc_accuracy(const float* prob_fake, const float* prob_real, int B, int n_per_image, float* out,
                                   void* stream) {
  CSMRI_CHECK_ARG((prob_fake || prob_real) && out && B > 0 && B <= 64 && n_per_image > 0);
  hipLaunchKernelGGL(disc_accuracy_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, prob_fake, prob_real, B,
                     n_per_image, out);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}
