// csmri_dc: k-space data consistency = batched 2-D complex FFT + mask merge +
// inverse FFT, fp32, interleaved complex (float2) layout.
//
//   out = orthoIFFT2( (1 - m) * orthoFFT2(x) + k0 )          myfft.py:131-163
//
// A 256x256 complex fp32 slice is 512 KiB and does not fit one CU's 160 KiB LDS,
// so the transform is decomposed into three HBM passes, each of which keeps a
// tile of 16 independent 1-D transforms in LDS:
//   pass 1  row FFTs                 (16 rows per workgroup, coalesced 2 KiB rows)
//   pass 2  per 16-column strip: column FFT -> ortho scale -> (1-m)*k + k0 ->
//           column inverse FFT; the merged k-space never touches HBM
//   pass 3  row inverse FFTs + ortho scale (+ optional channel-padded copy that
//           is the next conv block's input)
// The passes run in place on `out`, so the traffic is 7 x B*H*W*8 bytes
// (x, k0 reads; mask bytes; 3 writes + 2 re-reads of the intermediate).
// 1-D transform: Stockham autosort, radix-4 stages (+ one radix-2 stage for odd
// log2 N), twiddles from an LDS table built with sincospif at kernel start.
#include "common.h"

#define DC_T 16           // independent transforms per tile
#define DC_TP (DC_T + 1)  // LDS pitch (bank-conflict padding)
#define DC_THREADS 256

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// In-LDS FFT of DC_T interleaved transforms of length N.  Data layout a[n*DC_TP + t].
// sign = -1 forward, +1 inverse (unnormalised).  Returns the buffer holding the result.
__device__ float2* fft_tile(float2* a, float2* b, const float2* tw, int N, int sign, int tid) {
  int Ns = 1;
  while (Ns < N) {
    const int rem = N / Ns;
    if ((rem & 3) == 0) {
      const int nb = N >> 2, tstep = N / (Ns * 4);
      for (int idx = tid; idx < nb * DC_T; idx += DC_THREADS) {
        const int col = idx % DC_T, j = idx / DC_T;
        const int k = j & (Ns - 1);
        float2 v0 = a[j * DC_TP + col];
        float2 v1 = a[(j + nb) * DC_TP + col];
        float2 v2 = a[(j + 2 * nb) * DC_TP + col];
        float2 v3 = a[(j + 3 * nb) * DC_TP + col];
        if (k) {
          float2 w1 = tw[k * tstep], w2 = tw[2 * k * tstep], w3 = tw[3 * k * tstep];
          if (sign > 0) { w1.y = -w1.y; w2.y = -w2.y; w3.y = -w3.y; }
          v1 = cmul(v1, w1); v2 = cmul(v2, w2); v3 = cmul(v3, w3);
        }
        float2 t0 = make_float2(v0.x + v2.x, v0.y + v2.y);
        float2 t1 = make_float2(v0.x - v2.x, v0.y - v2.y);
        float2 t2 = make_float2(v1.x + v3.x, v1.y + v3.y);
        float2 d = make_float2(v1.x - v3.x, v1.y - v3.y);
        // (v1 - v3) * (sign * i)
        float2 t3 = sign < 0 ? make_float2(d.y, -d.x) : make_float2(-d.y, d.x);
        const int j0 = (j - k) * 4 + k;
        b[j0 * DC_TP + col] = make_float2(t0.x + t2.x, t0.y + t2.y);
        b[(j0 + Ns) * DC_TP + col] = make_float2(t1.x + t3.x, t1.y + t3.y);
        b[(j0 + 2 * Ns) * DC_TP + col] = make_float2(t0.x - t2.x, t0.y - t2.y);
        b[(j0 + 3 * Ns) * DC_TP + col] = make_float2(t1.x - t3.x, t1.y - t3.y);
      }
      Ns *= 4;
    } else {
      const int nb = N >> 1, tstep = N / (Ns * 2);
      for (int idx = tid; idx < nb * DC_T; idx += DC_THREADS) {
        const int col = idx % DC_T, j = idx / DC_T;
        const int k = j & (Ns - 1);
        float2 v0 = a[j * DC_TP + col];
        float2 v1 = a[(j + nb) * DC_TP + col];
        if (k) {
          float2 w1 = tw[k * tstep];
          if (sign > 0) w1.y = -w1.y;
          v1 = cmul(v1, w1);
        }
        const int j0 = (j - k) * 2 + k;
        b[j0 * DC_TP + col] = make_float2(v0.x + v1.x, v0.y + v1.y);
        b[(j0 + Ns) * DC_TP + col] = make_float2(v0.x - v1.x, v0.y - v1.y);
      }
      Ns *= 2;
    }
    __syncthreads();
    float2* t = a; a = b; b = t;
  }
  return a;
}

__device__ __forceinline__ void build_twiddles(float2* tw, int N, int tid) {
  for (int i = tid; i < N; i += DC_THREADS) {
    float s, c;
    sincospif(2.0f * (float)i / (float)N, &s, &c);
    tw[i] = make_float2(c, -s);     // exp(-2 pi i k / N)
  }
}

// passes 1 and 3: FFT along W for DC_T rows per workgroup.
__global__ __launch_bounds__(DC_THREADS) void dc_rows_kernel(
    const float* src, int src_ps, float2* dst, void* out_pad, int out_pad_dt,
    int W, int sign, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float2* a = (float2*)smem;
  float2* b = a + W * DC_TP;
  float2* tw = b + W * DC_TP;
  const int tid = threadIdx.x;
  const size_t row0 = (size_t)blockIdx.x * DC_T;
  build_twiddles(tw, W, tid);
  for (int idx = tid; idx < DC_T * W; idx += DC_THREADS) {
    const int r = idx / W, n = idx - r * W;
    a[n * DC_TP + r] = *(const float2*)(src + ((row0 + r) * W + n) * (size_t)src_ps);
  }
  __syncthreads();
  float2* res = fft_tile(a, b, tw, W, sign, tid);
  for (int idx = tid; idx < DC_T * W; idx += DC_THREADS) {
    const int r = idx / W, n = idx - r * W;
    float2 v = res[n * DC_TP + r];
    v.x *= scale; v.y *= scale;
    const size_t o = (row0 + r) * W + n;
    dst[o] = v;
    if (out_pad) {
      if (out_pad_dt == CSMRI_F32) {
        f32x4_t lo = (f32x4_t){v.x, v.y, 0.f, 0.f}, z = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        f32x4_t* pp = (f32x4_t*)out_pad + o * 2;
        pp[0] = lo; pp[1] = z;
      } else {
        u32x4_t q = (u32x4_t){(unsigned)f32_to_bf16_bits(v.x) | ((unsigned)f32_to_bf16_bits(v.y) << 16), 0u, 0u, 0u};
        ((u32x4_t*)out_pad)[o] = q;
      }
    }
  }
}

// pass 2: per strip of DC_T columns: FFT along H, merge, inverse FFT along H.
__global__ __launch_bounds__(DC_THREADS) void dc_cols_kernel(
    float2* __restrict__ data, const float2* __restrict__ k0, const uint8_t* __restrict__ mask,
    int H, int W, float scale, float2* __restrict__ kout, int keep_sampled) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float2* a = (float2*)smem;
  float2* b = a + H * DC_TP;
  float2* tw = b + H * DC_TP;
  const int tid = threadIdx.x;
  const int strips = W / DC_T;
  const int img = blockIdx.x / strips, c0 = (blockIdx.x - img * strips) * DC_T;
  const size_t base = (size_t)img * H * W + c0;
  build_twiddles(tw, H, tid);
  for (int idx = tid; idx < DC_T * H; idx += DC_THREADS) {
    const int h = idx / DC_T, c = idx - h * DC_T;
    a[h * DC_TP + c] = data[base + (size_t)h * W + c];
  }
  __syncthreads();
  float2* res = fft_tile(a, b, tw, H, -1, tid);
  for (int idx = tid; idx < DC_T * H; idx += DC_THREADS) {
    const int h = idx / DC_T, c = idx - h * DC_T;
    const size_t o = base + (size_t)h * W + c;
    float2 k = res[h * DC_TP + c];
    k.x *= scale; k.y *= scale;
    // (1 - m) * k + k0 with m in {0,1}: bit-exact integer mask test
    // keep_sampled (forward model, csmri_undersample): m * k instead, and the k-space is an output
    float2 v = (mask[o] != 0) == (keep_sampled != 0) ? k : make_float2(0.f, 0.f);
    if (k0) { float2 q = k0[o]; v.x += q.x; v.y += q.y; }
    if (kout) kout[o] = v;
    res[h * DC_TP + c] = v;
  }
  __syncthreads();
  float2* other = (res == a) ? b : a;
  float2* r2 = fft_tile(res, other, tw, H, +1, tid);
  for (int idx = tid; idx < DC_T * H; idx += DC_THREADS) {
    const int h = idx / DC_T, c = idx - h * DC_T;
    data[base + (size_t)h * W + c] = r2[h * DC_TP + c];
  }
}

static bool is_pow2_in_range(int n) { return n >= 32 && n <= 512 && (n & (n - 1)) == 0; }

extern "C" size_t csmri_dc_work_bytes(int B, int H, int W) {
  (void)B; (void)H; (void)W;
  return 0;  // the three passes run in place on `out`
}

extern "C" int csmri_dc(const float* x, int x_pix_stride, const float* k0, const uint8_t* mask,
                        float* out, void* out_pad, int out_pad_dtype, float* work, int B, int H,
                        int W, void* stream) {
  (void)work;
  CSMRI_CHECK_ARG(x && mask && out && B > 0 && x_pix_stride >= 2 && x_pix_stride % 2 == 0);
  if (!is_pow2_in_range(H) || !is_pow2_in_range(W)) return CSMRI_E_UNSUPPORTED;
  if (((uintptr_t)x | (uintptr_t)out | (uintptr_t)k0 | (uintptr_t)out_pad) & 15) return CSMRI_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const float scale = 1.0f / sqrtf((float)H * (float)W);
  const int lds_rows = (2 * W * DC_TP + W) * (int)sizeof(float2);
  const int lds_cols = (2 * H * DC_TP + H) * (int)sizeof(float2);
  CSMRI_SET_MAX_LDS(dc_rows_kernel, lds_rows);
  CSMRI_SET_MAX_LDS(dc_cols_kernel, lds_cols);
  const int row_blocks = B * H / DC_T, col_blocks = B * (W / DC_T);
  hipLaunchKernelGGL(dc_rows_kernel, dim3(row_blocks), dim3(DC_THREADS), lds_rows, st,
                     x, x_pix_stride, (float2*)out, (void*)nullptr, 0, W, -1, 1.0f);
  CSMRI_LAUNCH_CHECK();
  hipLaunchKernelGGL(dc_cols_kernel, dim3(col_blocks), dim3(DC_THREADS), lds_cols, st,
                     (float2*)out, (const float2*)k0, mask, H, W, scale, (float2*)nullptr, 0);
  CSMRI_LAUNCH_CHECK();
  hipLaunchKernelGGL(dc_rows_kernel, dim3(row_blocks), dim3(DC_THREADS), lds_rows, st,
                     (const float*)out, 2, (float2*)out, out_pad, out_pad_dtype, W, +1, scale);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

// The forward model that produces a training sample from a (complex) image, on the device:
//   kspace = m * orthoFFT2(img),   inp = orthoIFFT2(kspace)
// (data/reconstruction/rec_transforms.py:18-57 -> compressed_sensing.py:460-512: numpy complex128
// on the host in the reference).  Same three passes as csmri_dc with the merge replaced by the
// mask product and the k-space strip written out on the way.
extern "C" int csmri_undersample(const float* img, const uint8_t* mask, float* kspace, float* inp, int B, int H,
                                 int W, void* stream) {
  CSMRI_CHECK_ARG(img && mask && kspace && inp && B > 0);
  if (!is_pow2_in_range(H) || !is_pow2_in_range(W)) return CSMRI_E_UNSUPPORTED;
  if (((uintptr_t)img | (uintptr_t)kspace | (uintptr_t)inp) & 15) return CSMRI_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const float scale = 1.0f / sqrtf((float)H * (float)W);
  const int lds_rows = (2 * W * DC_TP + W) * (int)sizeof(float2);
  const int lds_cols = (2 * H * DC_TP + H) * (int)sizeof(float2);
  CSMRI_SET_MAX_LDS(dc_rows_kernel, lds_rows);
  CSMRI_SET_MAX_LDS(dc_cols_kernel, lds_cols);
  const int row_blocks = B * H / DC_T, col_blocks = B * (W / DC_T);
  hipLaunchKernelGGL(dc_rows_kernel, dim3(row_blocks), dim3(DC_THREADS), lds_rows, st,
                     img, 2, (float2*)inp, (void*)nullptr, 0, W, -1, 1.0f);
  CSMRI_LAUNCH_CHECK();
  hipLaunchKernelGGL(dc_cols_kernel, dim3(col_blocks), dim3(DC_THREADS), lds_cols, st,
                     (float2*)inp, (const float2*)nullptr, mask, H, W, scale, (float2*)kspace, 1);
  CSMRI_LAUNCH_CHECK();
  hipLaunchKernelGGL(dc_rows_kernel, dim3(row_blocks), dim3(DC_THREADS), lds_rows, st,
                     (const float*)inp, 2, (float2*)inp, (void*)nullptr, 0, W, +1, scale);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}
