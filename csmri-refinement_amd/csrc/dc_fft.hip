// csmri_dc: k-space data consistency = batched 2-D complex FFT + mask merge + inverse FFT, fp32,
// interleaved complex (float2) layout.
//
//   out = orthoIFFT2( (1 - m) * orthoFFT2(x) + k0 )          myfft.py:131-163
//
// A 256x256 complex fp32 slice is 512 KiB and does not fit one CU's 160 KiB LDS, so the transform is three
// HBM passes (the intermediate stays in L2 / Infinity Cache between them):
//   pass 1  row FFTs: ONE WAVE PER ROW, the whole 1-D transform in registers -- no LDS, no barrier
//   pass 2  per strip of 8 columns (64-byte row segments, coalesced): column FFT -> ortho scale ->
//           (1-m)*k + k0 -> column inverse FFT, one wave per column out of an LDS image of the strip; the
//           merged k-space never touches HBM
//   pass 3  row inverse FFTs (wave per row) + ortho scale (+ the channel-padded copy that is the next conv
//           block's input)
// The passes run in place on `out`: algorithmic traffic 3 x B*H*W*8 B + mask, moved 7 x B*H*W*8 B.
//
// 1-D transform of N = L*R points in one wave (L = 64 lanes, R = N/64 values per lane; N = 32: two
// transforms per wave on its half-waves): lane l holds points l + L*q.  FORWARD = radix-2 decimation in
// frequency: the spans >= L pair REGISTERS of a lane, the spans < L pair LANES through wave shuffles
// (xor 32, 16, .. 1: ds_bpermute / DPP, no LDS memory).  Its output is bit-reversed: lane l then holds the R
// CONSECUTIVE frequencies R*rev(l) + rev(q), which it stores as one contiguous run -- HBM keeps the natural
// order and every 64-byte sector is written whole.  INVERSE = decimation in time, which consumes exactly that
// bit-reversed register arrangement (lane l loads its R consecutive inputs) and ends in natural order, lane-
// contiguous.  So the pair FFT -> iFFT needs no reordering anywhere, and in pass 2 the merge runs on the
// registers between the two.  Twiddles: a 512-entry table of exp(-2 pi i k / 512) built at COMPILE time
// (constexpr, double precision, rounded once), each lane keeps its <= 8 factors in registers.
#include "common.h"

// ---- compile-time twiddle table ------------------------------------------------------------------------
struct cf2 { float x, y; };
constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double cx_sin_small(double x) {     // |x| <= pi/4: Taylor, error < 1e-17
  double x2 = x * x, term = x, sum = x;
  for (int k = 1; k < 12; ++k) { term *= -x2 / ((2 * k) * (2 * k + 1)); sum += term; }
  return sum;
}
constexpr double cx_cos_small(double x) {
  double x2 = x * x, term = 1.0, sum = 1.0;
  for (int k = 1; k < 12; ++k) { term *= -x2 / ((2 * k - 1) * (2 * k)); sum += term; }
  return sum;
}
constexpr double cx_cos_turn(int k, int n) {   // cos(2 pi k / n), octant reduction keeps the argument <= pi/4
  k = ((k % n) + n) % n;
  if (8 * k <= n) return cx_cos_small(2.0 * kPi * k / n);
  if (8 * k <= 3 * n) return cx_sin_small(kPi / 2 - 2.0 * kPi * k / n);
  if (8 * k <= 5 * n) return -cx_cos_small(kPi - 2.0 * kPi * k / n);
  if (8 * k <= 7 * n) return cx_sin_small(2.0 * kPi * k / n - 3 * kPi / 2);
  return cx_cos_small(2.0 * kPi - 2.0 * kPi * k / n);
}
constexpr double cx_sin_turn(int k, int n) { return cx_cos_turn(4 * k - n, 4 * n); }   // sin(t) = cos(t - pi/2)
struct TwTable { cf2 v[512]; };
constexpr TwTable make_tw() {
  TwTable t{};
  for (int k = 0; k < 512; ++k) { t.v[k].x = (float)cx_cos_turn(k, 512); t.v[k].y = (float)(-cx_sin_turn(k, 512)); }
  return t;
}
__device__ const TwTable g_tw512 = make_tw();      // exp(-2 pi i k / 512)

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) {      // a * conj(b)
  return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
__device__ __forceinline__ float2 tw512(int idx) { const cf2 t = g_tw512.v[idx & 511]; return make_float2(t.x, t.y); }

template <int LOGN> struct FftCfg {
  static constexpr int N = 1 << LOGN, LOGL = LOGN < 6 ? LOGN : 6, L = 1 << LOGL, R = N / L, LOGR = LOGN - LOGL;
  static constexpr int TPW = 64 / L;             // transforms per wave
};

// the lane's twiddle factors: reg[t] for the register-level span N >> t, lane[s] for the lane-level span L >> s
template <int LOGN> struct LaneTw {
  float2 reg[FftCfg<LOGN>::LOGR > 0 ? FftCfg<LOGN>::LOGR : 1];
  float2 lane[FftCfg<LOGN>::LOGL];
  // lane-level stages are written branch-free: the pair's LOWER lane (bit `half` of the lane index clear) uses the
  // twiddle 1 and the sign +1, the UPPER lane its factor and the sign -1 -- a divergent if/else costs both sides
  // plus the exec-mask bookkeeping (41 branches per transform in the first version)
  float sgn[FftCfg<LOGN>::LOGL];
  __device__ __forceinline__ void init(int l) {
    typedef FftCfg<LOGN> C;
#pragma unroll
    for (int t = 0; t < C::LOGR; ++t) reg[t] = tw512((l << t) * (512 / C::N));            // w_{N >> t}^l
#pragma unroll
    for (int s = 0; s < C::LOGL; ++s) {
      const int half = C::L >> (s + 1);
      const bool upper = (l & half) != 0;
      const float2 w = tw512((l & (half - 1)) * (256 / (half > 0 ? half : 1)));              // w_{2 half}^(l mod half)
      lane[s] = upper ? w : make_float2(1.f, 0.f);
      sgn[s] = upper ? -1.f : 1.f;
    }
  }
};

__device__ __forceinline__ int rev_bits(int v, int bits) { return (int)(__brev((unsigned)v) >> (32 - bits)); }
template <int LOGR> __host__ __device__ constexpr int rev_const(int q) {
  int r = 0;
  for (int i = 0; i < LOGR; ++i) r |= ((q >> i) & 1) << (LOGR - 1 - i);
  return r;
}
// exp(-2 pi i j / h2) for the register-level stages (compile-time after unrolling)
__device__ __forceinline__ float2 reg_const_tw(int j, int h2) { return tw512(j * (512 / h2)); }

// forward (decimation in frequency), natural input v[q] = x[l + L q] -> bit-reversed output
template <int LOGN>
__device__ __forceinline__ void fft_dif(float2 (&v)[FftCfg<LOGN>::R], const LaneTw<LOGN>& tw, int lane) {
  typedef FftCfg<LOGN> C;
#pragma unroll
  for (int t = 0; t < C::LOGR; ++t) {
    const int h = C::R >> (t + 1);                 // register distance of the pair
#pragma unroll
    for (int q = 0; q < C::R; ++q) {
      if (q & h) continue;
      const float2 a = v[q], b = v[q + h];
      v[q] = make_float2(a.x + b.x, a.y + b.y);
      float2 d = make_float2(a.x - b.x, a.y - b.y);
      d = cmul(d, tw.reg[t]);
      const int j = q & (h - 1);
      if (j) d = cmul(d, reg_const_tw(j, 2 * h));
      v[q + h] = d;
    }
  }
#pragma unroll
  for (int s = 0; s < C::LOGL; ++s) {
    const int half = C::L >> (s + 1);
    float px[C::R], py[C::R];
#pragma unroll
    for (int q = 0; q < C::R; ++q) { px[q] = __shfl_xor(v[q].x, half); py[q] = __shfl_xor(v[q].y, half); }
#pragma unroll
    for (int q = 0; q < C::R; ++q) {
      // lower lane: v + p; upper lane: (p - v) * w
      const float2 d = make_float2(fmaf(tw.sgn[s], v[q].x, px[q]), fmaf(tw.sgn[s], v[q].y, py[q]));
      v[q] = half > 1 ? cmul(d, tw.lane[s]) : d;
    }
  }
}

// inverse (decimation in time, conjugated twiddles), bit-reversed input -> natural output v[q] = y[l + L q]
template <int LOGN>
__device__ __forceinline__ void fft_dit_inv(float2 (&v)[FftCfg<LOGN>::R], const LaneTw<LOGN>& tw, int lane) {
  typedef FftCfg<LOGN> C;
#pragma unroll
  for (int s = C::LOGL - 1; s >= 0; --s) {
    const int half = C::L >> (s + 1);
    // both lanes of a pair need conj(w) * b, b = the upper lane's value: the upper lane forms it (the lower
    // lane's factor is 1), both exchange, then lower = a + w'b, upper = a - w'b = p - (own w'b)
    float2 wb[C::R];
    float px[C::R], py[C::R];
#pragma unroll
    for (int q = 0; q < C::R; ++q) {
      wb[q] = half > 1 ? cmulc(v[q], tw.lane[s]) : v[q];
      px[q] = __shfl_xor(wb[q].x, half); py[q] = __shfl_xor(wb[q].y, half);
    }
#pragma unroll
    for (int q = 0; q < C::R; ++q) v[q] = make_float2(fmaf(tw.sgn[s], wb[q].x, px[q]), fmaf(tw.sgn[s], wb[q].y, py[q]));
  }
#pragma unroll
  for (int t = C::LOGR - 1; t >= 0; --t) {
    const int h = C::R >> (t + 1);
#pragma unroll
    for (int q = 0; q < C::R; ++q) {
      if (q & h) continue;
      float2 b = cmulc(v[q + h], tw.reg[t]);
      const int j = q & (h - 1);
      if (j) b = cmulc(b, reg_const_tw(j, 2 * h));
      const float2 a = v[q];
      v[q] = make_float2(a.x + b.x, a.y + b.y);
      v[q + h] = make_float2(a.x - b.x, a.y - b.y);
    }
  }
}

// Complex element storage of the image-side buffers: interleaved fp32 (float2) or interleaved bf16 (4 bytes per
// complex value -- the "bf16 cFFT" of BASELINE config 5: arithmetic stays fp32 in registers / LDS, only the HBM
// images and the intermediate between the passes are rounded; k-space data k0 / kout stays fp32).
#define DC_IO_BF16_PADDED 4           // bf16 pixels of >= 4 channels: value = channels (0,1) + channels (2,3)
template <int IO> __device__ __forceinline__ float2 ldc(const void* p, size_t scalar_idx) {
  if (IO == CSMRI_F32) return *(const float2*)((const float*)p + scalar_idx);
  if (IO == DC_IO_BF16_PADDED) {      // a CSMRI_BF16_SPLIT gradient in full; a plain padded one unchanged (zeros there)
    const u32x2_t u = *(const u32x2_t*)((const unsigned short*)p + scalar_idx);
    return make_float2(__uint_as_float(u[0] << 16) + __uint_as_float(u[1] << 16),
                       __uint_as_float(u[0] & 0xffff0000u) + __uint_as_float(u[1] & 0xffff0000u));
  }
  const unsigned u = *(const unsigned*)((const unsigned short*)p + scalar_idx);
  return make_float2(__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u));
}
template <int IO> __device__ __forceinline__ void stc(void* p, size_t complex_idx, float2 v) {
  if (IO == CSMRI_F32) ((float2*)p)[complex_idx] = v;
  else ((unsigned*)p)[complex_idx] = (unsigned)f32_to_bf16_bits(v.x) | ((unsigned)f32_to_bf16_bits(v.y) << 16);
}

#define DC_THREADS 256               // row passes: 4 waves = 4 rows per workgroup
#define DC_CTHREADS 512              // column pass: 8 waves = one column of the strip each
// columns per workgroup in pass 2 (template parameter STRIP): 8 = 64-byte row segments, 16 = whole 128-byte lines (every
// fetched line used once: 8 fetched 134.5 MB for 71.4 MB of inputs, r04 PMC).  Same-box A/B (profiles/r05_dc_strip_ab.log):
// 16 is 3 % faster at 64 x 256^2 and 15-30 % SLOWER at 8 x 256^2, 2 x 512^2 and 16 x 512^2 (half the workgroups, one per CU
// at 512 rows): the pass is bound by latency and occupancy, not by the fetched bytes.  dc_strip() picks.
#define DC_PITCH (STRIP + 1)         // float2 pitch of the LDS strip image (conflict-free column reads)

// passes 1 and 3 (and the row halves of csmri_fft2): 1-D transforms along W, one per wave (two for W = 32).
//   INV = false: src natural order with pixel stride src_ps floats -> dst natural order (dense)
//   INV = true : src dense natural order -> dst dense natural order (+ optional channel-padded copy), x scale
template <int LOGN, bool INV, int IO, int IO_IN = IO>
__global__ __launch_bounds__(DC_THREADS) void dc_rows_kernel(const void* __restrict__ src, int src_ps,
                                                             void* __restrict__ dst, void* out_pad, int out_pad_dt,
                                                             int rows, float scale) {
  typedef FftCfg<LOGN> C;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l = lane & (C::L - 1);
  int row = (blockIdx.x * (DC_THREADS / 64) + wave) * C::TPW + (lane >> C::LOGL);
  const bool live = row < rows;                       // shuffles need every lane: dead rows compute on row 0
  if (!live) row = 0;
  LaneTw<LOGN> tw;
  tw.init(l);
  float2 v[C::R];
  const size_t base = (size_t)row * C::N;
  const int kbase = C::R * rev_bits(l, C::LOGL);      // first of the lane's R consecutive frequencies
  if (!INV) {
#pragma unroll
    for (int q = 0; q < C::R; ++q) v[q] = ldc<IO_IN>(src, (base + l + C::L * q) * (size_t)src_ps);
    fft_dif<LOGN>(v, tw, lane);
    if (!live) return;
#pragma unroll
    for (int q = 0; q < C::R; ++q) {
      float2 o = v[q];
      o.x *= scale; o.y *= scale;
      stc<IO>(dst, base + kbase + rev_const<C::LOGR>(q), o);
    }
  } else {
#pragma unroll
    for (int q = 0; q < C::R; ++q) v[q] = ldc<IO>(src, (base + kbase + rev_const<C::LOGR>(q)) * (size_t)src_ps);
    fft_dit_inv<LOGN>(v, tw, lane);
    if (!live) return;
#pragma unroll
    for (int q = 0; q < C::R; ++q) {
      float2 o = v[q];
      o.x *= scale; o.y *= scale;
      const size_t p = base + l + C::L * q;
      stc<IO>(dst, p, o);
      if (out_pad) {
        if (out_pad_dt == CSMRI_F32) {
          f32x4_t* pp = (f32x4_t*)out_pad + p * 2;
          pp[0] = (f32x4_t){o.x, o.y, 0.f, 0.f}; pp[1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        } else {
          const unsigned hx = f32_to_bf16_bits(o.x), hy = f32_to_bf16_bits(o.y);
          unsigned lo = 0u;
          if (out_pad_dt == CSMRI_BF16_SPLIT)        // channels 2,3: what the rounding of channels 0,1 dropped
            lo = (unsigned)f32_to_bf16_bits(o.x - bf16_bits_to_f32((unsigned short)hx)) |
                 ((unsigned)f32_to_bf16_bits(o.y - bf16_bits_to_f32((unsigned short)hy)) << 16);
          ((u32x4_t*)out_pad)[p] = (u32x4_t){hx | (hy << 16), lo, 0u, 0u};
        }
      }
    }
  }
}

// pass 2: a strip of STRIP columns of one image.
//   MODE 0  FFT along H -> x scale -> mask merge (+ k0) -> inverse FFT along H   (csmri_dc / csmri_undersample)
//   MODE 1  FFT along H only (x scale), natural order out                        (csmri_fft2 forward)
//   MODE 2  inverse FFT along H only (x scale)                                   (csmri_fft2 inverse)
template <int LOGN, int MODE, int IO, int STRIP>
__global__ __launch_bounds__(DC_CTHREADS) void dc_cols_kernel(void* __restrict__ data, const float2* __restrict__ k0,
                                                             const uint8_t* __restrict__ mask, int W, float scale,
                                                             float2* __restrict__ kout, int keep_sampled) {
  typedef FftCfg<LOGN> C;
  constexpr int H = C::N;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float2* A = (float2*)smem;                       // [H][DC_PITCH]
  float2* K = A + H * DC_PITCH;                    // [H][DC_PITCH]  k0 strip, then the merged k-space (kout)
  uint8_t* M = (uint8_t*)(K + H * DC_PITCH);       // [H][STRIP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int strips = W / STRIP;
  const int img = blockIdx.x / strips, c0 = (blockIdx.x - img * strips) * STRIP;
  const size_t base = (size_t)img * H * W + c0;
  for (int idx = tid; idx < H * STRIP; idx += DC_CTHREADS) {
    const int h = idx / STRIP, c = idx - h * STRIP;
    const size_t o = base + (size_t)h * W + c;
    A[h * DC_PITCH + c] = ldc<IO>(data, 2 * o);
    if (MODE == 0) {
      if (k0) K[h * DC_PITCH + c] = k0[o];
      M[h * STRIP + c] = mask[o];
    }
  }
  __syncthreads();
  const int l = lane & (C::L - 1);
  LaneTw<LOGN> tw;
  tw.init(l);
  const int kbase = C::R * rev_bits(l, C::LOGL);
  // a wave transforms C::TPW columns at a time (1; H = 32: two, on its half-waves): the strip takes PASSES rounds of the
  // workgroup's 8 waves
  constexpr int WAVES = DC_CTHREADS / 64, PASSES = (STRIP + WAVES * C::TPW - 1) / (WAVES * C::TPW);
#pragma unroll 1
  for (int ps = 0; ps < PASSES; ++ps) {
    const int c = (ps * WAVES + wave) * C::TPW + (lane >> C::LOGL);
    if ((ps * WAVES + wave) * C::TPW >= STRIP) break;          // wave-uniform
    float2 v[C::R];
    if (MODE != 2) {
#pragma unroll
      for (int q = 0; q < C::R; ++q) v[q] = A[(l + C::L * q) * DC_PITCH + c];
      fft_dif<LOGN>(v, tw, lane);
    }
    if (MODE == 0) {
#pragma unroll
      for (int q = 0; q < C::R; ++q) {
        const int ky = kbase + rev_const<C::LOGR>(q);
        float2 k = v[q];
        k.x *= scale; k.y *= scale;
        // (1 - m) * k + k0 with m in {0,1}: bit-exact integer mask test; keep_sampled (forward model): m * k
        float2 o = (M[ky * STRIP + c] != 0) == (keep_sampled != 0) ? k : make_float2(0.f, 0.f);
        if (k0) { const float2 z = K[ky * DC_PITCH + c]; o.x += z.x; o.y += z.y; }
        if (kout) K[ky * DC_PITCH + c] = o;
        v[q] = o;
      }
    } else if (MODE == 1) {
#pragma unroll
      for (int q = 0; q < C::R; ++q)
        A[(kbase + rev_const<C::LOGR>(q)) * DC_PITCH + c] = make_float2(v[q].x * scale, v[q].y * scale);
    } else {
#pragma unroll
      for (int q = 0; q < C::R; ++q) v[q] = A[(kbase + rev_const<C::LOGR>(q)) * DC_PITCH + c];
    }
    if (MODE != 1) {
      fft_dit_inv<LOGN>(v, tw, lane);
      const float s2 = MODE == 2 ? scale : 1.f;
#pragma unroll
      for (int q = 0; q < C::R; ++q) A[(l + C::L * q) * DC_PITCH + c] = make_float2(v[q].x * s2, v[q].y * s2);
    }
  }
  __syncthreads();
  for (int idx = tid; idx < H * STRIP; idx += DC_CTHREADS) {
    const int h = idx / STRIP, c = idx - h * STRIP;
    const size_t o = base + (size_t)h * W + c;
    stc<IO>(data, o, A[h * DC_PITCH + c]);
    if (MODE == 0 && kout) kout[o] = K[h * DC_PITCH + c];
  }
}

static int log2_in_range(int n) {
  for (int s = 5; s <= 9; ++s) if (n == (1 << s)) return s;
  return -1;
}

template <int LOGN, bool INV, int IO, int IO_IN = IO>
static int launch_rows(const void* src, int src_ps, void* dst, void* out_pad, int out_pad_dt, int rows,
                       float scale, hipStream_t st) {
  constexpr int per_wg = (DC_THREADS / 64) * FftCfg<LOGN>::TPW;
  hipLaunchKernelGGL((dc_rows_kernel<LOGN, INV, IO, IO_IN>), dim3(cdiv(rows, per_wg)), dim3(DC_THREADS), 0, st,
                     src, src_ps, dst, out_pad, out_pad_dt, rows, scale);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}
template <int IO, int IO_IN = IO>
static int rows_pass(int logw, bool inv, const void* src, int src_ps, void* dst, void* out_pad, int out_pad_dt,
                     int rows, float scale, hipStream_t st) {
#define ROWS(LW) (inv ? launch_rows<LW, true, IO>(src, src_ps, dst, out_pad, out_pad_dt, rows, scale, st) \
                      : launch_rows<LW, false, IO, IO_IN>(src, src_ps, dst, out_pad, out_pad_dt, rows, scale, st))
  switch (logw) {
    case 5: return ROWS(5); case 6: return ROWS(6); case 7: return ROWS(7); case 8: return ROWS(8); case 9: return ROWS(9);
  }
#undef ROWS
  return CSMRI_E_UNSUPPORTED;
}

template <int LOGN, int MODE, int IO, int STRIP>
static int launch_cols_s(void* data, const float2* k0, const uint8_t* mask, int B, int W, float scale,
                         float2* kout, int keep, hipStream_t st) {
  constexpr int H = 1 << LOGN;
  constexpr int lds = 2 * H * DC_PITCH * (int)sizeof(float2) + H * STRIP;
  CSMRI_SET_MAX_LDS((dc_cols_kernel<LOGN, MODE, IO, STRIP>), lds);
  hipLaunchKernelGGL((dc_cols_kernel<LOGN, MODE, IO, STRIP>), dim3(B * (W / STRIP)), dim3(DC_CTHREADS), lds, st,
                     data, k0, mask, W, scale, kout, keep);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}
// 16-column strips where they still give >= 4 workgroups per CU and two of them fit a CU's LDS (<= 256 rows)
static int dc_strip(int B, int H, int W) { return (H <= 256 && (long long)B * (W / 16) >= 1024) ? 16 : 8; }
template <int LOGN, int MODE, int IO>
static int launch_cols(void* data, const float2* k0, const uint8_t* mask, int B, int W, float scale,
                       float2* kout, int keep, hipStream_t st) {
  if (dc_strip(B, 1 << LOGN, W) == 16) return launch_cols_s<LOGN, MODE, IO, 16>(data, k0, mask, B, W, scale, kout, keep, st);
  return launch_cols_s<LOGN, MODE, IO, 8>(data, k0, mask, B, W, scale, kout, keep, st);
}
template <int MODE, int IO>
static int cols_pass(int logh, void* data, const float2* k0, const uint8_t* mask, int B, int W, float scale,
                     float2* kout, int keep, hipStream_t st) {
  switch (logh) {
    case 5: return launch_cols<5, MODE, IO>(data, k0, mask, B, W, scale, kout, keep, st);
    case 6: return launch_cols<6, MODE, IO>(data, k0, mask, B, W, scale, kout, keep, st);
    case 7: return launch_cols<7, MODE, IO>(data, k0, mask, B, W, scale, kout, keep, st);
    case 8: return launch_cols<8, MODE, IO>(data, k0, mask, B, W, scale, kout, keep, st);
    case 9: return launch_cols<9, MODE, IO>(data, k0, mask, B, W, scale, kout, keep, st);
  }
  return CSMRI_E_UNSUPPORTED;
}

extern "C" size_t csmri_dc_work_bytes(int B, int H, int W) {
  (void)B; (void)H; (void)W;
  return 0;  // the three passes run in place on `out`
}

template <int IO, int IO_IN = IO>
static int dc_passes(const void* x, int x_pix_stride, const float* k0, const uint8_t* mask, void* out, void* out_pad,
                     int out_pad_dtype, int B, int H, int W, hipStream_t st) {
  CSMRI_CHECK_ARG(x && mask && out && B > 0 && x_pix_stride >= 2 && x_pix_stride % 2 == 0);
  const int lh = log2_in_range(H), lw = log2_in_range(W);
  if (lh < 0 || lw < 0) return CSMRI_E_UNSUPPORTED;
  if (((uintptr_t)x | (uintptr_t)out | (uintptr_t)k0 | (uintptr_t)out_pad) & 15) return CSMRI_E_ALIGN;
  const float scale = 1.0f / sqrtf((float)H * (float)W);
  int rc = rows_pass<IO, IO_IN>(lw, false, x, x_pix_stride, out, nullptr, 0, B * H, 1.0f, st);
  if (rc != CSMRI_OK) return rc;
  rc = cols_pass<0, IO>(lh, out, (const float2*)k0, mask, B, W, scale, nullptr, 0, st);
  if (rc != CSMRI_OK) return rc;
  return rows_pass<IO>(lw, true, out, 2, out, out_pad, out_pad_dtype, B * H, scale, st);
}
extern "C" int csmri_dc(const float* x, int x_pix_stride, const float* k0, const uint8_t* mask,
                        float* out, void* out_pad, int out_pad_dtype, float* work, int B, int H,
                        int W, void* stream) {
  (void)work;
  return dc_passes<CSMRI_F32>(x, x_pix_stride, k0, mask, out, out_pad, out_pad_dtype, B, H, W, (hipStream_t)stream);
}
// bf16 image storage ("bf16 cFFT"): x, out and the intermediate between the passes are interleaved bf16
extern "C" int csmri_dc_bf16(const void* x, int x_pix_stride, const float* k0, const uint8_t* mask, void* out,
                             void* out_pad, int out_pad_dtype, int B, int H, int W, void* stream) {
  return dc_passes<CSMRI_BF16>(x, x_pix_stride, k0, mask, out, out_pad, out_pad_dtype, B, H, W, (hipStream_t)stream);
}

// fp32 arithmetic and output, the input image read as bf16 (channels 0,1 of a bf16 tensor with pixel stride
// x_pix_stride): the DC adjoint applied directly to the channel-padded bf16 gradient a convolution's
// data-gradient kernel wrote, without a conversion pass in between
extern "C" int csmri_dc_in_bf16(const void* x, int x_dtype, int x_pix_stride, const float* k0, const uint8_t* mask,
                                float* out, void* out_pad, int out_pad_dtype, int B, int H, int W, void* stream) {
  CSMRI_CHECK_ARG(x_dtype == CSMRI_BF16 || x_dtype == CSMRI_BF16_SPLIT);
  // the caller DECLARES the format: CSMRI_BF16_SPLIT reads channels (0,1) + (2,3) of a channel-padded pixel, CSMRI_BF16
  // channels 0,1 alone whatever the other channels hold
  if (x_dtype == CSMRI_BF16_SPLIT) {
    CSMRI_CHECK_ARG(x_pix_stride >= 4 && x_pix_stride % 4 == 0);
    return dc_passes<CSMRI_F32, DC_IO_BF16_PADDED>(x, x_pix_stride, k0, mask, out, out_pad, out_pad_dtype, B, H, W,
                                                    (hipStream_t)stream);
  }
  return dc_passes<CSMRI_F32, CSMRI_BF16>(x, x_pix_stride, k0, mask, out, out_pad, out_pad_dtype, B, H, W,
                                           (hipStream_t)stream);
}

// The forward model that produces a training sample from a (complex) image, on the device:
//   kspace = m * orthoFFT2(img),   inp = orthoIFFT2(kspace)
// (data/reconstruction/rec_transforms.py:18-57 -> compressed_sensing.py:460-512: numpy complex128
// on the host in the reference).  Same three passes as csmri_dc with the merge replaced by the
// mask product and the k-space strip written out on the way.
extern "C" int csmri_undersample(const float* img, const uint8_t* mask, float* kspace, float* inp, int B, int H,
                                 int W, void* stream) {
  CSMRI_CHECK_ARG(img && mask && kspace && inp && B > 0);
  const int lh = log2_in_range(H), lw = log2_in_range(W);
  if (lh < 0 || lw < 0) return CSMRI_E_UNSUPPORTED;
  if (((uintptr_t)img | (uintptr_t)kspace | (uintptr_t)inp) & 15) return CSMRI_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const float scale = 1.0f / sqrtf((float)H * (float)W);
  int rc = rows_pass<CSMRI_F32>(lw, false, img, 2, inp, nullptr, 0, B * H, 1.0f, st);
  if (rc != CSMRI_OK) return rc;
  rc = cols_pass<0, CSMRI_F32>(lh, inp, nullptr, mask, B, W, scale, (float2*)kspace, 1, st);
  if (rc != CSMRI_OK) return rc;
  return rows_pass<CSMRI_F32>(lw, true, inp, 2, inp, nullptr, 0, B * H, scale, st);
}

// Stand-alone batched 2-D FFT / inverse FFT of interleaved complex fp32 images: the operation behind the
// reference's Fft2d / Ifft2d Functions (myfft.py:78-128), whose backward passes are the same transforms in
// the other direction (myfft.py:92-102,119-128).  ortho != 0: both directions scaled by 1/sqrt(HW) (the
// reference's normalized=True); ortho == 0: forward unscaled, inverse scaled by 1/(HW) (pytorch_fft).
template <int IO>
static int fft2_passes(const void* x, void* out, int B, int H, int W, int inverse, int ortho, hipStream_t st) {
  CSMRI_CHECK_ARG(x && out && B > 0);
  const int lh = log2_in_range(H), lw = log2_in_range(W);
  if (lh < 0 || lw < 0) return CSMRI_E_UNSUPPORTED;
  if (((uintptr_t)x | (uintptr_t)out) & 15) return CSMRI_E_ALIGN;
  const float scale = ortho ? 1.0f / sqrtf((float)H * (float)W) : (inverse ? 1.0f / ((float)H * (float)W) : 1.0f);
  int rc = rows_pass<IO>(lw, inverse != 0, x, 2, out, nullptr, 0, B * H, 1.0f, st);
  if (rc != CSMRI_OK) return rc;
  return inverse ? cols_pass<2, IO>(lh, out, nullptr, nullptr, B, W, scale, nullptr, 0, st)
                 : cols_pass<1, IO>(lh, out, nullptr, nullptr, B, W, scale, nullptr, 0, st);
}
extern "C" int csmri_fft2(const float* x, float* out, int B, int H, int W, int inverse, int ortho, void* stream) {
  return fft2_passes<CSMRI_F32>(x, out, B, H, W, inverse, ortho, (hipStream_t)stream);
}
extern "C" int csmri_fft2_bf16(const void* x, void* out, int B, int H, int W, int inverse, int ortho, void* stream) {
  return fft2_passes<CSMRI_BF16>(x, out, B, H, W, inverse, ortho, (hipStream_t)stream);
}
