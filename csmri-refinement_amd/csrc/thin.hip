// Thin layers: convolutions with ONE real channel on one side -- the discriminator's first layer (1 -> 64,
// reference models/discriminators.py:137-150 on |x| of training/adversarial_training.py:33-40), its data gradient
// (64 -> 1), the U-Net head (32 -> 1, models/unet.py:247-250) and the discriminator's final conv (1024 -> 1,
// models/discriminators.py:160-172).  Their tensors carry the single channel padded to 8, so the MFMA kernels of
// gconv.hip spend an 8-wide K (or N) side on 7 zeros and, worse, a 16-byte gather per tap and position: they ran
// at 4-8 x the time the bytes need (41.7 + 33.5 us forward, 68.8 us data gradient at the bench shapes for
// 50 / 25 MB).  These layers are HBM-bound vector work:
//
//   thin_in1_kernel   Cin_real = 1: a thread owns (output position, 8 output channels); its KH*KW*8 weights sit in
//                     registers for the whole launch, the KH*KW input scalars of a position are 2-byte loads shared
//                     by the threads of that position (same address within a wave: one request), one 16-byte store
//                     per thread and position -> 128 contiguous bytes per position.
//   thin_out1_kernel  Cout_real = 1: G lanes (8, or a whole wave for deep-K / few positions) share one output
//                     position, each reads 16 bytes of every tap's channel run and of the weight row, 8 FMAs, and
//                     the lanes combine by wave shuffles (fixed order); lane 0 stores the value and 7 zeros.
//
// Both follow csmri_gconv_desc's addressing contract exactly (border rule, strides, parity classes, output window),
// so every epilogue consumer (csmri_fold_halo, ...) is unchanged.  fp32 accumulation, one rounding on output.
#include "gconv_params.h"

__device__ __forceinline__ void thin_decomp(const GParams& p, int m, int& b, int& oy, int& ox) {
  const int HoWo = p.Ho * p.Wo;
  if (p.howo_shift >= 0) { b = m >> p.howo_shift; const int r = m & (HoWo - 1); oy = r >> p.wo_shift; ox = r & (p.Wo - 1); }
  else { b = m / HoWo; const int r = m - b * HoWo; oy = r / p.Wo; ox = r - oy * p.Wo; }
}

__device__ __forceinline__ void thin_store8(char* base, size_t elem_off, int dt, const float (&v)[8]) {
  if (dt == CSMRI_BF16) {
    const u32x2_t lo = pack4_bf16((f32x4_t){v[0], v[1], v[2], v[3]}), hi = pack4_bf16((f32x4_t){v[4], v[5], v[6], v[7]});
    *(u32x4_t*)(base + elem_off * 2) = (u32x4_t){lo[0], lo[1], hi[0], hi[1]};
  } else {
    *(f32x4_t*)(base + elem_off * 4) = (f32x4_t){v[0], v[1], v[2], v[3]};
    *(f32x4_t*)(base + elem_off * 4 + 16) = (f32x4_t){v[4], v[5], v[6], v[7]};
  }
}

// ------------------------------------------------------------------------------------------------------------
template <int DT, int TH, int TW>
__global__ __launch_bounds__(256) void thin_in1_kernel(const GParams p) {
  constexpr int NT = TH * TW;
  typedef typename DTraits<DT>::T T;
  const int NC = p.Cout >> 3;                      // 16-byte output chunks per position (power of two <= 32)
  const int chunk = threadIdx.x & (NC - 1), pl = threadIdx.x / NC, lanes = 256 / NC;
  const int n0 = chunk * 8;
  float w[NT][8];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) w[t][j] = (float)((const T*)p.w)[(size_t)(n0 + j) * p.Kp + t * 8];
  float bias[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bias[j] = p.bias ? p.bias[n0 + j] : 0.f;
  const T* in = (const T*)p.in0;
  for (int m = blockIdx.x * lanes + pl; m < p.M; m += gridDim.x * lanes) {
    int b, oy, ox;
    thin_decomp(p, m, b, oy, ox);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = bias[j];
    const int ib = b * p.Hin * p.Win;
#pragma unroll
    for (int ty = 0; ty < TH; ++ty) {
      int u = oy * p.S + p.dy0 + ty * p.dys;
      bool oku = true;
      if (p.border == CSMRI_BORDER_REFLECT) u = reflect_idx(u, p.Hin); else oku = (unsigned)u < (unsigned)p.Hin;
#pragma unroll
      for (int tx = 0; tx < TW; ++tx) {
        int v = ox * p.S + p.dx0 + tx * p.dxs;
        bool ok = oku;
        if (p.border == CSMRI_BORDER_REFLECT) v = reflect_idx(v, p.Win); else ok = ok && (unsigned)v < (unsigned)p.Win;
        const float x = ok ? (float)in[(size_t)(ib + u * p.Win + v) * p.ps0] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += x * w[ty * TW + tx][j];
      }
    }
    if (p.slope != 1.f) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = acc[j] < 0.f ? acc[j] * p.slope : acc[j];
    }
    const OutPos op = gconv_out_pos(p, b, oy * p.osy + p.ooy, ox * p.osx + p.oox);
    thin_store8(op.base, op.opix + n0, p.out_dt, acc);
  }
}

int thin_in1_eligible(const csmri_gconv_desc* d) {
  if (d->dtype != CSMRI_BF16 && d->dtype != CSMRI_F32) return 0;
  if (d->Cin != 8 || d->cin_real != 1 || d->in1 || d->upsample || d->nclass > 1) return 0;
  if (d->stats_partial || d->splitk > 1 || d->g_src) return 0;
  if (d->TH != 4 || d->TW != 4) return 0;
  const int nc = d->Cout / 8;
  if (d->Cout % 8 || nc < 1 || nc > 32 || (nc & (nc - 1))) return 0;
  return 1;
}

int thin_in1_launch(const GParams& p, const csmri_gconv_desc* d, hipStream_t st) {
  const int lanes = 256 / (p.Cout / 8);
  long long blocks = ((long long)p.M + lanes - 1) / lanes;
  if (blocks > 2048) blocks = 2048;                // grid-stride: the weight registers are loaded once per thread
  if (d->dtype == CSMRI_BF16) hipLaunchKernelGGL((thin_in1_kernel<CSMRI_BF16, 4, 4>), dim3((int)blocks), dim3(256), 0, st, p);
  else hipLaunchKernelGGL((thin_in1_kernel<CSMRI_F32, 4, 4>), dim3((int)blocks), dim3(256), 0, st, p);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

// ------------------------------------------------------------------------------------------------------------
template <int DT> __device__ __forceinline__ float thin_dot8(const char* a, const char* b) {
  float s = 0.f;
  if constexpr (DT == CSMRI_BF16) {
    const u32x4_t x = *(const u32x4_t*)a, y = *(const u32x4_t*)b;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      s += __uint_as_float(x[q] << 16) * __uint_as_float(y[q] << 16);
      s += __uint_as_float(x[q] & 0xffff0000u) * __uint_as_float(y[q] & 0xffff0000u);
    }
  } else {
    const f32x4_t x0 = *(const f32x4_t*)a, x1 = *(const f32x4_t*)(a + 16), y0 = *(const f32x4_t*)b, y1 = *(const f32x4_t*)(b + 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) { s += x0[q] * y0[q]; s += x1[q] * y1[q]; }
  }
  return s;
}

template <int DT, int G>
__global__ __launch_bounds__(256) void thin_out1_kernel(const GParams p) {
  constexpr int ES = DTraits<DT>::ES;
  const int kc = threadIdx.x & (G - 1), pl = threadIdx.x / G, lanes = 256 / G;
  const int cls = blockIdx.z;
  const int ooy = p.ooy + (p.nclass == 4 ? (cls >> 1) : 0), oox = p.oox + (p.nclass == 4 ? (cls & 1) : 0);
  const char* wrow = p.w + (size_t)cls * (size_t)p.wcs * ES;          // output channel 0 of this class
  const float bias = p.bias ? p.bias[0] : 0.f;
  // (all lanes of a position group run the same trip count: the shuffles below stay convergent)
  for (int m0 = blockIdx.x * lanes; m0 < p.M; m0 += gridDim.x * lanes) {
    const int m = m0 + pl;
    const bool mv = m < p.M;
    int b = 0, oy = 0, ox = 0;
    if (mv) thin_decomp(p, m, b, oy, ox);
    float acc = 0.f;
    if (mv) {
      const size_t ib = (size_t)b * p.Hin * p.Win;
      for (int ty = 0; ty < p.TH; ++ty) {
        int u = oy * p.S + p.dy0 + ty * p.dys;
        bool oku = true;
        if (p.border == CSMRI_BORDER_REFLECT) u = reflect_idx(u, p.Hin); else oku = (unsigned)u < (unsigned)p.Hin;
        for (int tx = 0; tx < p.TW; ++tx) {
          int v = ox * p.S + p.dx0 + tx * p.dxs;
          bool ok = oku;
          if (p.border == CSMRI_BORDER_REFLECT) v = reflect_idx(v, p.Win); else ok = ok && (unsigned)v < (unsigned)p.Win;
          if (!ok) continue;
          const char* xp = p.in0 + (ib + (size_t)u * p.Win + v) * p.ps0 * ES;
          const char* wp = wrow + (size_t)(ty * p.TW + tx) * p.Cin * ES;
          for (int c = kc * 8; c < p.Cin; c += G * 8) acc += thin_dot8<DT>(xp + (size_t)c * ES, wp + (size_t)c * ES);
        }
      }
    }
#pragma unroll
    for (int o = 1; o < G; o <<= 1) acc += __shfl_xor(acc, o);
    if (mv && kc == 0) {
      float v = acc + bias;
      if (p.slope != 1.f) v = v < 0.f ? v * p.slope : v;
      const float out[8] = {v, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const OutPos op = gconv_out_pos(p, b, oy * p.osy + ooy, ox * p.osx + oox);
      thin_store8(op.base, op.opix, p.out_dt, out);
    }
  }
}

int thin_out1_eligible(const csmri_gconv_desc* d) {
  if (d->dtype != CSMRI_BF16 && d->dtype != CSMRI_F32) return 0;
  if (d->Cout != 8 || d->cout_real != 1 || d->in1 || d->upsample) return 0;
  if (d->stats_partial || d->splitk > 1 || d->g_src) return 0;
  if (d->Cin % 8) return 0;
  return 1;
}
static int thin_out1_group(const csmri_gconv_desc* d) {
  // deep K on few positions (the discriminator's final conv: 16 taps x 1024 channels at 200-400 positions): a wave
  // per position; otherwise 8 lanes per position
  const long long m = (long long)d->B * d->Ho * d->Wo * (d->nclass > 0 ? d->nclass : 1);
  return (m < 32768 && d->Cin >= 512) ? 64 : 8;
}

int thin_out1_launch(const GParams& p, const csmri_gconv_desc* d, hipStream_t st) {
  const int G = thin_out1_group(d), lanes = 256 / G;
  long long blocks = ((long long)p.M + lanes - 1) / lanes;
  if (blocks > 8192) blocks = 8192;
  dim3 grid((int)blocks, 1, p.nclass);
#define THIN_O(DT_, G_) hipLaunchKernelGGL((thin_out1_kernel<DT_, G_>), grid, dim3(256), 0, st, p)
  if (d->dtype == CSMRI_BF16) { if (G == 64) THIN_O(CSMRI_BF16, 64); else THIN_O(CSMRI_BF16, 8); }
  else { if (G == 64) THIN_O(CSMRI_F32, 64); else THIN_O(CSMRI_F32, 8); }
#undef THIN_O
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

void thin_kernel_name(const csmri_gconv_desc* d, char* buf, int n) {
  if (thin_in1_eligible(d)) snprintf(buf, n, "thin_in1_kernel<%d, 4, 4>", d->dtype);
  else snprintf(buf, n, "thin_out1_kernel<%d, %d>", d->dtype, thin_out1_group(d));
}
