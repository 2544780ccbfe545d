// Thin layer: the data gradient of the discriminator's first layer (64 -> 1 channel, stride 2: reference
// models/discriminators.py:137-150 applied to |x| of training/adversarial_training.py:33-40), which sits on the
// generator backward's critical chain.  Its output tensor carries the single channel padded to 8, so the MFMA kernel
// of gconv.hip spent a 16-wide N side on one channel: 68.8 us at the bench shape for 25 MB of operands.
//
//   thin_out1_kernel  Cout_real = 1, parity-class (stride-2 data-gradient) descriptors: 8 lanes share one output
//                     position, each reads 16 bytes of every tap's channel run and of the weight row, 8 FMAs, and
//                     the lanes combine by wave shuffles (fixed order); lane 0 stores the value and 7 zeros: 45 us.
//
//   thin_out1_tile_kernel (round 4, below): the bf16 / 64-channel case of the bench through an LDS tile: 16 x 16
//                     positions of the class grid per workgroup, dY staged once, v_dot2c_f32_bf16; 2.4x faster alone.
//
// Both follow csmri_gconv_desc's addressing contract exactly (border rule, strides, parity classes, output window), so
// the consumers of its output (csmri_fold_halo) are unchanged.  fp32 accumulation, one rounding on output.
// Measured and NOT kept (round 3, profiles/r03_thin_layers.log): the same idea for the forward of that layer (a thread
// per position x 8 channels, weights in registers: 145 vs 42 us), for the U-Net head and the discriminator's final
// conv forward (21 vs 10 us, 24 vs 17 us) -- the gathers of a position's taps are latency-bound on the vector unit.
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

template <int DT, int G, int TH, int TW>
__global__ __launch_bounds__(256) void thin_out1_kernel(const GParams p) {
  constexpr int ES = DTraits<DT>::ES, NT = TH * TW;
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
    const size_t ib = (size_t)b * p.Hin * p.Win;
    // all taps' addresses first, then all loads of a channel block in flight together, then the arithmetic
    const char* xp[NT];
    bool ok[NT];
#pragma unroll
    for (int ty = 0; ty < TH; ++ty) {
      int u = oy * p.S + p.dy0 + ty * p.dys;
      bool oku = mv;
      if (p.border == CSMRI_BORDER_REFLECT) u = reflect_idx(u, p.Hin); else oku = oku && (unsigned)u < (unsigned)p.Hin;
#pragma unroll
      for (int tx = 0; tx < TW; ++tx) {
        int v = ox * p.S + p.dx0 + tx * p.dxs;
        bool o = oku;
        if (p.border == CSMRI_BORDER_REFLECT) v = reflect_idx(v, p.Win); else o = o && (unsigned)v < (unsigned)p.Win;
        ok[ty * TW + tx] = o;
        xp[ty * TW + tx] = o ? p.in0 + (ib + (size_t)u * p.Win + v) * p.ps0 * ES : (const char*)wrow;   // (any readable address)
      }
    }
    for (int c = kc * 8; c < p.Cin; c += G * 8) {
      float part[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) part[t] = thin_dot8<DT>(xp[t] + (size_t)c * ES, wrow + ((size_t)t * p.Cin + c) * ES);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc += ok[t] ? part[t] : 0.f;
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

// ------------------------------------------------------------------------------------------------------------
// thin_out1_tile_kernel (round 4): the same layer through an LDS tile.  thin_out1_kernel gathers 4 taps x 128 B per
// output pixel and parity class straight from global memory: 268 MB of requests (155 MB of HBM traffic by the PMC passes)
// for 25 MB of operands, 41 us on the generator backward's critical chain.  Here a workgroup owns 16 x 16 positions of
// the class grid of one image -- a 32 x 32 block of dX -- stages the 17 x 17 dY pixels those positions read ONCE
// (zero outside the map: data-gradient descriptors have a zero border) plus the 4 classes x 4 taps weight rows, and a
// thread computes its position's four class outputs from LDS with v_dot2c_f32_bf16.  Same addressing contract, same
// arithmetic order per output (taps in order, channels in order within a tap) up to the fused pair products.

template <int CIN>
__global__ __launch_bounds__(256) void thin_out1_tile_kernel(const GParams p, int tiles_x, int tiles_y) {
  constexpr int PIX = 17 * 17, PB = CIN * 2;                       // bytes per staged pixel (bf16)
  constexpr int PITCH = PB + 16;                                   // (+16: consecutive pixels start 4 banks apart)
  __shared__ __attribute__((aligned(16))) char sm[PIX * PITCH + 16 * PB];
  char* W = sm + PIX * PITCH;
  const int tid = threadIdx.x;
  int t = blockIdx.x;
  const int b = t / (tiles_x * tiles_y);
  t -= b * tiles_x * tiles_y;
  const int ty0 = (t / tiles_x) * 16, tx0 = (t % tiles_x) * 16;
  // weights: [class][tap][channel], 16 rows of PB bytes
  for (int i = tid; i < 16 * (PB / 16); i += 256) {
    const int row = i / (PB / 16), ch = i - row * (PB / 16);
    const int cls = row >> 2, tap = row & 3;
    *(u32x4_t*)(W + row * PB + ch * 16) =
        *(const u32x4_t*)(p.w + ((size_t)cls * (size_t)p.wcs + (size_t)tap * CIN) * 2 + ch * 16);
  }
  // dY pixels (u, v) = (ty0 + dy0 - 1 + r, tx0 + dx0 - 1 + c), r, c in 0..16
  const size_t ib = (size_t)b * p.Hin * p.Win;
  for (int i = tid; i < PIX * (PB / 16); i += 256) {
    const int px = i / (PB / 16), ch = i - px * (PB / 16);
    const int r = px / 17, c = px - r * 17;
    const int u = ty0 + p.dy0 - 1 + r, v = tx0 + p.dx0 - 1 + c;
    u32x4_t val = (u32x4_t){0u, 0u, 0u, 0u};
    if ((unsigned)u < (unsigned)p.Hin && (unsigned)v < (unsigned)p.Win)
      val = *(const u32x4_t*)(p.in0 + (ib + (size_t)u * p.Win + v) * (size_t)p.ps0 * 2 + ch * 16);
    *(u32x4_t*)(sm + px * PITCH + ch * 16) = val;
  }
  __syncthreads();
  const int ly = tid >> 4, lx = tid & 15;
  const int oy = ty0 + ly, ox = tx0 + lx;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  // tap (ty, tx) reads dY (oy + dy0 - ty, ox + dx0 - tx) = staged pixel (ly + 1 - ty, lx + 1 - tx)
#pragma unroll
  for (int tap = 0; tap < 4; ++tap) {
    const char* xp = sm + ((ly + 1 - (tap >> 1)) * 17 + (lx + 1 - (tap & 1))) * PITCH;
#pragma unroll
    for (int c16 = 0; c16 < PB / 16; ++c16) {
      const u32x4_t x = *(const u32x4_t*)(xp + c16 * 16);
#pragma unroll
      for (int cls = 0; cls < 4; ++cls) {
        const u32x4_t w = *(const u32x4_t*)(W + (cls * 4 + tap) * PB + c16 * 16);      // (same address in every lane: broadcast)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          // (inline asm: with __builtin_amdgcn_fdot2_f32_bf16 on bit-cast vector elements hipcc 7.2 emitted all four
          // products of a 16-byte chunk on element 0 -- channels 0,1 counted four times, 2..7 never)
          asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc[cls]) : "v"(x[q]), "v"(w[q]));
      }
    }
  }
  if (oy < p.Ho && ox < p.Wo) {
    const float bias = p.bias ? p.bias[0] : 0.f;
#pragma unroll
    for (int cls = 0; cls < 4; ++cls) {
      float v = acc[cls] + bias;
      if (p.slope != 1.f) v = v < 0.f ? v * p.slope : v;
      const float out[8] = {v, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const OutPos op = gconv_out_pos(p, b, oy * p.osy + p.ooy + (cls >> 1), ox * p.osx + p.oox + (cls & 1));
      thin_store8(op.base, op.opix, p.out_dt, out);
    }
  }
}

static bool thin_tile_ok(const csmri_gconv_desc* d) {
  return d->dtype == CSMRI_BF16 && d->Cin == 64 && d->in_s == 1 && d->dy_step == -1 && d->dx_step == -1 &&
         d->border == CSMRI_BORDER_ZERO && d->in0_pix_stride % 8 == 0;
}

int thin_out1_eligible(const csmri_gconv_desc* d) {
  if (d->dtype != CSMRI_BF16 && d->dtype != CSMRI_F32) return 0;
  if (d->Cout != 8 || d->cout_real != 1 || d->nclass != 4 || d->in1 || d->upsample) return 0;
  if (d->stats_partial || d->splitk > 1 || d->g_src) return 0;
  if (d->Cin % 8 || d->Cin > 128 || d->TH != 2 || d->TW != 2) return 0;
  return 1;
}

int thin_out1_launch(const GParams& p, const csmri_gconv_desc* d, hipStream_t st) {
  if (thin_tile_ok(d)) {
    const int tx = (d->Wo + 15) / 16, ty = (d->Ho + 15) / 16;
    hipLaunchKernelGGL((thin_out1_tile_kernel<64>), dim3(tx * ty * d->B), dim3(256), 0, st, p, tx, ty);
    CSMRI_LAUNCH_CHECK();
    return CSMRI_OK;
  }
  const int lanes = 256 / 8;
  long long blocks = ((long long)p.M + lanes - 1) / lanes;
  if (blocks > 8192) blocks = 8192;
  dim3 grid((int)blocks, 1, p.nclass);
  if (d->dtype == CSMRI_BF16) hipLaunchKernelGGL((thin_out1_kernel<CSMRI_BF16, 8, 2, 2>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((thin_out1_kernel<CSMRI_F32, 8, 2, 2>), grid, dim3(256), 0, st, p);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

void thin_kernel_name(const csmri_gconv_desc* d, char* buf, int n) {
  if (thin_tile_ok(d)) snprintf(buf, n, "thin_out1_tile_kernel<64>");
  else snprintf(buf, n, "thin_out1_kernel<%d, 8, 2, 2>", d->dtype);
}
