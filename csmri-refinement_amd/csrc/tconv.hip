// tconv: stride-1 convolutions with few input channels (Cin_pad in {8,16,32,64}), bf16.
//
// For these layers (RecNet 2->32->32->2, the 256^2/128^2 U-Net layers, VGG conv1_x/conv2_1,
// and their input-gradients) the implicit-GEMM gather of gconv.hip re-fetches every input
// pixel once per filter tap (9-16x) and spends its time in address arithmetic and the
// L1/TA path, not in MFMA or HBM.  Here a workgroup owns a 16x16 output tile, loads the
// (16+KH-1)x(16+KW-1) input patch ONCE into LDS -- applying the border rule (zero / reflect),
// the optional nearest x2 upsampling and the two-source channel concat while loading -- and
// then runs the whole K loop out of LDS with no further barrier: per 32-wide K chunk each
// wave reads 4 pixel fragments (ds_read_b128 at a tap-shifted address), takes the weight
// fragments straight from global memory (tiny, L2-resident, prefetched one chunk ahead) and
// issues 4*FN v_mfma_f32_16x16x32_bf16.  HBM traffic = one read of the input (x1.27 halo)
// + one write of the output: these layers become bandwidth-bound as they should be.
//
// K chunk = 32 consecutive packed-K elements = one tap x 32 channels (Cin >= 32) or 32/Cin
// horizontally adjacent taps (Cin = 8, 16; the packed filter width is padded to that multiple
// with zero weights), which is one contiguous 64-byte run of the pixel-major LDS patch.
#include "mma_core.h"
#include "gconv_params.h"

template <int CIN, int FN, bool STREAM>
__global__ __launch_bounds__(256) void tconv_kernel(const GParams p) {
  constexpr int RB = CIN * 2;                       // bytes per patch pixel
  constexpr int VPP = RB / 16;                      // 16-byte vectors per pixel
  constexpr int TPC = CIN >= 32 ? 1 : 32 / CIN;     // taps per K chunk
  constexpr int KCH = CIN >= 32 ? CIN / 32 : 1;     // K chunks per tap
  constexpr int BN = FN * 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const int tiles_x = (p.Wo + 15) >> 4, tiles_y = (p.Ho + 15) >> 4;
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int b = t / (tiles_x * tiles_y);
  t -= b * tiles_x * tiles_y;
  const int tyi = t / tiles_x, txi = t - tyi * tiles_x;
  const int y0 = tyi * 16, x0 = txi * 16, n0 = blockIdx.y * BN;
  const int TPW = 16 + p.TW - 1, TPH = 16 + p.TH - 1;
  const int Hv = p.ups ? 2 * p.Hin : p.Hin, Wv = p.ups ? 2 * p.Win : p.Win;

  // ---- patch load: global -> LDS, once -----------------------------------------------
  // batches of LB independent 16-byte loads per thread are issued before any LDS write so
  // that a workgroup pays one memory round trip per batch, not one per vector
  constexpr int LB = 6;
  const int rowvecs = TPW * VPP, nvec = TPH * rowvecs;
  for (int base = tid; base < nvec; base += 256 * LB) {
    u32x4_t vals[LB];
    int offs[LB];
#pragma unroll
    for (int j = 0; j < LB; ++j) {
      const int v = base + j * 256;
      vals[j] = (u32x4_t){0u, 0u, 0u, 0u};
      offs[j] = -1;
      if (v < nvec) {
        const int py = v / rowvecs, rem = v - py * rowvecs;
        const int px = rem / VPP, cv = rem - px * VPP;
        int u = y0 + p.dy0 + py, w = x0 + p.dx0 + px;
        if (p.border == CSMRI_BORDER_REFLECT) { u = reflect_idx(u, Hv); w = reflect_idx(w, Wv); }
        const bool ok = (unsigned)u < (unsigned)Hv && (unsigned)w < (unsigned)Wv;
        if (p.ups) { u >>= 1; w >>= 1; }
        const int P = py * TPW + px;
        offs[j] = P * RB + ((RB == 128 ? (cv ^ (P & 3)) : cv) << 4);
        if (ok) {
          const size_t pix = ((size_t)b * p.Hin + u) * p.Win + w;
          const int c = cv * 8;
          const char* src = (c < p.c0) ? p.in0 + (pix * p.ps0 + c) * 2 : p.in1 + (pix * p.ps1 + (c - p.c0)) * 2;
          vals[j] = *(const u32x4_t*)src;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < LB; ++j)
      if (offs[j] >= 0) *(u32x4_t*)(smem + offs[j]) = vals[j];
  }
  {
    // weights: all chunks (WHOLE) or stage 0 = first 2 chunks (STREAM) -> LDS tiles
    const int nq_all = p.TH * (p.TW / TPC) * KCH;
    const int nload = STREAM ? min(2, nq_all) : nq_all;
    const char* wsrc0 = p.w + (size_t)n0 * p.Kp * 2;
    char* wl0 = smem + p.nsteps;
    for (int v = tid; v < nload * BN * 4; v += 256) {
      const int c = v / (BN * 4), row = (v >> 2) % BN, sl = v & 3;
      *(u32x4_t*)(wl0 + c * (BN * 64) + tile_off(row, sl)) =
          *(const u32x4_t*)(wsrc0 + ((size_t)row * p.Kp + (size_t)c * 32 + sl * 8) * 2);
    }
  }
  __syncthreads();

  // ---- K loop out of LDS -----------------------------------------------------------------
  f32x4_t acc[FN][4];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int f = 0; f < 4; ++f) acc[i][f] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  int pbase[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) pbase[f] = (4 * wv + f) * TPW + r16;
  const int groups_x = p.TW / TPC;
  const int nq = p.TH * groups_x * KCH;
  const char* wrow = p.w + ((size_t)(n0 + r16) * p.Kp + g * 8) * 2;
  const size_t wfn = (size_t)16 * p.Kp * 2;
  // Weights go through LDS so that the four waves share one fetch (an ablation showed the
  // per-wave L2 re-fetch of the weight fragments, not MFMA or the patch load, bounding the
  // loop).  Layout: per 32-wide K chunk a [BN][64 B] tile with the mma_core swizzle.
  //   WHOLE : all nq chunks are resident (loaded with the patch, no further barrier)
  //   STREAM: SC chunks per stage, double buffered, one barrier per stage
  (void)wrow; (void)wfn;
  int ty = 0, txg = 0, cb = 0;
  auto compute = [&](const char* wt) {      // wt: this chunk's [BN][64 B] weight tile
    const int poff = ty * TPW + txg * TPC;
    u32x4_t a[4], bw[FN];
#pragma unroll
    for (int i = 0; i < FN; ++i) bw[i] = *(const u32x4_t*)(wt + tile_off(i * 16 + r16, g));
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int P = pbase[f] + poff;
      int off;
      if (RB == 128) off = P * RB + (((cb * 4 + g) ^ (P & 3)) << 4);
      else off = P * RB + cb * 64 + g * 16;          // CIN<32: spans TPC adjacent pixels
      a[f] = *(const u32x4_t*)(smem + off);
    }
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int f = 0; f < 4; ++f)
        acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bw[i]),
                                                            __builtin_bit_cast(bf16x8_t, a[f]), acc[i][f], 0, 0, 0);
    if (++cb == KCH) { cb = 0; if (++txg == groups_x) { txg = 0; ++ty; } }
  };
  char* wl = smem + p.nsteps;                  // weight region starts after the patch (host: 16-B aligned)
  constexpr int WT = BN * 64;                  // bytes of one chunk's weight tile
  const char* wsrc = p.w + (size_t)n0 * p.Kp * 2;
  if (!STREAM) {
    // (the loads were issued together with the patch, see wload_whole above the barrier)
    for (int q = 0; q < nq; ++q) compute(wl + q * WT);
  } else {
    constexpr int SC = 2;                      // chunks per stage
    constexpr int SV = SC * BN * 4;            // 16-byte vectors per stage
    constexpr int SI = (SV + 255) / 256;
    const int nst = (nq + SC - 1) / SC;
    u32x4_t wr[SI];
    auto sload = [&](int st) {
#pragma unroll
      for (int it = 0; it < SI; ++it) {
        const int v = tid + it * 256;
        const int c = v / (BN * 4), row = (v >> 2) % BN, sl = v & 3;
        wr[it] = (u32x4_t){0u, 0u, 0u, 0u};
        if (v < SV && st * SC + c < nq)
          wr[it] = *(const u32x4_t*)(wsrc + ((size_t)row * p.Kp + (size_t)(st * SC + c) * 32 + sl * 8) * 2);
      }
    };
    auto sstore = [&](int buf) {
#pragma unroll
      for (int it = 0; it < SI; ++it) {
        const int v = tid + it * 256;
        const int c = v / (BN * 4), row = (v >> 2) % BN, sl = v & 3;
        if (v < SV) *(u32x4_t*)(wl + (buf * SC + c) * WT + tile_off(row, sl)) = wr[it];
      }
    };
    // stage 0 was stored before the first barrier (see below); pipeline the rest
    for (int st = 0; st < nst; ++st) {
      const bool more = st + 1 < nst;
      if (more) sload(st + 1);
#pragma unroll
      for (int c = 0; c < SC; ++c)
        if (st * SC + c < nq) compute(wl + ((st & 1) * SC + c) * WT);
      if (more) sstore((st + 1) & 1);
      __syncthreads();
    }
  }

  // ---- epilogue (same contract as gconv) -----------------------------------------------------
  float s1[FN][4], s2[FN][4];
  if (p.stats) {
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) { s1[i][r] = 0.f; s2[i][r] = 0.f; }
  }
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const int oy = y0 + 4 * wv + f, ox = x0 + r16;
    const bool mv = oy < p.Ho && ox < p.Wo;
    const size_t pp = ((size_t)b * p.Hout_t + (oy * p.osy + p.ooy)) * p.Wout_t + (ox * p.osx + p.oox);
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      const int n = n0 + i * 16 + g * 4;
      if (!(mv && n < p.Cout)) continue;
      f32x4_t v = acc[i][f];
      if (p.bias) v += *(const f32x4_t*)(p.bias + n);
      if (p.stats) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[i][r] += v[r]; s2[i][r] += v[r] * v[r]; }
      }
      if (p.slope != 1.f) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] < 0.f ? v[r] * p.slope : v[r];
      }
      if (p.gsrc) {
        f32x4_t gs = load4(p.gsrc, pp * p.gps + n, p.gdt);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gs[r] > 0.f ? v[r] : v[r] * p.gslope;
      }
      store4(p.out, pp * p.ops + n, p.out_dt, v);
    }
  }
  if (p.stats) {
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a1 = s1[i][r], a2 = s2[i][r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { a1 += __shfl_xor(a1, o); a2 += __shfl_xor(a2, o); }
        const int n = n0 + i * 16 + g * 4 + r;
        if (r16 == 0 && n < p.Cout) {
          float* row = p.stats + (size_t)(blockIdx.x * 4 + wv) * 2 * p.Cout;
          row[n] = a1; row[p.Cout + n] = a2;
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------
int tconv_eligible(const csmri_gconv_desc* d) {
  if (d->dtype != CSMRI_BF16 || d->in_s != 1 || d->dy_step != 1 || d->dx_step != 1) return 0;
  if ((d->nclass > 1) || d->splitk > 1) return 0;
  if (!(d->Cin == 8 || d->Cin == 16 || d->Cin == 32 || d->Cin == 64)) return 0;
  static const char* maxcin_env = getenv("CSMRI_TCONV_MAXCIN");     // A/B knob
  if (d->Cin > (maxcin_env ? atoi(maxcin_env) : 64)) return 0;
  const int tpc = d->Cin >= 32 ? 1 : 32 / d->Cin;
  if (d->TW % tpc) return 0;
  if (d->out_sy != 1 || d->out_sx != 1) return 0;
  if ((long long)d->Ho * d->Wo < 64 * 64) return 0;            // small maps: generic / split-K path
  const size_t lds = (size_t)(16 + d->TH - 1) * (16 + d->TW - 1) * d->Cin * 2;
  if (lds > 96 * 1024) return 0;
  return 1;
}
int tconv_fn(const csmri_gconv_desc* d) { return d->Cout > 32 ? 4 : (d->Cout > 16 ? 2 : 1); }
int tconv_stats_rows(const csmri_gconv_desc* d) {
  return d->B * ((d->Ho + 15) / 16) * ((d->Wo + 15) / 16) * 4;
}

template <int CIN, int FN, bool STREAM>
static int launch_tconv(const GParams& p0, const csmri_gconv_desc* d, hipStream_t st) {
  GParams p = p0;
  const int patch = ((16 + d->TH - 1) * (16 + d->TW - 1) * CIN * 2 + 15) & ~15;
  const int tpc = CIN >= 32 ? 1 : 32 / CIN, kch = CIN >= 32 ? CIN / 32 : 1;
  const int nq = d->TH * (d->TW / tpc) * kch;
  const int lds = patch + (STREAM ? 2 * 2 : nq) * FN * 16 * 64;
  p.nsteps = patch;                 // tconv reuses this field: byte offset of the weight tiles
  static int attr = 0;
  auto kern = tconv_kernel<CIN, FN, STREAM>;
  if (lds > attr) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    attr = lds;
  }
  const int tiles = d->B * ((d->Ho + 15) / 16) * ((d->Wo + 15) / 16);
  dim3 grid(tiles, (d->Cout + FN * 16 - 1) / (FN * 16), 1);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

int tconv_launch(const GParams& p, const csmri_gconv_desc* d, hipStream_t st) {
  const int fn = tconv_fn(d);
  // whole filter resident in LDS when patch + weights stay under 64 KiB (>= 2 workgroups/CU)
  const int tpc_ = d->Cin >= 32 ? 1 : 32 / d->Cin, kch_ = d->Cin >= 32 ? d->Cin / 32 : 1;
  const int nq_ = d->TH * (d->TW / tpc_) * kch_;
  const int patch_ = (16 + d->TH - 1) * (16 + d->TW - 1) * d->Cin * 2;
  const bool stream = patch_ + nq_ * fn * 16 * 64 > 64 * 1024;
#define TC(C_, F_) do { if (stream) return launch_tconv<C_, F_, true>(p, d, st); return launch_tconv<C_, F_, false>(p, d, st); } while (0)
#define TCC(C_) do { if (fn == 4) TC(C_, 4); else if (fn == 2) TC(C_, 2); else TC(C_, 1); } while (0)
  switch (d->Cin) {
    case 8: TCC(8);
    case 16: TCC(16);
    case 32: TCC(32);
    default: TCC(64);
  }
#undef TCC
#undef TC
}

void tconv_kernel_name(const csmri_gconv_desc* d, char* buf, int n) {
  const int fn = tconv_fn(d);
  const int tpc_ = d->Cin >= 32 ? 1 : 32 / d->Cin, kch_ = d->Cin >= 32 ? d->Cin / 32 : 1;
  const int nq_ = d->TH * (d->TW / tpc_) * kch_;
  const int patch_ = (16 + d->TH - 1) * (16 + d->TW - 1) * d->Cin * 2;
  const bool stream = patch_ + nq_ * fn * 16 * 64 > 64 * 1024;
  snprintf(buf, n, "tconv_kernel<%d, %d, %s>", d->Cin, fn, stream ? "true" : "false");
}
