// csmri_gconv: implicit-GEMM "gather convolution" on MFMA for gfx950.
//
// GEMM view: D[n][m] = sum_k W[n][k] * X[m][k]
//   m = output position (b, oy, ox)           (BM positions per workgroup)
//   n = output channel                         (BN channels per workgroup)
//   k = (tap, input channel); X[m][k] is gathered on the fly from the NHWC
//       input with the border rule / x2 upsampling / two-source concat applied
//       in the address computation -- no padded, upsampled or concatenated tensor
//       is ever materialised in HBM.
// Staging: 16-byte global loads -> registers -> swizzled LDS tiles (double
// buffered, one barrier per K step), next step's loads issued before the MFMAs of
// the current step.  256 threads = 4 waves; each wave owns a (BN/WN)x(BM/WM)
// block of D as 16x16 MFMA fragments.
//
// Algorithmic work: 2*M*N*K FLOP; bytes: M*Cin*ES*(taps reuse served by L2/LDS)
// + N*K*ES weights + M*N*ES output.
#include "mma_core.h"
#include "gconv_params.h"

template <int DT, int BM, int BN, int WM, int WN, int KC>
__global__ __launch_bounds__(256) void gconv_kernel(const GParams p) {
  using Tr = DTraits<DT>;
  constexpr int VE = Tr::VE, BKE = Tr::BKE, ES = Tr::ES;
  constexpr int WTM = BM / WM, WTN = BN / WN, FM = WTM / 16, FN = WTN / 16;
  constexpr int RQ = BM / 64;            // Q rows per thread
  constexpr int QI = KC * RQ;            // Q vectors per thread per step
  constexpr int PV = KC * BN * 4;        // P vectors per step (whole workgroup)
  constexpr int PI = (PV + 255) / 256;
  constexpr int TILE_Q = BM * 64, TILE_P = BN * 64;
  constexpr int BUF = KC * (TILE_Q + TILE_P);
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int t = xcd_remap(blockIdx.x, p.mtiles * p.ntiles);
  const int mt = t / p.ntiles, nt = t - mt * p.ntiles;
  const int cls = blockIdx.z % p.nclass, ks = blockIdx.z / p.nclass;
  const int m0 = mt * BM, n0 = nt * BN;
  const int s_begin = ks * p.steps_per_split;
  const int s_end = min(p.nsteps, s_begin + p.steps_per_split);
  const char* wbase = p.w + (size_t)cls * (size_t)p.wcs * ES;
  const int ooy = p.ooy + (p.nclass == 4 ? (cls >> 1) : 0);
  const int oox = p.oox + (p.nclass == 4 ? (cls & 1) : 0);
  const int Hv = p.ups ? 2 * p.Hin : p.Hin, Wv = p.ups ? 2 * p.Win : p.Win;
  const int HoWo = p.Ho * p.Wo;

  // ---- per-thread gather state -------------------------------------------
  const int chunk = tid & 3, qrow = tid >> 2;
  int by[RQ], bx[RQ], ib[RQ];
#pragma unroll
  for (int j = 0; j < RQ; ++j) {
    int m = m0 + qrow + j * 64;
    if (m < p.M) {
      int b = m / HoWo, r = m - b * HoWo, oy = r / p.Wo, ox = r - oy * p.Wo;
      by[j] = oy * p.S + p.dy0; bx[j] = ox * p.S + p.dx0; ib[j] = b * p.Hin * p.Win;
    } else { by[j] = 0; bx[j] = 0; ib[j] = -1; }
  }
  int ty[KC], tx[KC], ci[KC], pix[KC][RQ];
  auto compute_pix = [&](int kc) {
    const bool tapok = ty[kc] < p.TH;
    const int oy_ = ty[kc] * p.dys, ox_ = tx[kc] * p.dxs;
#pragma unroll
    for (int j = 0; j < RQ; ++j) {
      int u = by[j] + oy_, v = bx[j] + ox_;
      bool ok = tapok && ib[j] >= 0;
      if (p.border == CSMRI_BORDER_REFLECT) { u = reflect_idx(u, Hv); v = reflect_idx(v, Wv); }
      else ok = ok && (unsigned)u < (unsigned)Hv && (unsigned)v < (unsigned)Wv;
      if (p.ups) { u >>= 1; v >>= 1; }
      pix[kc][j] = ok ? ib[j] + u * p.Win + v : -1;
    }
  };
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) {
    int k = (s_begin * KC + kc) * BKE + chunk * VE;
    int tap = k / p.Cin;
    ci[kc] = k - tap * p.Cin; ty[kc] = tap / p.TW; tx[kc] = tap - ty[kc] * p.TW;
    compute_pix(kc);
  }

  u32x4_t qreg[QI], preg[PI];
  auto load_step = [&](int s) {
#pragma unroll
    for (int it = 0; it < QI; ++it) {
      const int kc = it / RQ, j = it % RQ;
      const int px = pix[kc][j];
      u32x4_t v = (u32x4_t){0u, 0u, 0u, 0u};
      if (px >= 0) {
        const int c = ci[kc];
        const char* src = (c < p.c0) ? p.in0 + ((size_t)px * p.ps0 + c) * ES
                                     : p.in1 + ((size_t)px * p.ps1 + (c - p.c0)) * ES;
        v = *(const u32x4_t*)src;
      }
      qreg[it] = v;
    }
#pragma unroll
    for (int it = 0; it < PI; ++it) {
      const int v = tid + it * 256;
      if (PV % 256 == 0 || v < PV) {
        const int row = (v >> 2) % BN, kc = (v >> 2) / BN;
        preg[it] = *(const u32x4_t*)(wbase + ((size_t)(n0 + row) * p.Kp +
                                              (size_t)(s * KC + kc) * BKE + chunk * VE) * ES);
      }
    }
    // advance the tap state to step s+1
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      ci[kc] += BKE * KC;
      if (ci[kc] >= p.Cin) {
        do { ci[kc] -= p.Cin; if (++tx[kc] == p.TW) { tx[kc] = 0; ++ty[kc]; } } while (ci[kc] >= p.Cin);
        compute_pix(kc);
      }
    }
  };
  auto store_step = [&](int buf) {
    char* base = smem + buf * BUF;
#pragma unroll
    for (int it = 0; it < QI; ++it) {
      const int kc = it / RQ, row = qrow + (it % RQ) * 64;
      *(u32x4_t*)(base + kc * TILE_Q + tile_off(row, chunk)) = qreg[it];
    }
#pragma unroll
    for (int it = 0; it < PI; ++it) {
      const int v = tid + it * 256;
      if (PV % 256 == 0 || v < PV) {
        const int row = (v >> 2) % BN, kc = (v >> 2) / BN;
        *(u32x4_t*)(base + KC * TILE_Q + kc * TILE_P + tile_off(row, chunk)) = preg[it];
      }
    }
  };

  MmaCore<DT, FN, FM> core;
  core.zero();

  if (s_begin < s_end) {
    load_step(s_begin);
    store_step(0);
    __syncthreads();
    for (int s = s_begin; s < s_end; ++s) {
      const int cur = (s - s_begin) & 1;
      const bool more = s + 1 < s_end;
      if (more) load_step(s + 1);
      const char* base = smem + cur * BUF;
#pragma unroll
      for (int kc = 0; kc < KC; ++kc)
        core.step(base + KC * TILE_Q + kc * TILE_P, base + kc * TILE_Q, wn * WTN, wm * WTM, lane);
      if (more) store_step(cur ^ 1);
      __syncthreads();
    }
  }

  // ---- epilogue -------------------------------------------------------------
  const int g = lane >> 4, r16 = lane & 15;
  float s1[FN][4], s2[FN][4];
  if (p.stats) {
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) { s1[i][r] = 0.f; s2[i][r] = 0.f; }
  }
#pragma unroll
  for (int j = 0; j < FM; ++j) {
    const int m = m0 + wm * WTM + j * 16 + r16;
    const bool mv = m < p.M;
    OutPos op; op.base = p.out; op.opix = 0; op.gpix = 0; op.g_ok = true;
    if (mv) {
      int b = m / HoWo, r = m - b * HoWo, oy = r / p.Wo, ox = r - oy * p.Wo;
      op = gconv_out_pos(p, b, oy * p.osy + ooy, ox * p.osx + oox);
    }
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      const int n = n0 + wn * WTN + i * 16 + g * 4;
      f32x4_t v = core.acc[i][j];
      if (p.splitk > 1) {
        if (mv && n < p.Cout)
          *(f32x4_t*)(p.slab + (((size_t)cls * p.splitk + ks) * p.M + m) * p.Cout + n) = v;
        continue;
      }
      if (!(mv && n < p.Cout)) continue;
      if (p.bias) { f32x4_t bb = *(const f32x4_t*)(p.bias + n); v += bb; }
      if (p.stats) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[i][r] += v[r]; s2[i][r] += v[r] * v[r]; }
      }
      if (p.slope != 1.f) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] < 0.f ? v[r] * p.slope : v[r];
      }
      if (p.gsrc && op.g_ok) {
        f32x4_t gs = load4(p.gsrc, op.gpix + n, p.gdt);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gs[r] > 0.f ? v[r] : v[r] * p.gslope;
      }
      store4(op.base, op.opix + n, p.out_dt, v);
    }
  }
  if (p.stats && p.splitk == 1) {
    // reduce over the 16 pixels held by the 16 lanes of each lane group
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a = s1[i][r], b = s2[i][r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
        const int n = n0 + wn * WTN + i * 16 + g * 4 + r;
        if (r16 == 0 && n < p.Cout) {
          // partial sums are channel-major: [2][Cout][rows], rows = mtiles * WM (finalize reads coalesced)
          const size_t R = (size_t)p.mtiles * WM, r = (size_t)mt * WM + wm;
          p.stats[(size_t)n * R + r] = a; p.stats[((size_t)p.Cout + n) * R + r] = b;
        }
      }
  }
}

// split-K second stage: sum slabs, then the same epilogue (bias/act/actgrad)
__global__ void gconv_reduce_kernel(const GParams p) {
  const int HoWo = p.Ho * p.Wo;
  const unsigned nv = p.Cout / 4;
  const unsigned per_class = (unsigned)p.M * nv;            // host: M * Cout * nclass < 2^31
  const unsigned total = per_class * p.nclass;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int cls = (int)(i / per_class);
    const unsigned rem = i - cls * per_class;
    const int m = (int)(rem / nv), n = (int)(rem - (unsigned)m * nv) * 4;
    const float* slab = p.slab + (size_t)cls * p.splitk * p.M * p.Cout + (size_t)m * p.Cout + n;
    const size_t zstride = (size_t)p.M * p.Cout;
    f32x4_t v = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < p.splitk; ++z) v += *(const f32x4_t*)(slab + z * zstride);
    OutPos op;
    if (p.dense_out) {                 // output position index == m: no decomposition (two integer divisions per vector)
      op.base = p.out; op.opix = (size_t)m * p.ops; op.gpix = (size_t)m * p.gps; op.g_ok = true;
    } else {
      int b, oy, ox;
      if (p.howo_shift >= 0) { b = m >> p.howo_shift; const int r = m & (HoWo - 1); oy = r >> p.wo_shift; ox = r & (p.Wo - 1); }
      else { b = m / HoWo; const int r = m - b * HoWo; oy = r / p.Wo; ox = r - oy * p.Wo; }
      const int ooy = p.ooy + (p.nclass == 4 ? (cls >> 1) : 0), oox = p.oox + (p.nclass == 4 ? (cls & 1) : 0);
      op = gconv_out_pos(p, b, oy * p.osy + ooy, ox * p.osx + oox);
    }
    if (p.bias) v += *(const f32x4_t*)(p.bias + n);
    if (p.slope != 1.f)
      for (int q = 0; q < 4; ++q) v[q] = v[q] < 0.f ? v[q] * p.slope : v[q];
    if (p.gsrc && op.g_ok) {
      f32x4_t gs = load4(p.gsrc, op.gpix + n, p.gdt);
      for (int q = 0; q < 4; ++q) v[q] = gs[q] > 0.f ? v[q] : v[q] * p.gslope;
    }
    store4(op.base, op.opix + n, p.out_dt, v);
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
struct GConfig { int BM, BN, WM, KC; };

static GConfig pick_config(const csmri_gconv_desc* d) {
  GConfig c;
  c.KC = d->dtype == CSMRI_BF16 ? 2 : 1;
  if (d->Cout > 64) { c.BM = 128; c.BN = 128; c.WM = 2; }
  else if (d->Cout > 32) { c.BM = 128; c.BN = 64; c.WM = 2; }
  else if (d->Cout > 16) { c.BM = 256; c.BN = 32; c.WM = 4; }
  else { c.BM = 256; c.BN = 16; c.WM = 4; }
  return c;
}

static int desc_M(const csmri_gconv_desc* d) { return d->B * d->Ho * d->Wo; }

extern "C" int csmri_gconv_stats_rows(const csmri_gconv_desc* d0) {
  csmri_gconv_desc t = *d0;                          // (asked before the caller has allocated the rows: dispatch as the launch will)
  if (!t.stats_partial) t.stats_partial = (float*)16;
  const csmri_gconv_desc* d = &t;
  if (uconv_eligible(d)) return uconv_stats_rows(d);
  if (tconv_eligible(d)) return tconv_stats_rows(d);
  if (gpipe_eligible(d)) return gpipe_stats_rows(d);
  GConfig c = pick_config(d);
  return cdiv(desc_M(d), c.BM) * c.WM;
}

extern "C" size_t csmri_gconv_slab_bytes(const csmri_gconv_desc* d) {
  if (d->splitk <= 1) return 0;
  int nclass = d->nclass > 0 ? d->nclass : 1;
  return (size_t)nclass * d->splitk * desc_M(d) * d->Cout * sizeof(float);
}

#ifndef UCONV_BEFORE_PCONV2
#define UCONV_BEFORE_PCONV2 0     // 64 -> 128 channels 3 x 3 (VGG conv2_1): which of the two patch kernels takes it
#endif
#ifndef SK_T1
#define SK_T1 512
#endif
#ifndef SK_T2
#define SK_T2 768
#endif
extern "C" int csmri_gconv_suggest_splitk(const csmri_gconv_desc* d) {
  { csmri_gconv_desc t = *d; t.splitk = 1;
    if (thin_out1_eligible(&t) || tconv_eligible(&t) || pconv2_eligible(&t) || uconv_eligible(&t)) return 1;
    if (gpipe_eligible(&t)) return gpipe_splitk(&t); }
  GConfig c = pick_config(d);
  if (gconv_glds_eligible(d)) { c.BM = 128; c.BN = gconv_glds_bn(d); }
  if (d->dtype == CSMRI_FP8) { c.BM = 128; c.BN = gconv_fp8_bn(d); c.KC = 1; }
  int nclass = d->nclass > 0 ? d->nclass : 1;
  long long tiles = (long long)cdiv(desc_M(d), c.BM) * cdiv(d->Cout, c.BN) * nclass;
  int bke = d->dtype == CSMRI_FP8 ? 128 : (d->dtype == CSMRI_BF16 ? 32 : 16) * c.KC;
  int nsteps = cdiv((long long)d->TH * d->TW * d->Cin, bke);
  // measured (tools/bench_conv.py, main + reduce; bench.py): from ~224 tiles on the two-buffer kernel without
  // split-K beats 2-3 slices + reduce (36 vs 44 us on a 256-tile layer, +2 % on the step); ~192-tile problems want
  // 3 slices (768 workgroups), smaller ones ~512 workgroups (less slab traffic for the same latency hiding)
  // ... except deep-K problems at exactly one workgroup per CU (VGG conv4_x data gradients on 8 images: 256 tiles,
  // 72 steps): 2 slices = 2 workgroups per CU overlap each other's loads, 53.7 vs 64.1 us incl. the reduce
  // (tools/sk_sweep.sh); at 32 steps (U-Net 128 -> 128 4x4) the same split loses 2x
  const bool deep = tiles <= 256 && nsteps >= 64 && !d->out_halo && nclass == 1;
  if (tiles >= 224 && !deep) return 1;
  const int target = deep ? SK_T1 : tiles >= 192 ? SK_T2 : SK_T1;
  int sk = (int)((target + tiles - 1) / tiles);
  int maxsk = nsteps / 8; if (maxsk < 1) maxsk = 1;
  if (sk > maxsk) sk = maxsk;
  if (sk > 32) sk = 32;
  return sk < 1 ? 1 : sk;
}

template <int DT, int BM, int BN, int WM, int WN, int KC>
static int launch_gconv(const GParams& p, hipStream_t st) {
  constexpr int lds = 2 * KC * (BM + BN) * 64;
  auto kern = gconv_kernel<DT, BM, BN, WM, WN, KC>;
  CSMRI_SET_MAX_LDS(kern, lds);
  dim3 grid(p.mtiles * p.ntiles, 1, p.nclass * p.splitk);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

static int build_params(const csmri_gconv_desc* d, GParams& p, GConfig& c) {
  CSMRI_CHECK_ARG(d && d->in0 && d->w && d->out);
  CSMRI_CHECK_ARG(d->dtype == CSMRI_F32 || d->dtype == CSMRI_BF16 || d->dtype == CSMRI_FP8);
  if (d->dtype == CSMRI_FP8 && !gconv_fp8_eligible(d)) return CSMRI_E_UNSUPPORTED;
  CSMRI_CHECK_ARG(d->out_dtype == CSMRI_F32 || d->out_dtype == CSMRI_BF16);
  CSMRI_CHECK_ARG(d->Cin > 0 && d->Cin % 8 == 0 && d->Cout > 0 && d->Cout % 8 == 0);
  CSMRI_CHECK_ARG(d->B > 0 && d->Hin > 0 && d->Win > 0 && d->Ho > 0 && d->Wo > 0);
  CSMRI_CHECK_ARG(d->TH > 0 && d->TW > 0 && d->in_s > 0);
  CSMRI_CHECK_ARG(d->in0_pix_stride % 8 == 0 && d->out_pix_stride % 4 == 0);
  if (d->in1) CSMRI_CHECK_ARG(d->c0 > 0 && d->c0 % 8 == 0 && d->c0 < d->Cin && d->in1_pix_stride % 8 == 0);
  if (((uintptr_t)d->in0 | (uintptr_t)d->w | (uintptr_t)d->out | (uintptr_t)d->in1) & 15) return CSMRI_E_ALIGN;
  const int nclass = d->nclass > 0 ? d->nclass : 1;
  CSMRI_CHECK_ARG(nclass == 1 || nclass == 4);
  const int splitk = d->splitk > 0 ? d->splitk : 1;
  if (splitk > 1) CSMRI_CHECK_ARG(d->slab != nullptr);
  if (splitk > 1 && d->stats_partial) return CSMRI_E_UNSUPPORTED;
  if (d->g_src) CSMRI_CHECK_ARG(d->g_pix_stride % 4 == 0);

  c = pick_config(d);
  p.in0 = (const char*)d->in0; p.in1 = (const char*)d->in1;
  p.ps0 = d->in0_pix_stride; p.ps1 = d->in1_pix_stride; p.c0 = d->in1 ? d->c0 : d->Cin;
  p.B = d->B; p.Hin = d->Hin; p.Win = d->Win; p.Cin = d->Cin; p.ups = d->upsample; p.border = d->border;
  p.TH = d->TH; p.TW = d->TW; p.S = d->in_s; p.dy0 = d->dy0; p.dys = d->dy_step; p.dx0 = d->dx0; p.dxs = d->dx_step;
  p.w = (const char*)d->w; p.Kp = d->Kp; p.nclass = nclass; p.wcs = d->w_class_stride;
  p.out = (char*)d->out; p.ops = d->out_pix_stride; p.Hout_t = d->Hout_t; p.Wout_t = d->Wout_t;
  p.Ho = d->Ho; p.Wo = d->Wo; p.osy = d->out_sy; p.osx = d->out_sx; p.ooy = d->out_oy; p.oox = d->out_ox;
  p.Cout = d->Cout; p.out_dt = d->out_dtype;
  p.bias = d->bias; p.slope = d->act_slope; p.gsrc = (const char*)d->g_src; p.gps = d->g_pix_stride;
  p.gslope = d->g_slope; p.gdt = d->g_dtype;
  p.stats = d->stats_partial; p.splitk = splitk; p.slab = d->slab;
  p.out2 = (char*)d->out_halo; p.o2ps = d->halo_pix_stride;
  p.win_y0 = d->win_y0; p.win_x0 = d->win_x0; p.win_h = d->win_h; p.win_w = d->win_w;
  if (d->out_halo) {
    CSMRI_CHECK_ARG(d->win_h > 0 && d->win_w > 0 && d->halo_pix_stride % 4 == 0 && !d->stats_partial);
    if ((uintptr_t)d->out_halo & 15) return CSMRI_E_ALIGN;
  }
  p.M = desc_M(d);
  const int bke = d->dtype == CSMRI_FP8 ? 128 : (d->dtype == CSMRI_BF16 ? 32 : 16) * c.KC;
  p.nsteps = cdiv((long long)d->TH * d->TW * d->Cin, bke);
  CSMRI_CHECK_ARG((long long)p.nsteps * bke <= d->Kp);
  p.steps_per_split = cdiv(p.nsteps, splitk);
  p.mtiles = cdiv(p.M, c.BM); p.ntiles = cdiv(d->Cout, c.BN); p.nt_major = 0;
  auto lg2 = [](long long v) { int s = 0; while ((1ll << s) < v) ++s; return (1ll << s) == v ? s : -1; };
  p.wo_shift = lg2(d->Wo); p.howo_shift = lg2((long long)d->Ho * d->Wo);
  if (p.wo_shift < 0 || p.howo_shift < 0) p.wo_shift = p.howo_shift = -1;
  p.dq0 = d->dtype == CSMRI_FP8 ? d->in_dequant : nullptr; p.dq1 = d->dtype == CSMRI_FP8 ? d->w_dequant : nullptr;
  p.outq = (char*)d->out_q; p.oqps = d->out_q_pix_stride; p.oqs = d->out_q_scale; p.oamax = (unsigned*)d->out_amax;
  if (d->out_q) CSMRI_CHECK_ARG(d->out_q_scale && d->out_q_pix_stride % 8 == 0 && !((uintptr_t)d->out_q & 7));
  p.dense_out = !d->out_halo && nclass == 1 && d->out_sy == 1 && d->out_sx == 1 && d->out_oy == 0 && d->out_ox == 0 &&
                d->Hout_t == d->Ho && d->Wout_t == d->Wo;
  const long long in_px = (long long)d->B * d->Hin * d->Win, out_px = (long long)d->B * d->Hout_t * d->Wout_t;
  const long long max_ps = d->in0_pix_stride > d->in1_pix_stride ? d->in0_pix_stride : d->in1_pix_stride;
  p.off32 = in_px * max_ps * 2 < (1ll << 31) && out_px * d->out_pix_stride * 4 < (1ll << 31) &&
            out_px * (d->g_src ? d->g_pix_stride : 0) * 4 < (1ll << 31);
  return CSMRI_OK;
}

#ifndef GRED_MAX_BLOCKS
#define GRED_MAX_BLOCKS 4096
#endif
static int launch_reduce(const GParams& p, hipStream_t st) {
  long long total = (long long)p.M * (p.Cout / 4) * p.nclass;
  if (total * 4 >= (1ll << 31)) return CSMRI_E_UNSUPPORTED;
  int blocks = (int)((total + 255) / 256); if (blocks > GRED_MAX_BLOCKS) blocks = GRED_MAX_BLOCKS;
  hipLaunchKernelGGL(gconv_reduce_kernel, dim3(blocks), dim3(256), 0, st, p);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

// template instance csmri_gconv dispatches to for this problem, spelled as rocprofv3 prints it
extern "C" int csmri_gconv_kernel_name(const csmri_gconv_desc* d, char* buf, int n) {
  CSMRI_CHECK_ARG(d && buf && n > 0);
  if (d->dtype == CSMRI_FP8 && pconv2_eligible(d)) { snprintf(buf, n, "pconv2_kernel<3, 3, %d, true>", pconv2_bn(d)); return CSMRI_OK; }
  if (d->dtype == CSMRI_FP8) { gconv_fp8_kernel_name(d, buf, n); return CSMRI_OK; }
  if (thin_out1_eligible(d)) { thin_kernel_name(d, buf, n); return CSMRI_OK; }
  if (UCONV_BEFORE_PCONV2 && uconv_eligible(d)) { uconv_kernel_name(d, buf, n); return CSMRI_OK; }
  if (pconv2_eligible(d)) { snprintf(buf, n, "pconv2_kernel<3, 3, %d, false>", pconv2_bn(d)); return CSMRI_OK; }
  if (uconv_eligible(d)) { uconv_kernel_name(d, buf, n); return CSMRI_OK; }
  if (tconv_eligible(d)) { tconv_kernel_name(d, buf, n); return CSMRI_OK; }
  if (gpipe_eligible(d)) { gpipe_kernel_name(d, buf, n); return CSMRI_OK; }
  if (gconv_glds_eligible(d)) { gconv_glds_kernel_name(d, buf, n); return CSMRI_OK; }
  GConfig c = pick_config(d);
  const int wn = c.BN >= 64 ? 2 : 1;
  snprintf(buf, n, "gconv_kernel<%d, %d, %d, %d, %d, %d>", d->dtype, c.BM, c.BN, c.WM, wn,
           d->dtype == CSMRI_BF16 ? c.KC : 1);
  return CSMRI_OK;
}

// second stage of a split-K csmri_gconv launched with CSMRI_GCONV_DEFER_REDUCE
extern "C" int csmri_gconv_reduce(const csmri_gconv_desc* d, void* stream) {
  GParams p; GConfig c;
  int rc = build_params(d, p, c);
  if (rc != CSMRI_OK) return rc;
  if (p.splitk <= 1) return CSMRI_OK;
  return launch_reduce(p, (hipStream_t)stream);
}

extern "C" int csmri_gconv(const csmri_gconv_desc* d, void* stream) {
  GParams p; GConfig c;
  int rc = build_params(d, p, c);
  if (rc != CSMRI_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  if ((d->out_q || d->out_amax) && !pconv2_eligible(d)) return CSMRI_E_UNSUPPORTED;
  if (d->dtype == CSMRI_FP8 && pconv2_eligible(d)) return pconv2_launch(p, d, st);
  if (d->dtype == CSMRI_FP8) {
    rc = gconv_fp8_launch(p, d, st);
    if (rc != CSMRI_OK) return rc;
    if (p.splitk > 1 && !(d->flags & CSMRI_GCONV_DEFER_REDUCE)) return launch_reduce(p, st);
    return CSMRI_OK;
  }
  if (thin_out1_eligible(d)) return thin_out1_launch(p, d, st);
  if (UCONV_BEFORE_PCONV2 && uconv_eligible(d)) return uconv_launch(p, d, st);
  if (pconv2_eligible(d)) return pconv2_launch(p, d, st);
  if (uconv_eligible(d)) return uconv_launch(p, d, st);
  if (tconv_eligible(d)) return tconv_launch(p, d, st);
  if (gpipe_eligible(d)) {
    rc = gpipe_launch(p, d, st);
    if (rc != CSMRI_OK) return rc;
    if (p.splitk > 1 && !(d->flags & CSMRI_GCONV_DEFER_REDUCE)) return launch_reduce(p, st);
    return CSMRI_OK;
  }
  if (gconv_glds_eligible(d)) {
    rc = gconv_glds_launch(p, d, st);
    if (rc != CSMRI_OK) return rc;
    if (p.splitk > 1 && !(d->flags & CSMRI_GCONV_DEFER_REDUCE)) return launch_reduce(p, st);
    return CSMRI_OK;
  }
#define GC(DT_, BM_, BN_, WM_, WN_, KC_) rc = launch_gconv<DT_, BM_, BN_, WM_, WN_, KC_>(p, st)
  if (d->dtype == CSMRI_BF16) {
    if (c.BN == 128 && c.KC == 1) GC(CSMRI_BF16, 128, 128, 2, 2, 1);
    else if (c.BN == 128) GC(CSMRI_BF16, 128, 128, 2, 2, 2);
    else if (c.BN == 64) GC(CSMRI_BF16, 128, 64, 2, 2, 2);
    else if (c.BN == 32) GC(CSMRI_BF16, 256, 32, 4, 1, 2);
    else GC(CSMRI_BF16, 256, 16, 4, 1, 2);
  } else {
    if (c.BN == 128) GC(CSMRI_F32, 128, 128, 2, 2, 1);
    else if (c.BN == 64) GC(CSMRI_F32, 128, 64, 2, 2, 1);
    else if (c.BN == 32) GC(CSMRI_F32, 256, 32, 4, 1, 1);
    else GC(CSMRI_F32, 256, 16, 4, 1, 1);
  }
#undef GC
  if (rc != CSMRI_OK) return rc;
  if (p.splitk > 1 && !(d->flags & CSMRI_GCONV_DEFER_REDUCE)) return launch_reduce(p, st);
  return CSMRI_OK;
}
