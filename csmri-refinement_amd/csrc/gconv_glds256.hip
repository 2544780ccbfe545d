// gconv_glds256: the LDS-DMA implicit-GEMM convolution of gconv_glds.hip on a 256 x 256 x 64 tile
// with 8 waves (2 x 4, 128 x 64 outputs per wave) and two LDS buffers, for the layers that have
// the shape for it: Cout % 256 == 0 and enough positions that 256-row tiles still fill the chip
// (VGG19 conv3_x / conv4_x and their data gradients).
//
// Against the 128 x 128 tile: half the global->LDS bytes per FLOP (each operand row is shared by
// twice as many outputs), 0.375 instead of 0.5 ds_read_b128 per MFMA (12 fragment reads feed 32
// MFMAs), and ONE workgroup per CU whose two waves per SIMD alternate between fragment reads and
// MFMAs while the next K step streams into the other buffer (it has a whole step, >= 2 x 1024
// MFMA cycles per SIMD, to land: the vmcnt(0) in front of the barrier finds it complete).
// Same LDS image, source-side swizzle, zero page and epilogue as gconv_glds.hip; no BatchNorm
// partial sums (those layers stay on the 128-row kernels).
#include "mma_core.h"
#include "gconv_params.h"

__device__ __attribute__((aligned(16))) char g256_zero_page[16];

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ int g256_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int BN>
__global__ __launch_bounds__(512, 2) void gconv_glds256_kernel(const GParams p) {
  constexpr int BM = 256, NW = 8, WN = BN / 64, WM = NW / WN;   // 2 x 4 waves of 128 x 64, or 4 x 2 of 64 x 64
  constexpr int WTM = BM / WM, WTN = 64, FM = WTM / 16, FN = WTN / 16;
  constexpr int GA = BM / (8 * NW), GB = BN / (8 * NW);         // 8-row groups per wave
  constexpr int TILE_Q = BM * 128, BUF = (BM + BN) * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const int t = xcd_remap(blockIdx.x, p.mtiles * p.ntiles);
  const int mt = p.nt_major ? t % p.mtiles : t / p.ntiles;
  const int nt = p.nt_major ? t / p.mtiles : t - mt * p.ntiles;
  const int cls = blockIdx.z % p.nclass, ks = blockIdx.z / p.nclass;
  const int m0 = mt * BM, n0 = nt * BN;
  const int s_begin = ks * p.steps_per_split;
  const int s_end = min(p.nsteps, s_begin + p.steps_per_split);
  const int ooy = p.ooy + (p.nclass == 4 ? (cls >> 1) : 0);
  const int oox = p.oox + (p.nclass == 4 ? (cls & 1) : 0);
  const int Hv = p.ups ? 2 * p.Hin : p.Hin, Wv = p.ups ? 2 * p.Win : p.Win;
  const int HoWo = p.Ho * p.Wo;

  const int lrow = lane >> 3;
  const int chunk = (lane & 7) ^ ((4 * (wid & 1) + (lane >> 4)) & 7);   // (row >> 1) & 7 of this lane's rows

  int by[GA], bx[GA], ib[GA];
#pragma unroll
  for (int j = 0; j < GA; ++j) {
    const int m = m0 + (j * NW + wid) * 8 + lrow;
    if (m < p.M) {
      int b, oy, ox;
      if (p.howo_shift >= 0) { b = m >> p.howo_shift; const int r = m & (HoWo - 1); oy = r >> p.wo_shift; ox = r & (p.Wo - 1); }
      else { b = m / HoWo; const int r = m - b * HoWo; oy = r / p.Wo; ox = r - oy * p.Wo; }
      by[j] = oy * p.S + p.dy0; bx[j] = ox * p.S + p.dx0; ib[j] = b * p.Hin * p.Win;
    } else { by[j] = 0; bx[j] = 0; ib[j] = -1; }
  }
  int k0 = s_begin * 64;
  int tap = k0 / p.Cin, ci = k0 - tap * p.Cin;
  int ty = tap / p.TW, tx = tap - ty * p.TW;
  const char* aptr[GA]; unsigned ainc[GA];
  auto compute_ptrs = [&]() {
    const int oy_ = ty * p.dys, ox_ = tx * p.dxs;
    const bool second = ci >= p.c0;
    const char* src = second ? p.in1 + (size_t)(ci - p.c0 + chunk * 8) * 2 : p.in0 + (size_t)(ci + chunk * 8) * 2;
    const size_t ps = (size_t)(second ? p.ps1 : p.ps0) * 2;
#pragma unroll
    for (int j = 0; j < GA; ++j) {
      int u = by[j] + oy_, v = bx[j] + ox_;
      bool ok = ib[j] >= 0;
      if (p.border == CSMRI_BORDER_REFLECT) { u = reflect_idx(u, Hv); v = reflect_idx(v, Wv); }
      else ok = ok && (unsigned)u < (unsigned)Hv && (unsigned)v < (unsigned)Wv;
      if (p.ups) { u >>= 1; v >>= 1; }
      const int pix = ib[j] + u * p.Win + v;
      // byte offset: one 32-bit multiply when the tensor is < 2 GiB (host check), else 64-bit
      const char* g = p.off32 ? src + (unsigned)pix * (unsigned)ps : src + (size_t)pix * ps;
      aptr[j] = ok ? g : g256_zero_page;
      ainc[j] = ok ? 128u : 0u;
    }
  };
  compute_ptrs();
  const char* wptr[GB];
#pragma unroll
  for (int j = 0; j < GB; ++j)
    wptr[j] = p.w + ((size_t)cls * (size_t)p.wcs + (size_t)(n0 + (j * NW + wid) * 8 + lrow) * p.Kp + chunk * 8) * 2 +
              (size_t)s_begin * 128;

  auto issue_a = [&](char* buf) {
#pragma unroll
    for (int j = 0; j < GA; ++j) {
      __builtin_amdgcn_global_load_lds((gptr_t)aptr[j], (lptr_t)(buf + (j * NW + wid) * 1024), 16, 0, 0);
      aptr[j] += ainc[j];
    }
  };
  auto issue_b = [&](char* buf) {
#pragma unroll
    for (int j = 0; j < GB; ++j) {
      __builtin_amdgcn_global_load_lds((gptr_t)wptr[j], (lptr_t)(buf + TILE_Q + (j * NW + wid) * 1024), 16, 0, 0);
      wptr[j] += 128;
    }
  };
  auto issue = [&](char* buf) { issue_a(buf); issue_b(buf); };
  auto advance = [&]() {          // after every issue: on to the next 64 channels / tap / concat source
    ci += 64;
    if (ci == p.Cin) { ci = 0; if (++tx == p.TW) { tx = 0; ++ty; } compute_ptrs(); }
    else if (ci == p.c0) compute_ptrs();
  };

  f32x4_t acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int r16 = lane & 15, g = lane >> 4;

  auto load_frags = [&](const char* buf, int kc, bf16x8_t* pf, bf16x8_t* qf) {
#pragma unroll
    for (int i = 0; i < FN; ++i)
      pf[i] = *(const bf16x8_t*)(buf + TILE_Q + g256_off(wn * WTN + i * 16 + r16, kc * 4 + g));
#pragma unroll
    for (int j = 0; j < FM; ++j)
      qf[j] = *(const bf16x8_t*)(buf + g256_off(wm * WTM + j * 16 + r16, kc * 4 + g));
  };
  auto mma = [&](const bf16x8_t* pf, const bf16x8_t* qf) {
    // the second wave of each SIMD (w + 4) loses the MFMA pipe to the older one and arrives ~1100 cycles
    // late at every barrier (in-kernel stamps): give it the higher priority
    if (wid >= 4) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int j = 0; j < FM; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[i], qf[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  // (tried: K order (64-channel chunk, tap) instead of (tap, chunk), so that the shifted re-reads of
  //  the same input rows hit in L2 -- PMC shows ~3x the input fetched from HBM on conv3_x -- but the
  //  per-step pointer recomputation costs more than it saves: 793 vs 900 TFLOP/s.)
  // two LDS buffers: step s+1 streams in while step s is multiplied; one barrier per step.  The
  // first fragment reads of a step are issued BEFORE the next step's LDS-DMA (address updates +
  // 8 DMA instructions per thread), so their latency runs under that issue work.
#ifdef CSMRI_DBG_STAMPS
  // diagnostic build: per-wave cycle sums of the loop phases, written to p.slab (host passes a buffer)
  unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, last_t;
#define STAMP(i) do { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
    ph[i] += t_ - last_t; last_t = t_; } while (0)
  { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last_t) :: "memory"); }
#else
#define STAMP(i) do {} while (0)
#endif
  if (s_begin < s_end) { issue(smem); advance(); }
  STAMP(2);
  for (int s = s_begin; s < s_end; ++s) {
    const int par = (s - s_begin) & 1;
    const char* cur = smem + par * BUF;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    STAMP(0);
    bf16x8_t pf[FN], qf[FM];
    load_frags(cur, 0, pf, qf);
    STAMP(1);
    // the next step's DMA in two halves, one in front of each MFMA block: a wave is never stuck behind
    // eight DMA issues (in-kernel stamps: 640 of 3750 cycles per step when issued in one burst)
    if (s + 1 < s_end) issue_a(smem + (par ^ 1) * BUF);
    STAMP(2);
    mma(pf, qf);
    STAMP(3);
    load_frags(cur, 1, pf, qf);
    if (s + 1 < s_end) issue_b(smem + (par ^ 1) * BUF);
    STAMP(4);
    mma(pf, qf);
    STAMP(5);
    if (s + 1 < s_end) advance();
  }
#ifdef CSMRI_DBG_STAMPS
  if (lane == 0 && p.slab && p.splitk == 1) {
    unsigned long long* dbg = (unsigned long long*)p.slab + ((size_t)blockIdx.x * NW + wid) * 8;
#pragma unroll
    for (int i = 0; i < 6; ++i) dbg[i] = ph[i];
  }
#endif

  // ---- epilogue (gconv_glds.hip's, without the BatchNorm partial sums) -----------------------
#pragma unroll
  for (int j = 0; j < FM; ++j) {
    const int m = m0 + wm * WTM + j * 16 + r16;
    const bool mv = m < p.M;
    OutPos op; op.base = p.out; op.opix = 0; op.gpix = 0; op.g_ok = true;
    if (mv) {
      if (p.dense_out) {               // position index == m: no decomposition
        op.opix = (size_t)m * p.ops; op.gpix = (size_t)m * p.gps;
      } else {
        int b, oy, ox;
        if (p.howo_shift >= 0) { b = m >> p.howo_shift; const int r = m & (HoWo - 1); oy = r >> p.wo_shift; ox = r & (p.Wo - 1); }
        else { b = m / HoWo; const int r = m - b * HoWo; oy = r / p.Wo; ox = r - oy * p.Wo; }
        op = gconv_out_pos(p, b, oy * p.osy + ooy, ox * p.osx + oox);
      }
    }
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      const int n = n0 + wn * WTN + i * 16 + g * 4;
      f32x4_t v = acc[i][j];
      if (p.splitk > 1) {
        if (mv) *(f32x4_t*)(p.slab + (((size_t)cls * p.splitk + ks) * p.M + m) * p.Cout + n) = v;
        continue;
      }
      if (!mv) continue;
      if (p.bias) { f32x4_t bb = *(const f32x4_t*)(p.bias + n); v += bb; }
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
}

// ---------------------------------------------------------------------------------------------
// (measured: 256 x 128 tiles -- 8 waves of 64 x 64 -- lose to the 128-row kernels everywhere: only <256> is built)
static int g256_bn(const csmri_gconv_desc* d) { (void)d; return 256; }

static long long g256_tiles(const csmri_gconv_desc* d) {
  const int nclass = d->nclass > 0 ? d->nclass : 1;
  return (long long)cdiv((long long)d->B * d->Ho * d->Wo, 256) * (d->Cout / g256_bn(d)) * nclass;
}

// Measured (tools/bench_conv.py): the 256-row tiles win where they fill the chip WITHOUT split-K
// (one workgroup per CU: at least ~224 tiles); below that the 128-row kernels with their slabs do better.
int gconv_glds256_eligible(const csmri_gconv_desc* d) {
  if (!gconv_glds_eligible(d)) return 0;
  if (d->Cout % g256_bn(d) || d->stats_partial) return 0;
  return g256_tiles(d) >= 224;
}

int gconv_glds256_splitk(const csmri_gconv_desc* d) { (void)d; return 1; }

const char* gconv_glds256_name(const csmri_gconv_desc* d) {
  (void)d;
  return "gconv_glds256_kernel<256>";
}

int gconv_glds256_launch(const GParams& p0, const csmri_gconv_desc* d, hipStream_t st) {
  GParams p = p0;
  p.nsteps = d->TH * d->TW * d->Cin / 64;
  p.steps_per_split = cdiv(p.nsteps, p.splitk);
  const int bn = g256_bn(d);
  p.mtiles = cdiv(p.M, 256); p.ntiles = d->Cout / bn;
  const long long w_elems = (long long)d->Cout * d->TH * d->TW * d->Cin * p.nclass;
  const long long x_elems = (long long)d->B * d->Hin * d->Win * d->Cin;
  p.nt_major = w_elems > x_elems;
  CSMRI_SET_MAX_LDS(gconv_glds256_kernel<256>, 2 * (256 + 256) * 128);
  dim3 grid(p.mtiles * p.ntiles, 1, p.nclass * p.splitk);
  hipLaunchKernelGGL(gconv_glds256_kernel<256>, grid, dim3(512), 2 * (256 + 256) * 128, st, p);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}
