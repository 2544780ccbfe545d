// gconv8p: implicit-GEMM gather convolution on a 256 x 256 x 64 tile, 8 waves, as a PHASED pipeline:
// every 64-deep K tile is four phases of 16 MFMAs (one quadrant of the wave's 128 x 64 outputs), the two
// waves of each SIMD run one barrier interval apart (while one multiplies, the other reads fragments and
// issues LDS-DMA), and the LDS-DMA of the tile AFTER next stays in flight across the barriers, retired by
// counted s_waitcnt vmcnt -- the loop never drains the memory pipe.  (Structure: the 8-phase GEMM schedule of
// cdna_hip_programming.md section 5, rebuilt around this library's gather addressing; the kernel it
// replaces, gconv_glds256, waits for vmcnt(0) at every K tile.)
//
// GEMM view as in gconv_glds.hip: D[n][m] = sum_k W[n][k] X[m][k], m = output position (gathered from the
// NHWC input: zero / reflection borders, x2 upsampling, two-source concat), n = output channel,
// k = (tap, channel).  LDS: two K-tile buffers of 64 KiB, each 256 position rows + 256 channel rows of
// 128 bytes, 16-byte slots XOR-swizzled on the SOURCE side of the LDS-DMA (slot ^= (row >> 1) & 7).
//
// Half-tiles.  A wave (wm, wn) owns positions wm*128..+128 and channels wn*64..+64.  "X-lo" are the first
// 64 positions of both wm, "X-hi" the others; "W-lo" the first 32 channels of every wn, "W-hi" the others.
// Each half-tile is 16 KiB = 2 LDS-DMA instructions per thread.  Phases of K tile t (buffer t & 1):
//   P1  reads X-lo, W-lo   multiplies (lo, lo)   stages W-hi of tile t+1
//   P2  reads W-hi         multiplies (lo, hi)   stages X-hi of tile t+1
//   P3  reads X-hi         multiplies (hi, hi)   stages X-lo of tile t+2   (its region was last read in P1)
//   P4  reads nothing      multiplies (hi, lo)   stages W-lo of tile t+2   (W-lo fragments stay in registers)
// (a phase's stage is issued in the middle of its MFMA block: cheapest place for the issue, see G8P_MMA)
// Hazards (the two wave groups are one barrier interval apart, so "a phase later" is two intervals):
//   WAR  a region is re-staged at least two phases after its last fragment read;
//   RAW  a region is read one phase after the phase whose leading s_waitcnt vmcnt(6) retires it in EVERY wave
//        (6 = the three half-tiles issued after it), with the phase's barriers in between.
// Past the last K tile the stage slots keep issuing (from a zero page, into regions nobody reads again), so
// the counted waits never need a tail variant.
#include "mma_core.h"
#include "gconv_params.h"

// rows outside the image / past the last K tile read zeros.  Their pointers advance by 128 B per K tile like
// every other row's (no per-row increment register); a pointer is recomputed at least every Cin/64 <= 16
// tiles, so 4 KiB of zeros cover any run
__device__ __attribute__((aligned(16))) char g8p_zero_page[4096];

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ int g8p_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

#define G8P_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define G8P_LGKM0() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

__global__ __launch_bounds__(512, 2) void gconv8p_kernel(const GParams p) {
  constexpr int BM = 256, BN = 256;
  constexpr int TILE_X = BM * 128, BUF = (BM + BN) * 128;      // bytes
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;
  const int t = xcd_remap(blockIdx.x, p.mtiles * p.ntiles);
  const int mt = p.nt_major ? t % p.mtiles : t / p.ntiles;
  const int nt = p.nt_major ? t / p.mtiles : t - mt * p.ntiles;
  const int cls = blockIdx.z % p.nclass, ks = blockIdx.z / p.nclass;
  const int m0 = mt * BM, n0 = nt * BN;
  const int s_begin = ks * p.steps_per_split;
  const int s_end = min(p.nsteps, s_begin + p.steps_per_split);
  const int T = s_end - s_begin;
  const int ooy = p.ooy + (p.nclass == 4 ? (cls >> 1) : 0);
  const int oox = p.oox + (p.nclass == 4 ? (cls & 1) : 0);
  const int Hv = p.ups ? 2 * p.Hin : p.Hin, Wv = p.ups ? 2 * p.Win : p.Win;
  const int HoWo = p.Ho * p.Wo;

  // ---- LDS-DMA addressing: this thread's rows of the four half-tiles ---------------------------------
  const int lrow = lane >> 3, slot = lane & 7;
  // rows of a wave's two pieces (8 rows each) inside a half-tile; the swizzle term of those rows depends on
  // (piece, lrow) only: every other row-offset below is a multiple of 16
  const int xrow_base = (wid >> 2) * 128 + (wid & 3) * 16;          // + hi*64 + j*8 + lrow
  const int wrow_base = (wid >> 1) * 64 + (wid & 1) * 16;           // + hi*32 + j*8 + lrow
  int by[2][2], bx[2][2], ib[2][2];                                 // [hi][j]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = m0 + xrow_base + h * 64 + j * 8 + lrow;
      if (m < p.M) {
        int b, oy, ox;
        if (p.howo_shift >= 0) { b = m >> p.howo_shift; const int r = m & (HoWo - 1); oy = r >> p.wo_shift; ox = r & (p.Wo - 1); }
        else { b = m / HoWo; const int r = m - b * HoWo; oy = r / p.Wo; ox = r - oy * p.Wo; }
        by[h][j] = oy * p.S + p.dy0; bx[h][j] = ox * p.S + p.dx0; ib[h][j] = b * p.Hin * p.Win;
      } else { by[h][j] = 0; bx[h][j] = 0; ib[h][j] = -1; }
    }
  // K position of the next X half-tile to stage, per half (the two halves are staged two phases apart)
  int xk[2], xci[2], xty[2], xtx[2];
  const char* aptr[2][2];
  int wk[2];
  // fast tap change (zero border, no upsampling, one source -- every VGG layer): the row's address for tap
  // (0, 0), channel 0 is fixed; a tap adds a wave-uniform byte offset and re-tests the border.  The general
  // rule below costs ~150 vector instructions per call, which in this loop lengthens a barrier interval of
  // all eight waves (in-kernel stamps: the intervals holding it ran 2-3x the others)
  const bool fast = p.border == CSMRI_BORDER_ZERO && !p.ups && p.in1 == nullptr;
  // K order.  Default: K tile s = (tap, 64-channel chunk), tap-major, as the weights are packed.  TAP-INNER (fast
  // path, several taps): s = (chunk, tap) -- consecutive K tiles re-read the SAME input rows shifted by one
  // pixel, which hit in L2, instead of sweeping the whole receptive field once per tap out of the Infinity
  // Cache (the counters showed ~3x the input fetched per 3x3 layer, and the stamps the X stages as the long
  // phases).  The weight K offset follows: (tap * Cin + chunk * 64) elements.
  const int ntaps = p.TH * p.TW;
  const bool ti = fast && ntaps > 1 && p.tap_inner;
  const char* rowptr[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int chunk = slot ^ ((j * 4 + (lrow >> 1)) & 7);
      const long long pix = (long long)ib[h][j] + (long long)by[h][j] * p.Win + bx[h][j];
      rowptr[h][j] = p.in0 + (pix * p.ps0 + chunk * 8) * 2;
    }
  auto x_ptrs = [&](int h) {
    const bool live = xk[h] < s_end;
    const int oy_ = xty[h] * p.dys, ox_ = xtx[h] * p.dxs;
    if (fast) {
      const long long delta = ((long long)(oy_ * p.Win + ox_) * p.ps0 + xci[h]) * 2;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bool ok = live & (ib[h][j] >= 0) & ((unsigned)(by[h][j] + oy_) < (unsigned)p.Hin) &
                        ((unsigned)(bx[h][j] + ox_) < (unsigned)p.Win);
        aptr[h][j] = ok ? rowptr[h][j] + delta : g8p_zero_page;
      }
      return;
    }
    const bool second = xci[h] >= p.c0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int chunk = slot ^ ((j * 4 + (lrow >> 1)) & 7);
      const char* src = second ? p.in1 + (size_t)(xci[h] - p.c0 + chunk * 8) * 2 : p.in0 + (size_t)(xci[h] + chunk * 8) * 2;
      const size_t ps = (size_t)(second ? p.ps1 : p.ps0) * 2;
      int u = by[h][j] + oy_, v = bx[h][j] + ox_;
      bool ok = live && ib[h][j] >= 0;
      if (p.border == CSMRI_BORDER_REFLECT) { u = reflect_idx(u, Hv); v = reflect_idx(v, Wv); }
      else ok = ok && (unsigned)u < (unsigned)Hv && (unsigned)v < (unsigned)Wv;
      if (p.ups) { u >>= 1; v >>= 1; }
      const int pix = ib[h][j] + u * p.Win + v;
      const char* g = p.off32 ? src + (unsigned)pix * (unsigned)ps : src + (size_t)pix * ps;
      aptr[h][j] = ok ? g : g8p_zero_page;
    }
  };
  const char* wrow[2][2]; long long wko[2];      // weight rows at K = 0; byte offset of the next K tile to stage
  const long long wtap_step = (long long)p.Cin * 2, wwrap = (long long)(ntaps - 1) * p.Cin * 2 - 128;
  int wtap[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    xk[h] = s_begin;
    if (ti) { const int ch = s_begin / ntaps, tap = s_begin - ch * ntaps; xci[h] = ch * 64; xty[h] = tap / p.TW; xtx[h] = tap - xty[h] * p.TW; }
    else { const int k0 = s_begin * 64, tap = k0 / p.Cin; xci[h] = k0 - tap * p.Cin; xty[h] = tap / p.TW; xtx[h] = tap - xty[h] * p.TW; }
    x_ptrs(h);
    wk[h] = s_begin;
    wtap[h] = ti ? s_begin % ntaps : 0;
    wko[h] = ti ? ((long long)wtap[h] * p.Cin + (long long)(s_begin / ntaps) * 64) * 2 : (long long)s_begin * 128;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int chunk = slot ^ ((j * 4 + (lrow >> 1)) & 7);
      wrow[h][j] = p.w + ((size_t)cls * (size_t)p.wcs + (size_t)(n0 + wrow_base + h * 32 + j * 8 + lrow) * p.Kp + chunk * 8) * 2;
    }
  }
  auto stage_x = [&](int h, char* buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      __builtin_amdgcn_global_load_lds((gptr_t)aptr[h][j], (lptr_t)(buf + (xrow_base + h * 64 + j * 8) * 128), 16, 0, 0);
      aptr[h][j] += 128;
    }
    ++xk[h];
    if (ti) {
      if (++xtx[h] == p.TW) { xtx[h] = 0; if (++xty[h] == p.TH) { xty[h] = 0; xci[h] += 64; } }
      x_ptrs(h);
      return;
    }
    xci[h] += 64;
    if (xci[h] == p.Cin) { xci[h] = 0; if (++xtx[h] == p.TW) { xtx[h] = 0; ++xty[h]; } x_ptrs(h); }
    else if (xci[h] == p.c0 || xk[h] == s_end) x_ptrs(h);       // second concat source / past the end: zero page
  };
  auto stage_w = [&](int h, char* buf) {
    const bool live = wk[h] < s_end;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const char* src = live ? wrow[h][j] + wko[h] : g8p_zero_page;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(buf + TILE_X + (wrow_base + h * 32 + j * 8) * 128), 16, 0, 0);
    }
    ++wk[h];
    if (ti) { if (++wtap[h] == ntaps) { wtap[h] = 0; wko[h] -= wwrap; } else wko[h] += wtap_step; }
    else wko[h] += 128;
  };

  // ---- fragments -------------------------------------------------------------------------------------
  const int r16 = lane & 15, g = lane >> 4;
  f32x4_t acc[4][8];                       // [n fragment][m fragment]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  bf16x8_t wl[2][2], wh[2][2], xf[4][2];      // [fragment][k half]; X-lo and X-hi fragments share registers
  auto read_x = [&](const char* buf, int h, bf16x8_t (*q)[2]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int kc = 0; kc < 2; ++kc)
        q[j][kc] = *(const bf16x8_t*)(buf + g8p_off(wm * 128 + h * 64 + j * 16 + r16, kc * 4 + g));
  };
  auto read_w = [&](const char* buf, int h, bf16x8_t (*q)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int kc = 0; kc < 2; ++kc)
        q[i][kc] = *(const bf16x8_t*)(buf + TILE_X + g8p_off(wn * 64 + h * 32 + i * 16 + r16, kc * 4 + g));
  };
// 16 MFMAs of one quadrant; STAGE (the phase's two LDS-DMA instructions + pointer updates) is issued after the
// first eight: among MFMAs an LDS-DMA issue costs ~60 cycles of the wave, in front of the barrier next to the
// fragment reads 100-185 (in-kernel stamps: the load sections, not the MFMA sections, set the interval length)
#define G8P_MMA(wq, xq, i0, j0, STAGE)                                                                       \
  do {                                                                                                       \
    __builtin_amdgcn_s_setprio(1);                                                                           \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                            \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                            \
      acc[i0 + i][j0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[i][0], xq[j][0], acc[i0 + i][j0 + j], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    STAGE;                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                            \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                            \
      acc[i0 + i][j0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[i][1], xq[j][1], acc[i0 + i][j0 + j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                           \
  } while (0)
#define G8P_BAR() __builtin_amdgcn_s_barrier()

#ifdef CSMRI_DBG_STAMPS
  // diagnostic build: per-wave cycle sums [phase 0..3][segment 0..4] written to p.slab (host passes a buffer):
  // segment 0 = wait + fragment reads + stage issue, 1 = first barrier, 2 = lgkmcnt wait, 3 = MFMAs, 4 = second barrier
  unsigned long long ph[4][5] = {}, last_t;
#define STAMP(P, S) do { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
    ph[P][S] += t_ - last_t; last_t = t_; } while (0)
#else
#define STAMP(P, S) do {} while (0)
#endif
  // ---- prologue: tile 0 complete, X-lo / W-lo of tile 1 ------------------------------------------------
  char* b0 = smem; char* b1 = smem + BUF;
  stage_x(0, b0); stage_w(0, b0); stage_w(1, b0); stage_x(1, b0);
  stage_x(0, b1); stage_w(0, b1);
  G8P_VMCNT(8);                              // X-lo, W-lo of tile 0 (this wave's pieces)
  G8P_BAR();
  if (wm == 1) G8P_BAR();                    // second wave of every SIMD: one barrier interval behind
  char* cur = b0; char* nxt = b1;
#ifdef CSMRI_DBG_STAMPS
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last_t) :: "memory");
#endif
  for (int tt = 0; tt < T; ++tt) {
    // P1
    G8P_VMCNT(6);                            // retires W-hi(t)
    read_x(cur, 0, xf); read_w(cur, 0, wl);
    STAMP(0, 0); G8P_BAR(); STAMP(0, 1); G8P_LGKM0(); STAMP(0, 2);
    G8P_MMA(wl, xf, 0, 0, stage_w(1, nxt));                        // W-hi(t+1)
    STAMP(0, 3); G8P_BAR(); STAMP(0, 4);
    // P2
    G8P_VMCNT(6);                            // retires X-hi(t)
    read_w(cur, 1, wh);
    STAMP(1, 0); G8P_BAR(); STAMP(1, 1); G8P_LGKM0(); STAMP(1, 2);
    G8P_MMA(wh, xf, 2, 0, stage_x(1, nxt));                        // X-hi(t+1)
    STAMP(1, 3); G8P_BAR(); STAMP(1, 4);
    // P3
    read_x(cur, 1, xf);
    STAMP(2, 0); G8P_BAR(); STAMP(2, 1); G8P_LGKM0(); STAMP(2, 2);
    G8P_MMA(wh, xf, 2, 4, stage_x(0, cur));                        // X-lo(t+2)
    STAMP(2, 3); G8P_BAR(); STAMP(2, 4);
    // P4
    G8P_VMCNT(6);                            // retires X-lo(t+1), W-lo(t+1)
    STAMP(3, 0); G8P_BAR(); STAMP(3, 1); STAMP(3, 2);
    G8P_MMA(wl, xf, 0, 4, stage_w(0, cur));                        // W-lo(t+2)
    STAMP(3, 3); G8P_BAR(); STAMP(3, 4);
    char* sw = cur; cur = nxt; nxt = sw;
  }
  if (wm == 0) G8P_BAR();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // trailing (zero-page) DMA must not outlive the workgroup's LDS
#ifdef CSMRI_DBG_STAMPS
  if (lane == 0 && p.slab && p.splitk == 1) {
    unsigned long long* dbg = (unsigned long long*)p.slab + ((size_t)blockIdx.x * 8 + wid) * 20;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 5; ++b) dbg[a * 5 + b] = ph[a][b];
  }
#endif

  // ---- epilogue (gconv_glds256's) ----------------------------------------------------------------------
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int m = m0 + wm * 128 + j * 16 + r16;
    const bool mv = m < p.M;
    OutPos op; op.base = p.out; op.opix = 0; op.gpix = 0; op.g_ok = true;
    if (mv) {
      if (p.dense_out) {
        op.opix = (size_t)m * p.ops; op.gpix = (size_t)m * p.gps;
      } else {
        int b, oy, ox;
        if (p.howo_shift >= 0) { b = m >> p.howo_shift; const int r = m & (HoWo - 1); oy = r >> p.wo_shift; ox = r & (p.Wo - 1); }
        else { b = m / HoWo; const int r = m - b * HoWo; oy = r / p.Wo; ox = r - oy * p.Wo; }
        op = gconv_out_pos(p, b, oy * p.osy + ooy, ox * p.osx + oox);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = n0 + wn * 64 + i * 16 + g * 4;
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

// ---------------------------------------------------------------------------------------------------------
static long long g8p_tiles(const csmri_gconv_desc* d) {
  const int nclass = d->nclass > 0 ? d->nclass : 1;
  return (long long)cdiv((long long)d->B * d->Ho * d->Wo, 256) * (d->Cout / 256) * nclass;
}

// EXPERIMENTAL, off by default (CSMRI_8P=1 enables it; tools/check_8p.py, tools/stamp_8p.py).  Measured on
// MI355X (round 2): 870-920 TFLOP/s on VGG conv3_2 at batch 16 against 940-950 for
// gconv_glds256 and slower than the 128-row kernels wherever K has to be split -- the barrier intervals that
// carry an X stage run 2-3x the others wherever the stage is issued, because a CU takes in rows that miss L2
// at 23-34 GB/s (MI355X_MICROARCH.md, gather table) and a 256 x 256 x 64 tile needs 64 KiB per microsecond.
int gconv8p_eligible(const csmri_gconv_desc* d) {
  static const char* on = getenv("CSMRI_8P");                  // opt-in
  static const char* off = getenv("CSMRI_NO_8P");              // A/B knob
  if (!on || atoi(on) == 0 || off) return 0;
  if (!gconv_glds_eligible(d)) return 0;
  if (d->Cout % 256 || d->stats_partial) return 0;
  static const char* mint = getenv("CSMRI_8P_MIN_TILES");        // A/B knob
  return g8p_tiles(d) >= (mint ? atoi(mint) : 8);
}

// K split so that the grid reaches about one workgroup per CU (one 8-wave workgroup owns a CU: 128 KiB LDS)
int gconv8p_splitk(const csmri_gconv_desc* d) {
  const long long tiles = g8p_tiles(d);
  const int nsteps = d->TH * d->TW * d->Cin / 64;
  static const char* tgt_env = getenv("CSMRI_8P_BLOCKS");       // tuning knob: target workgroups
  const int target = tgt_env ? atoi(tgt_env) : 256;
  if (tiles >= (target * 3) / 4) return 1;
  int sk = (int)((target + tiles / 2) / tiles);
  static const char* mins_env = getenv("CSMRI_8P_MIN_STEPS");     // tuning knob: fewest K tiles per split
  const int min_steps = mins_env ? atoi(mins_env) : 8;
  int maxsk = nsteps / min_steps; if (maxsk < 1) maxsk = 1;
  if (sk > maxsk) sk = maxsk;
  if (sk > 32) sk = 32;
  return sk < 1 ? 1 : sk;
}

int gconv8p_launch(const GParams& p0, const csmri_gconv_desc* d, hipStream_t st) {
  GParams p = p0;
  p.nsteps = d->TH * d->TW * d->Cin / 64;
  p.steps_per_split = cdiv(p.nsteps, p.splitk);
  p.mtiles = cdiv(p.M, 256); p.ntiles = d->Cout / 256;
  const long long w_elems = (long long)d->Cout * d->TH * d->TW * d->Cin * p.nclass;
  const long long x_elems = (long long)d->B * d->Hin * d->Win * d->Cin;
  p.nt_major = w_elems > x_elems;
  static const char* ti_env = getenv("CSMRI_8P_TAP_INNER");      // A/B knob
  p.tap_inner = ti_env ? atoi(ti_env) : 0;      // measured slower than the tap-major order (755 vs 870 TFLOP/s)
  constexpr int lds = 2 * (256 + 256) * 128;
  CSMRI_SET_MAX_LDS(gconv8p_kernel, lds);
  dim3 grid(p.mtiles * p.ntiles, 1, p.nclass * p.splitk);
  hipLaunchKernelGGL(gconv8p_kernel, grid, dim3(512), lds, st, p);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}
