// gconv_glds: the implicit-GEMM convolution of gconv.hip with the operand tiles staged by
// LDS-DMA (global_load_lds_dwordx4) instead of through registers.
//
//   D[n][m] = sum_k W[n][k] * X[m][k],  k = (tap, ci),  bf16 operands, fp32 accumulate
//
// Tile: 128 positions x BN channels x 64 K per step, 2 x 2 waves (256 threads) or, for the two-buffer 128-channel
// variant, 2 x 4 waves (512 threads: each wave issues half the LDS-DMA, see gconv_glds_launch).  One LDS buffer
// (16 KiB + BN*128 B): a step is {8 or 6 LDS-DMA per thread -> vmcnt(0) -> barrier -> 2x(8 or 6
// ds_read_b128 + 16 or 8 MFMA) -> barrier}; 3-4 workgroups per CU overlap one another's loads
// and MFMAs.  An LDS-DMA wave-instruction writes 1 KiB linearly (8 rows x 128 B), so the XOR
// swizzle that keeps the ds_read_b128 fragment reads conflict-free (16-byte slot ^= (row>>1)&7)
// is applied on the SOURCE side: the lane that owns slot s of row r fetches K-chunk
// s ^ ((r>>1)&7) of that row.  Out-of-image taps (zero border), tile rows past M and taps past
// the filter fetch a 16-byte zero page.
// Needs Cin % 64 == 0 (a K step never straddles a tap) and Cout % BN == 0; everything else goes
// to gconv.hip.  Epilogue (bias / leaky / act-derivative / BN partial sums / split-K slab) is the
// one of gconv.hip.
#include <algorithm>
#include <type_traits>
#include "mma_core.h"
#include "gconv_params.h"

__device__ __attribute__((aligned(16))) char g_zero_page[16];

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ int glds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// NW = 8: the same tile on 2 x 4 waves, every wave issuing half the LDS-DMA (two-buffer 128-channel variant)
template <int BN, int NST, int NW = 4>
__global__ __launch_bounds__(64 * NW, NST == 1 ? 3 : (NST == 2 ? 2 : 1)) void gconv_glds_kernel(const GParams p) {
  constexpr int BM = 128, WM = 2, WN = NW / 2;
  constexpr int WTM = BM / WM, WTN = BN / WN, FM = WTM / 16, FN = WTN / 16;
  constexpr int GA = 16 / NW, GB = BN / (8 * NW);      // 8-row groups per wave: positions / weights
  constexpr int TILE_Q = BM * 128, BUF = (BM + BN) * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const int t = xcd_remap(blockIdx.x, p.mtiles * p.ntiles);
  // an XCD owns a contiguous run of t: make the operand that is re-read across that run the small one
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

  // this lane's slot in every 8-row group it fills, and the K chunk that belongs there
  const int lrow = lane >> 3;
  const int chunk = (lane & 7) ^ ((4 * (wid & 1) + (lane >> 4)) & 7);

  // Source pixel of every (filter tap, tile row), computed ONCE into an LDS table (-1 = out of the image: zero page).  The
  // K loop used to rebuild them per wave at every tap change -- position decomposition, border rule, two multiplies per
  // row: with 128 input channels (a tap change every 2 steps) the hardware counters showed 3.8-7.9 vector instructions per
  // MFMA in these kernels against ~1 in the steady-state loop.
  int* const tab = (int*)(smem + NST * BUF);
  const int ntaps = p.TH * p.TW;
  {
    const int row = tid & 127, m = m0 + row;
    int b = 0, oy = 0, ox = 0;
    const bool mv = m < p.M;
    if (mv) {
      if (p.howo_shift >= 0) { b = m >> p.howo_shift; const int r = m & (HoWo - 1); oy = r >> p.wo_shift; ox = r & (p.Wo - 1); }
      else { b = m / HoWo; const int r = m - b * HoWo; oy = r / p.Wo; ox = r - oy * p.Wo; }
    }
    const int by = oy * p.S + p.dy0, bx = ox * p.S + p.dx0, ibase = b * p.Hin * p.Win;
    // (a split-K slice only visits the taps of its own K range)
    const int tap_lo = (s_begin * 64) / p.Cin, tap_hi = min(ntaps - 1, (max(s_end, s_begin + 1) * 64 - 1) / p.Cin);
    for (int tp = tap_lo + (tid >> 7); tp <= tap_hi; tp += (64 * NW) >> 7) {
      const int tyy = tp / p.TW, txx = tp - tyy * p.TW;
      int u = by + tyy * p.dys, v = bx + txx * p.dxs;
      bool ok = mv;
      if (p.border == CSMRI_BORDER_REFLECT) { u = reflect_idx(u, Hv); v = reflect_idx(v, Wv); }
      else ok = ok && (unsigned)u < (unsigned)Hv && (unsigned)v < (unsigned)Wv;
      if (p.ups) { u >>= 1; v >>= 1; }
      tab[tp * 128 + row] = ok ? ibase + u * p.Win + v : -1;
    }
  }
  __syncthreads();
  int k0 = s_begin * 64;
  int tap = k0 / p.Cin, ci = k0 - tap * p.Cin;
  int ty = tap / p.TW, tx = tap - ty * p.TW;
  const char* aptr[GA]; unsigned ainc[GA];
  auto compute_ptrs = [&]() {
    const bool second = ci >= p.c0;                     // wave-uniform: c0 % 64 == 0
    const char* src = second ? p.in1 + (size_t)(ci - p.c0 + chunk * 8) * 2 : p.in0 + (size_t)(ci + chunk * 8) * 2;
    const size_t ps = (size_t)(second ? p.ps1 : p.ps0) * 2;
    const int tp = ty * p.TW + tx;
#pragma unroll
    for (int j = 0; j < GA; ++j) {
      const int pix = tab[tp * 128 + (j * NW + wid) * 8 + lrow];
      const bool ok = pix >= 0;
      // byte offset: one 32-bit multiply when the tensor is < 2 GiB (host check), else 64-bit
      const char* g = p.off32 ? src + (unsigned)pix * (unsigned)ps : src + (size_t)pix * ps;
      aptr[j] = ok ? g : g_zero_page;
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
  auto advance = [&]() {
    ci += 64;
    if (ci == p.Cin) { ci = 0; if (++tx == p.TW) { tx = 0; ++ty; } compute_ptrs(); }
    else if (ci == p.c0) compute_ptrs();                // switch to the second concat source
  };
  auto issue = [&](int s, char* buf) {
#pragma unroll
    for (int j = 0; j < GA; ++j) {
      __builtin_amdgcn_global_load_lds((gptr_t)aptr[j], (lptr_t)(buf + (j * NW + wid) * 1024), 16, 0, 0);
      aptr[j] += ainc[j];
    }
#pragma unroll
    for (int j = 0; j < GB; ++j) {
      __builtin_amdgcn_global_load_lds((gptr_t)wptr[j], (lptr_t)(buf + TILE_Q + (j * NW + wid) * 1024), 16, 0, 0);
      wptr[j] += 128;
    }
    ci += 64;
    if (ci == p.Cin) { ci = 0; if (++tx == p.TW) { tx = 0; ++ty; } compute_ptrs(); }
    else if (ci == p.c0) compute_ptrs();                // switch to the second concat source
  };

  f32x4_t acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int r16 = lane & 15, g = lane >> 4;

  auto compute = [&](const char* buf) {
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      bf16x8_t pf[FN], qf[FM];
#pragma unroll
      for (int i = 0; i < FN; ++i)
        pf[i] = *(const bf16x8_t*)(buf + TILE_Q + glds_off(wn * WTN + i * 16 + r16, kc * 4 + g));
#pragma unroll
      for (int j = 0; j < FM; ++j)
        qf[j] = *(const bf16x8_t*)(buf + glds_off(wm * WTM + j * 16 + r16, kc * 4 + g));
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[i], qf[j], acc[i][j], 0, 0, 0);
    }
  };
  auto load_frags = [&](const char* buf, int kc, bf16x8_t* pf, bf16x8_t* qf) {
#pragma unroll
    for (int i = 0; i < FN; ++i)
      pf[i] = *(const bf16x8_t*)(buf + TILE_Q + glds_off(wn * WTN + i * 16 + r16, kc * 4 + g));
#pragma unroll
    for (int j = 0; j < FM; ++j)
      qf[j] = *(const bf16x8_t*)(buf + glds_off(wm * WTM + j * 16 + r16, kc * 4 + g));
  };
  auto mma = [&](const bf16x8_t* pf, const bf16x8_t* qf) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int j = 0; j < FM; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[i], qf[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
#ifdef CSMRI_DBG_STAMPS
  unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, last_t;
#define STAMP(i) do { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
    ph[i] += t_ - last_t; last_t = t_; } while (0)
  { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last_t) :: "memory"); }
#else
#define STAMP(i) do {} while (0)
#endif
  if constexpr (NST >= 3) {
    // ring of NST LDS buffers for grids of at most one workgroup per CU: NST-1 steps of LDS-DMA stay
    // in flight across the (raw) barrier, retired by counted vmcnt -- latency-bound skinny problems
    constexpr int L = GA + GB;                     // LDS-DMA instructions per thread per step
    static_assert(NST == 4, "vmcnt ladder below assumes 2 steps may stay in flight");
    const int total = s_end - s_begin;
    int issued = 0;
    for (; issued < NST - 1 && issued < total; ++issued) issue(s_begin + issued, smem + issued * BUF);
    for (int i = 0; i < total; ++i) {
      const int pending = issued - i - 1;
      if (pending >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * L) : "memory");
      else if (pending == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (issued < total) { issue(s_begin + issued, smem + (issued % NST) * BUF); ++issued; }
      compute(smem + (i % NST) * BUF);
    }
  } else if constexpr (NST == 2) {
    // two LDS buffers: step s+1 streams in while step s is multiplied; one barrier per step
    if (s_begin < s_end) issue(s_begin, smem);
    for (int s = s_begin; s < s_end; ++s) {
      char* cur = smem + ((s - s_begin) & 1) * BUF;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      STAMP(0);
      __syncthreads();
      STAMP(1);
      // first fragment reads ahead of the next step's DMA issue (their latency runs under the address
      // updates and the 8 or 6 DMA instructions), MFMA blocks at raised priority
      bf16x8_t pf[FN], qf[FM];
      char* nxt = smem + (((s - s_begin) & 1) ^ 1) * BUF;
      load_frags(cur, 0, pf, qf);
      if (s + 1 < s_end) issue_a(nxt);                  // the next step's DMA in two halves, one in front
      STAMP(2);
      mma(pf, qf);                                      // of each MFMA block
      STAMP(3);
      load_frags(cur, 1, pf, qf);
      if (s + 1 < s_end) { issue_b(nxt); advance(); }
      STAMP(4);
      mma(pf, qf);
      STAMP(5);
    }
  } else {
    for (int s = s_begin; s < s_end; ++s) {
      issue(s, smem);
      STAMP(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      STAMP(1);
      __syncthreads();
      STAMP(2);
      compute(smem);
      STAMP(3);
      __syncthreads();
      STAMP(4);
    }
  }

#ifdef CSMRI_DBG_STAMPS
  if (lane == 0 && p.slab && p.splitk == 1) {
    unsigned long long* dbg = (unsigned long long*)p.slab + ((size_t)blockIdx.x * 4 + wid) * 8;
#pragma unroll
    for (int i = 0; i < 6; ++i) dbg[i] = ph[i];
  }
#endif
  // ---- epilogue (same contract as gconv_kernel) --------------------------------------------
  // Straight-line since round 5, like pconv2 / tconv / uconv: the outputs leave through range-checked buffer stores (an
  // invalid lane -- tile rows past M, the other tensor of the windowed form -- carries an offset past the descriptor's
  // range and is dropped by the hardware), a fragment row's gate values are loaded as one batch, bf16 fragment pairs are
  // exchanged between lane rows so that a lane stores 16 B.  The old form (gconv_out_pos with the per-lane choice of
  // p.out / p.out2 -- compiled into vector loads of the kernel arguments --, 8-byte stores under exec branches) cost a
  // short-K tile more than its K loop: the stride-2 data gradients of the discriminator's first layers run 8-16 steps
  // per workgroup.
  const int r16e = lane & 15, ge = lane >> 4;
  if (p.splitk > 1) {
    // a split launch stores slab rows indexed by m: no output position needed
#pragma unroll
    for (int j = 0; j < FM; ++j) {
      const int m = m0 + wm * WTM + j * 16 + r16e;
      if (m < p.M) {
        float* row = p.slab + (((size_t)cls * p.splitk + ks) * p.M + m) * p.Cout + n0 + wn * WTN + ge * 4;
#pragma unroll
        for (int i = 0; i < FN; ++i) *(f32x4_t*)(row + i * 16) = acc[i][j];
      }
    }
    return;
  }
  const int es = p.out_dt == CSMRI_F32 ? 4 : 2, ges = p.gdt == CSMRI_F32 ? 4 : 2;
  const unsigned opx = (unsigned)p.B * (p.out2 ? p.win_h * p.win_w : p.Hout_t * p.Wout_t);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)(opx * (unsigned)p.ops * es), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_halo = __builtin_amdgcn_make_buffer_rsrc(p.out2 ? p.out2 : p.out, 0,
      (int)(p.out2 ? (unsigned)p.B * p.Hout_t * p.Wout_t * (unsigned)p.o2ps * es : 0u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_gate = __builtin_amdgcn_make_buffer_rsrc((void*)(p.gsrc ? p.gsrc : p.out), 0,
      (int)(p.gsrc ? opx * (unsigned)p.gps * ges : 0u), 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  static_assert(FN % 2 == 0, "16-byte stores pair two channel fragments");
  const bool bf_out = p.out_dt != CSMRI_F32, has_gate = p.gsrc != nullptr, has_act = p.slope != 1.f, has_stats = p.stats != nullptr;
  const int nb = n0 + wn * WTN;
  const unsigned lch_own = (unsigned)(nb + ge * 4), lch_pair = (unsigned)(nb + 8 * (ge >> 1) + 16 * (ge & 1));
  f32x4_t bb[FN];
#pragma unroll
  for (int i = 0; i < FN; ++i) bb[i] = p.bias ? *(const f32x4_t*)(p.bias + nb + i * 16 + ge * 4) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
  // (two copies of the tile loop, with and without the BatchNorm sums: their 2 x FN x 4 accumulators next to the gate
  //  batch spilled the three-workgroups-per-CU instance)
  auto tile_out = [&](auto stats_c) {
  constexpr bool STATS = decltype(stats_c)::value;
  float s1[STATS ? FN : 1][4], s2[STATS ? FN : 1][4];
#pragma unroll
  for (int i = 0; i < (STATS ? FN : 1); ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[i][r] = 0.f; s2[i][r] = 0.f; }
#pragma unroll
  for (int j = 0; j < FM; ++j) {
    const int m = m0 + wm * WTM + j * 16 + r16e;
    const bool mv = m < p.M;
    unsigned fpix = (unsigned)m, opix = (unsigned)m;
    bool inside = true;
    if (!p.dense_out) {
      int b, oy, ox;
      if (p.howo_shift >= 0) { b = m >> p.howo_shift; const int r = m & (HoWo - 1); oy = r >> p.wo_shift; ox = r & (p.Wo - 1); }
      else { b = m / HoWo; const int r = m - b * HoWo; oy = r / p.Wo; ox = r - oy * p.Wo; }
      const int ty_ = oy * p.osy + ooy, tx_ = ox * p.osx + oox;
      fpix = (unsigned)((b * p.Hout_t + ty_) * p.Wout_t + tx_); opix = fpix;
      if (p.out2) {
        const int cy = ty_ - p.win_y0, cx = tx_ - p.win_x0;
        inside = (unsigned)cy < (unsigned)p.win_h && (unsigned)cx < (unsigned)p.win_w;
        opix = (unsigned)((b * p.win_h + cy) * p.win_w + cx);
      }
    }
    const unsigned offo = (mv && inside) ? opix * (unsigned)(p.ops * es) : OOB;
    const unsigned offh = (mv && !inside) ? fpix * (unsigned)(p.o2ps * es) : OOB;
    f32x4_t gt[FN];
    if (!STATS && has_gate) {
      const unsigned offg = (mv && inside) ? opix * (unsigned)(p.gps * ges) : OOB;
#pragma unroll
      for (int i = 0; i < FN; ++i) {
        const unsigned go = offg + (lch_own + i * 16) * ges;
        if (p.gdt == CSMRI_F32) gt[i] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rs_gate, (int)go, 0, 0));
        else {
          const u32x2_t u = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(rs_gate, (int)go, 0, 0));
          gt[i] = (f32x4_t){__uint_as_float(u[0] << 16), __uint_as_float(u[0] & 0xffff0000u),
                            __uint_as_float(u[1] << 16), __uint_as_float(u[1] & 0xffff0000u)};
        }
      }
    }
    f32x4_t vv[FN];
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      f32x4_t v = acc[i][j] + bb[i];
      if constexpr (STATS) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float q = mv ? v[r] : 0.f; s1[i][r] += q; s2[i][r] += q * q; }
      }
      if (has_act) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], v[r] * p.slope);            // 0 <= slope <= 1 (gconv_glds_eligible)
      }
      if (!STATS && has_gate) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (gt[i][r] > 0.f || !inside) ? v[r] : v[r] * p.gslope;   // halo leaves ungated
      }
      vv[i] = v;
    }
    if (bf_out) {
#pragma unroll
      for (int i = 0; i < FN; i += 2) {
        const u32x2_t a = pack4_bf16(vv[i]), c = pack4_bf16(vv[i + 1]);
        const auto x0_ = __builtin_amdgcn_permlane16_swap(a[0], c[0], false, false);
        const auto x1_ = __builtin_amdgcn_permlane16_swap(a[1], c[1], false, false);
        const u32x4_t d = (u32x4_t){x0_[0], x1_[0], x0_[1], x1_[1]};
        const unsigned ch = (lch_pair + i * 16) * 2u;
        __builtin_amdgcn_raw_buffer_store_b128(d, rs_out, (int)(offo == OOB ? OOB : offo + ch), 0, 0);
        if (p.out2) __builtin_amdgcn_raw_buffer_store_b128(d, rs_halo, (int)(offh == OOB ? OOB : offh + ch), 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < FN; ++i) {
        const unsigned ch = (lch_own + i * 16) * 4u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, vv[i]), rs_out, (int)(offo == OOB ? OOB : offo + ch), 0, 0);
        if (p.out2) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, vv[i]), rs_halo, (int)(offh == OOB ? OOB : offh + ch), 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);                 // (one fragment row at a time: hoisted, the four rows' offsets and gates spill)
  }
  if constexpr (STATS) {
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a = s1[i][r], b = s2[i][r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
        const int n = nb + i * 16 + ge * 4 + r;
        if (r16e == 0) {
          // partial sums are channel-major: [2][Cout][rows], rows = mtiles * WM (finalize reads coalesced)
          const size_t R = (size_t)p.mtiles * WM, r = (size_t)mt * WM + wm;
          p.stats[(size_t)n * R + r] = a; p.stats[((size_t)p.Cout + n) * R + r] = b;
        }
      }
  }
  };
  if (has_stats) tile_out(std::true_type{}); else tile_out(std::false_type{});
}

// ---------------------------------------------------------------------------------------------
int gconv_glds_bn(const csmri_gconv_desc* d) { return d->Cout % 128 == 0 ? 128 : 64; }

int gconv_glds_eligible(const csmri_gconv_desc* d) {
  if (d->dtype != CSMRI_BF16) return 0;
  if (d->Cin % 64 || d->Cout % 64) return 0;
  if (d->in1 && d->c0 % 64) return 0;
  if (d->in0_pix_stride % 8 || (d->in1 && d->in1_pix_stride % 8)) return 0;
  if (d->TH * d->TW > 16) return 0;                     // rows of the source-pixel table
  if (d->stats_partial && d->g_src) return 0;           // the BatchNorm-sums instance of the epilogue carries no gate
  // epilogue: max(v, slope v); 32-bit byte offsets into the output / halo / gate tensors (buffer descriptors)
  if (!(d->act_slope >= 0.f && d->act_slope <= 1.f)) return 0;
  const long long out_px = (long long)d->B * d->Hout_t * d->Wout_t;
  const long long widest = std::max((long long)d->out_pix_stride, std::max((long long)(d->out_halo ? d->halo_pix_stride : 0),
                                                                           (long long)(d->g_src ? d->g_pix_stride : 0)));
  if (out_px * widest * 4 >= (1ll << 31)) return 0;
  return 1;
}

template <int BN, int NST, int NW = 4>
static int launch_glds(const GParams& p, hipStream_t st) {
  constexpr int lds = (128 + BN) * 128 * NST + 16 * 128 * 4;     // staging buffers + the (tap, row) source-pixel table
  dim3 grid(p.mtiles * p.ntiles, 1, p.nclass * p.splitk);
  auto kern = gconv_glds_kernel<BN, NST, NW>;
  CSMRI_SET_MAX_LDS(kern, lds);
  hipLaunchKernelGGL(kern, grid, dim3(64 * NW), lds, st, p);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

// LDS stages by grid size: one buffer when 1.25+ workgroups per CU overlap each other, two below that, a 4-deep ring for grids of at most half a workgroup per CU (re-measured after
// the two-buffer loop was tuned: it now beats the ring on 129..256-workgroup grids, e.g. 36 vs 41 us on
// the U-Net 128 -> 128 4x4 layer)
#ifndef GLDS_ST4_MAX
#define GLDS_ST4_MAX 128
#endif
// (two buffers up to 320 workgroups: same-box A/B of the bench step, slices/s resident, two boxes -- 256: -0.4 %,
// 320: +0.4 / +0.5 %, 384: +0.4 %, 512 (rounds 2-4): reference, 1024: -0.7 %; ring of four up to 64 or 256: -0.3 %)
#ifndef GLDS_ST2_MAX
#define GLDS_ST2_MAX 320
#endif
static int glds_stages(long long blocks) {
  return blocks <= GLDS_ST4_MAX ? 4 : (blocks <= GLDS_ST2_MAX ? 2 : 1);
}

int gconv_glds_launch(const GParams& p0, const csmri_gconv_desc* d, hipStream_t st) {
  GParams p = p0;
  const int bn = gconv_glds_bn(d);
  p.nsteps = d->TH * d->TW * d->Cin / 64;
  p.steps_per_split = cdiv(p.nsteps, p.splitk);
  p.mtiles = cdiv(p.M, 128); p.ntiles = d->Cout / bn;
  const long long w_elems = (long long)d->Cout * d->TH * d->TW * d->Cin * p.nclass;
  const long long x_elems = (long long)d->B * d->Hin * d->Win * d->Cin;
  p.nt_major = w_elems > x_elems;
  const int nst = glds_stages((long long)p.mtiles * p.ntiles * p.nclass * p.splitk);   // (kernel_name mirrors this)
  // 8 waves (2 x 4) instead of 4 (2 x 2) on the same tile: every wave issues half the LDS-DMA instructions.  A wave's
  // DMA stream is what limits the operand intake of these grid-limited layers (1-2 workgroups per CU): 481 -> 618
  // TFLOP/s on the U-Net 128 -> 128 4x4 layer, +4..12 % on the discriminator / VGG conv3-4 shapes, -2 % on 16 x 16
  // maps; 16 waves (2 x 8) lose 5-12 % again: one fragment column per wave doubles the LDS reads per MFMA
  // (profiles/r02_gconv_glds_8wave_sweep.log): the 8-wave layout is used for the two-buffer 128-channel variant only.
  if (bn == 128 && nst == 2) return launch_glds<128, 2, 8>(p, st);
  if (bn == 128) return nst == 4 ? launch_glds<128, 4>(p, st) : launch_glds<128, 1>(p, st);
  return nst == 4 ? launch_glds<64, 4>(p, st) : nst == 2 ? launch_glds<64, 2>(p, st) : launch_glds<64, 1>(p, st);
}

void gconv_glds_kernel_name(const csmri_gconv_desc* d, char* buf, int n) {
  const int bn = gconv_glds_bn(d), nclass = d->nclass > 0 ? d->nclass : 1, sk = d->splitk > 0 ? d->splitk : 1;
  const long long blocks = (long long)cdiv((long long)d->B * d->Ho * d->Wo, 128) * (d->Cout / bn) * nclass * sk;
  const int nst = glds_stages(blocks);
  const bool w8 = bn == 128 && nst == 2;
  snprintf(buf, n, "gconv_glds_kernel<%d, %d, %d>", bn, nst, w8 ? 8 : 4);
}
