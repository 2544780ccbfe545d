// gpipe: the implicit-GEMM gather convolution of gconv_glds.hip rebuilt as a PERSISTENT, ROLE-SPLIT, PIPELINED kernel
// (round 6) -- the discriminator's 4 x 4 stride-2 / stride-1 layers 2..6, forward and data gradient (reference
// models/discriminators.py:137-172), and whatever else gconv_glds used to take without a channel concat / upsampling.
//
//   D[n][m] = sum_k W[n][k] * X[m][k],  k = (tap, ci),  bf16 operands, fp32 accumulate; X gathered on the fly
//
// Why a new loop.  gconv_glds is {issue the step's LDS-DMA -> vmcnt(0) -> barrier -> multiply -> barrier} per 64-deep K
// step, three short-lived workgroups per CU hiding each other's stalls, 128 x 128 tiles.  On these layers (K loops of 8-64
// steps per workgroup, grids filled by split-K) a workgroup spends most of its life in its prologue (source-pixel table,
// first DMA round trip), its epilogue and the drain between them, and the 128 x 128 tile moves 32 KiB from L2 into LDS per
// 2.1 MFLOP (64 FLOP/B), which the ~70 GB/s per CU an LDS gather from L2 sustains (MI355X_MICROARCH.md "Indexed rows")
// caps near 1.1 PFLOP/s: the measured 0.23-0.6 PFLOP/s are ~half of that bound (profiles/r06_dl_shapes.log).
//
// Structure (pconv2.hip's skeleton with BOTH operands streamed):
//   * one workgroup of 12 waves per CU, persistent: it walks work items (m-tile, n-tile, class, K-slice) id, id + grid, ...
//     and the K steps of consecutive items form ONE flat sequence -- the operand stream never drains at an item boundary,
//     the first two stages of the next item are in flight while the compute waves store the current item's tile;
//   * tile 64 FM positions x BN channels (FM = 4 or 3: 256 / 192 rows, BN = 128 / 64) x 64 K per step: 85 / 77 FLOP/B;
//   * waves 8..11 LOADERS: own their tile rows' source pixels in registers (no LDS table, no barrier for it), recompute the
//     gather addresses at tap changes, issue the stage two steps ahead into a ring of three (counted vmcnt retires exactly
//     the stage read next); waves 0..7 COMPUTE: 4 position groups x 2 channel halves, the halves half a step apart
//     (ping-pong: on every SIMD one wave multiplies while the other reads its fragments);
//   * two workgroup barriers per step; a stage is read one phase after the barrier behind the wait that retired it and
//     refilled one phase after the barrier behind its last read;
//   * LDS images = gconv_glds's 128-byte rows, 16-byte slots XOR-swizzled on the DMA SOURCE side;
//   * epilogue = gconv_glds's straight-line form (buffer stores with out-of-range offsets, batched gate loads, 16-byte
//     stores of exchanged fragment pairs, BatchNorm partial sums per 64 positions, split-K slabs for csmri_gconv_reduce).
#include <algorithm>
#include <type_traits>
#include <utility>
#include "mma_core.h"
#include "gconv_params.h"

__device__ __attribute__((aligned(16))) char gp_zero_page[16];
typedef __attribute__((address_space(1))) const void* gpg_t;
typedef __attribute__((address_space(3))) void* gpl_t;

template <int N> __device__ __forceinline__ void gp_vmwait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
#define GP_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

#ifdef CSMRI_DBG_STAMPS
#define GP_STAMP(i) do { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
    ph[i] += t_ - last_t; last_t = t_; } while (0)
#define GP_STAMP_DECL unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_t; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last_t) :: "memory")
#define GP_STAMP_DUMP do { if (lane == 0 && p.slab && p.splitk == 1) { unsigned long long* dbg_ = (unsigned long long*)p.slab + ((size_t)blockIdx.x * 12 + wv) * 8; \
    for (int i_ = 0; i_ < 8; ++i_) dbg_[i_] = ph[i_]; } } while (0)
#else
#define GP_STAMP(i) do {} while (0)
#define GP_STAMP_DECL do {} while (0)
#define GP_STAMP_DUMP do {} while (0)
#endif

#define GP_TIE_A3 "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2])
#define GP_TIE_A4 "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]), "+v"(a[1][3])
#define GP_TIE_B2 "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1])
#define GP_TIE_B4 "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[0][2]), "+v"(b[0][3]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[1][2]), "+v"(b[1][3])

// n / d for a divisor fixed on the host (round-up magic number, branch-free form): q = (t + ((n - t) >> 1)) >> s, t = mulhi(m, n)
__device__ __forceinline__ unsigned gp_div(unsigned n, unsigned m, unsigned sh) { const unsigned t = __umulhi(m, n); return (t + ((n - t) >> 1)) >> sh; }

struct GpItem { int m0, n0, cls, ks, mt, s_begin, nst; };

template <int FM, int BN>
__global__ __launch_bounds__(768, 1) void gpipe_kernel(const GParams p) {
  constexpr int BM = 64 * FM, ABYTES = BM * 128, BBYTES = BN * 128, STG = ABYTES + BBYTES, NS = 3;
  constexpr int KA = BM / 32, KB = BN / 32, NP = KA + KB;          // LDS-DMA pieces (1 KiB) per loader wave and stage
  constexpr int FN = BN / 32;                                       // 16-channel fragments per compute wave
  static_assert((FM == 4 || FM == 3) && (BN == 128 || BN == 64) && NS * STG + 16 * BM * 4 <= 160 * 1024, "tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles = p.mtiles * p.ntiles, total = tiles * p.nclass * p.splitk, G = gridDim.x;
  const int n_my = blockIdx.x < total ? (total - (int)blockIdx.x + G - 1) / G : 0;
  if (n_my == 0) return;
  const int HoWo = p.Ho * p.Wo;

  // work item j of this workgroup.  Ids are dealt round-robin over the 8 XCDs: xcd_remap gives each XCD a contiguous run
  // of tiles, ordered so that the operand that is re-read across the run is the SMALL one (nt_major, as gconv_glds)
  auto item = [&](int j) __attribute__((always_inline)) {
    GpItem it;
    const int t = xcd_remap((int)blockIdx.x + j * G, total);
    const int kc = t / tiles, tt = t - kc * tiles;
    it.mt = p.nt_major ? tt % p.mtiles : tt / p.ntiles;
    const int nt = p.nt_major ? tt / p.mtiles : tt - it.mt * p.ntiles;
    it.cls = kc % p.nclass; it.ks = kc / p.nclass;
    it.m0 = it.mt * BM; it.n0 = nt * BN;
    it.s_begin = it.ks * p.steps_per_split;
    it.nst = min(p.nsteps, it.s_begin + p.steps_per_split) - it.s_begin;      // >= 1 (host)
    return it;
  };
  int S = 0;
  for (int j = 0; j < n_my; ++j) S += item(j).nst;
  // m -> (image, row, column) of the output class: shifts when Ho * Wo and Wo are powers of two, else host-made magic numbers
  auto decode = [&](int m, int& b, int& oy, int& ox) __attribute__((always_inline)) {
    if (p.howo_shift >= 0) { b = m >> p.howo_shift; const int r = m & (HoWo - 1); oy = r >> p.wo_shift; ox = r & (p.Wo - 1); }
    else { b = (int)gp_div((unsigned)m, p.dv_howo_m, p.dv_howo_s); const int r = m - b * HoWo; oy = (int)gp_div((unsigned)r, p.dv_wo_m, p.dv_wo_s); ox = r - oy * p.Wo; }
  };

  if (wv >= 8) {
    // =================================================== loader waves ===================================================
    const int L = wv - 8, lrow = lane >> 3;
    // piece k of a stage = rows 8 (L + 4 k) .. + 7 of the image; this lane's 16-byte slot holds K chunk `chunk` of its row
    const int chunk = (lane & 7) ^ ((4 * (L & 1) + (lane >> 4)) & 7);
    const char* aptr[KA]; unsigned ainc[KA];
    const char* wptr[KB];
    int ci = 0, tap = 0, left = 0, jn = 0;
    const int ntaps = p.TH * p.TW, sub = lane & 7;
    const char* in0 = p.in0 + chunk * 16;
    const char* zero_page = gp_zero_page;
    const unsigned ps2 = (unsigned)p.ps0 * 2u;
    // Source pixel of every (tap, tile row) of the current item: a table in LDS behind the ring, [16 taps][BM rows] ints,
    // -1 = outside the image (zero page).  The eight lanes that share a row build it together at the item's start (two
    // taps each) and read it back at every tap change -- one ds_read per row instead of the border arithmetic (the
    // arithmetic at every tap change cost the loaders ~1700 cycles per change, profiles/r06_gpipe_stamps2.log).  Only this
    // wave touches its rows' entries: no barrier, LDS operations of one wave stay in order.
    int* const tab = (int*)(smem + NS * STG) + 8 * L + lrow;          // + 32 k for piece k, + BM per tap
    const int tinv = (256 + p.TW - 1) / p.TW;                          // tap / TW for tap < 16, TW <= 4
    auto load_tap = [&]() __attribute__((always_inline)) {
      int pix[KA];
#pragma unroll
      for (int k = 0; k < KA; ++k) pix[k] = tab[tap * BM + 32 * k];
#pragma unroll
      for (int k = 0; k < KA; ++k) {
        const bool ok = pix[k] >= 0;
        const char* g = in0 + ((unsigned)pix[k] * ps2 + (unsigned)ci * 2u);         // 32-bit offsets (gpipe_eligible)
        aptr[k] = ok ? g : zero_page;
        ainc[k] = ok ? 128u : 0u;
      }
    };
    auto begin_item = [&]() __attribute__((always_inline)) {
      const GpItem it = item(jn++);
      left = it.nst;
#pragma unroll
      for (int k = 0; k < KA; ++k) {
        const int m = it.m0 + 8 * (L + 4 * k) + lrow;
        int b = 0, oy = 0, ox = 0;
        const bool mv = m < p.M;
        if (mv) decode(m, b, oy, ox);
        const int by = oy * p.S + p.dy0, bx = ox * p.S + p.dx0, ibase = b * p.Hin * p.Win;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const int t = sub + 8 * tt;
          if (t < ntaps) {
            const int ty = (t * tinv) >> 8, tx = t - ty * p.TW;
            int u = by + ty * p.dys, v = bx + tx * p.dxs;
            bool ok = mv;
            if (p.border == CSMRI_BORDER_REFLECT) { u = reflect_idx(u, p.Hin); v = reflect_idx(v, p.Win); }
            else ok = ok && (unsigned)u < (unsigned)p.Hin && (unsigned)v < (unsigned)p.Win;
            tab[t * BM + 32 * k] = ok ? ibase + u * p.Win + v : -1;
          }
        }
      }
      const int k0 = it.s_begin * 64;
      tap = k0 / p.Cin; ci = k0 - tap * p.Cin;
      load_tap();
#pragma unroll
      for (int k = 0; k < KB; ++k)
        wptr[k] = p.w + ((size_t)it.cls * (size_t)p.wcs + (size_t)(it.n0 + 8 * (L + 4 * k) + lrow) * p.Kp + chunk * 8) * 2 +
                  (size_t)it.s_begin * 128;
    };
    // A stage is issued in TWO parts, one per half step (an LDS-DMA instruction costs its wave ~100-160 cycles: the
    // whole stage in the first half made that half 1600 cycles long with the compute waves waiting at its barrier,
    // profiles/r06_gpipe_stamps1.log): pieces [0, N1) = the first N1 of {A pieces, then B pieces}, pieces [N1, NP) the rest.
#ifndef GPIPE_N1_NUM
#define GPIPE_N1_NUM 8        // swept 5 / 6 / 7 / 8 tenths on ten launches: 448 / 441 / 441 / 437 us in all (profiles/r06_gpipe_n1.log)
#endif
    constexpr int N1 = (NP * GPIPE_N1_NUM + 9) / 10;                  // first part: 80 % of the pieces (the second half also holds the wait and the tap / item bookkeeping)
    unsigned ring = 0;
    auto issue_part = [&](auto part_c) __attribute__((always_inline)) {
      constexpr int part = decltype(part_c)::value;
      char* dst = smem + ring + L * 1024;
#pragma unroll
      for (int k = 0; k < KA; ++k)
        if ((part == 0) == (k < N1)) {
          __builtin_amdgcn_global_load_lds((gpg_t)aptr[k], (gpl_t)(dst + k * 4096), 16, 0, 0);
          aptr[k] += ainc[k];
        }
#pragma unroll
      for (int k = 0; k < KB; ++k)
        if ((part == 0) == (KA + k < N1)) {
          __builtin_amdgcn_global_load_lds((gpg_t)wptr[k], (gpl_t)(dst + ABYTES + k * 4096), 16, 0, 0);
          wptr[k] += 128;
        }
      if constexpr (part == 1) {
        ring += STG; if (ring == NS * STG) ring = 0;
        if (--left == 0) { if (jn < n_my) begin_item(); }
        else {
          ci += 64;
          if (ci == p.Cin) { ci = 0; ++tap; load_tap(); }
        }
      }
    };
    begin_item();
    issue_part(std::integral_constant<int, 0>{}); issue_part(std::integral_constant<int, 1>{});
    if (S > 1) { issue_part(std::integral_constant<int, 0>{}); issue_part(std::integral_constant<int, 1>{}); gp_vmwait<NP>(); }
    else gp_vmwait<0>();
    __builtin_amdgcn_s_barrier();                      // stage 0 has landed
    GP_STAMP_DECL;
    for (int s = 0; s < S; ++s) {
      // ---- first half of step s: the first part of stage s+2 into the slot stage s-1 left
      if (s + 2 < S) issue_part(std::integral_constant<int, 0>{});
      GP_STAMP(0);
      __builtin_amdgcn_s_barrier();
      GP_STAMP(1);
      // ---- second half: retire stage s+1 (read from the next first half on) -- only the N1 pieces just issued are
      // younger --, then the second part of stage s+2
      if (s + 2 < S) { gp_vmwait<N1>(); GP_STAMP(2); issue_part(std::integral_constant<int, 1>{}); }
      else { gp_vmwait<0>(); GP_STAMP(2); }
      GP_STAMP(3);
      __builtin_amdgcn_s_barrier();
      GP_STAMP(4);
    }
    __builtin_amdgcn_s_barrier();
    GP_STAMP_DUMP;
    return;
  }

  // ===================================================== compute waves =====================================================
  const int wm = wv & 3, wn = wv >> 2;                // wn is also the ping-pong group (one wave of each per SIMD)
  const int r16 = lane & 15, g = lane >> 4;
  const unsigned lds0 = (unsigned)(size_t)(gpl_t)smem;
  f32x4_t acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int f = 0; f < FM; ++f) acc[i][f] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const unsigned swz = (unsigned)((r16 >> 1) & 7);
  const unsigned ab0 = lds0 + (wm * 16 * FM + r16) * 128 + ((g ^ swz) << 4), ab1 = lds0 + (wm * 16 * FM + r16) * 128 + (((4 + g) ^ swz) << 4);
  const unsigned bb0 = lds0 + ABYTES + (wn * (BN / 2) + r16) * 128 + ((g ^ swz) << 4), bb1 = lds0 + ABYTES + (wn * (BN / 2) + r16) * 128 + (((4 + g) ^ swz) << 4);

  // ---- epilogue of one item (gconv_glds's straight-line form) ----
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
  auto epilogue = [&](const GpItem& it) __attribute__((always_inline)) {
    // per-lane constants of the epilogue are rebuilt here from an opaque copy of the lane id: hoisted out of the step loop
    // they would stay live across it (the loop runs at the register cap: they were spilled and re-loaded from scratch)
    int le = lane;
    asm volatile("" : "+v"(le));
    const int r16 = le & 15, g = le >> 4;
    const int nb = it.n0 + wn * (BN / 2), mb = it.m0 + wm * 16 * FM;
    if (p.splitk > 1) {
      // a split item stores slab rows indexed by m: no output position needed
#pragma unroll
      for (int f = 0; f < FM; ++f) {
        const int m = mb + f * 16 + r16;
        if (m < p.M) {
          float* row = p.slab + (((size_t)it.cls * p.splitk + it.ks) * p.M + m) * p.Cout + nb + g * 4;
#pragma unroll
          for (int i = 0; i < FN; ++i) *(f32x4_t*)(row + i * 16) = acc[i][f];
        }
#pragma unroll
        for (int i = 0; i < FN; ++i) acc[i][f] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      }
      return;
    }
    const int ooy = p.ooy + (p.nclass == 4 ? (it.cls >> 1) : 0), oox = p.oox + (p.nclass == 4 ? (it.cls & 1) : 0);
    const unsigned lch_own = (unsigned)(nb + g * 4), lch_pair = (unsigned)(nb + 8 * (g >> 1) + 16 * (g & 1));
    f32x4_t bb[FN];
#pragma unroll
    for (int i = 0; i < FN; ++i) bb[i] = p.bias ? *(const f32x4_t*)(p.bias + nb + i * 16 + g * 4) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
    auto tile_out = [&](auto stats_c) __attribute__((always_inline)) {
      constexpr bool STATS = decltype(stats_c)::value;
      float s1[STATS ? FN : 1][4], s2[STATS ? FN : 1][4];
#pragma unroll
      for (int i = 0; i < (STATS ? FN : 1); ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[i][r] = 0.f; s2[i][r] = 0.f; }
#pragma unroll
      for (int f = 0; f < FM; ++f) {
        const int m = mb + f * 16 + r16;
        const bool mv = m < p.M;
        unsigned fpix = (unsigned)m, opix = (unsigned)m;
        bool inside = true;
        if (!p.dense_out) {
          int b, oy, ox;
          decode(mv ? m : 0, b, oy, ox);
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
          f32x4_t v = acc[i][f] + bb[i];
          acc[i][f] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
          if constexpr (STATS) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float q = mv ? v[r] : 0.f; s1[i][r] += q; s2[i][r] += q * q; }
          }
          if (has_act) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], v[r] * p.slope);            // 0 <= slope <= 1 (gpipe_eligible)
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
        __builtin_amdgcn_sched_barrier(0);             // (one fragment row at a time: hoisted, the rows' offsets and gates spill)
      }
      if constexpr (STATS) {
        // one partial row per wave = 64 consecutive positions (FM == 4: the host gives the statistics form 256-row tiles)
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float a = s1[i][r], b = s2[i][r];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
            const int n = nb + i * 16 + g * 4 + r;
            if (r16 == 0) {
              const size_t R = (size_t)p.mtiles * 4, rr = (size_t)it.mt * 4 + wm;     // [2][Cout][rows]
              p.stats[(size_t)n * R + rr] = a; p.stats[((size_t)p.Cout + n) * R + rr] = b;
            }
          }
      }
    };
    if (has_stats) tile_out(std::true_type{}); else tile_out(std::false_type{});
  };

  __builtin_amdgcn_s_barrier();                       // stage 0 has landed
  if (wn) __builtin_amdgcn_s_barrier();               // channel half 1: half a step behind
  int jc = 0;
  GpItem cur = item(jc++);
  int left = cur.nst;
  unsigned ring = 0;
  GP_STAMP_DECL;
  for (int s = 0; s < S; ++s) {
    // ======== read half: this step's fragments ========
    const unsigned a0 = ab0 + ring, a1 = ab1 + ring, b0 = bb0 + ring, b1 = bb1 + ring;
    ring += STG; if (ring == NS * STG) ring = 0;
    u32x4_t a[2][FM], b[2][FN];
#pragma unroll
    for (int i = 0; i < FN; ++i) GP_READ(b[0][i], b0, i * 2048);
#pragma unroll
    for (int f = 0; f < FM; ++f) GP_READ(a[0][f], a0, f * 2048);
#pragma unroll
    for (int i = 0; i < FN; ++i) GP_READ(b[1][i], b1, i * 2048);
#pragma unroll
    for (int f = 0; f < FM; ++f) GP_READ(a[1][f], a1, f * 2048);
    // (the reads are complete before the barrier behind which a loader may refill what they read; every fragment is a tied
    //  operand of the wait, so neither an MFMA nor a register copy of a fragment can be placed in front of it)
    if constexpr (FM == 4 && FN == 4) asm volatile("s_waitcnt lgkmcnt(0)" : GP_TIE_A4, GP_TIE_B4);
    else if constexpr (FM == 3 && FN == 4) asm volatile("s_waitcnt lgkmcnt(0)" : GP_TIE_A3, GP_TIE_B4);
    else if constexpr (FM == 4 && FN == 2) asm volatile("s_waitcnt lgkmcnt(0)" : GP_TIE_A4, GP_TIE_B2);
    else asm volatile("s_waitcnt lgkmcnt(0)" : GP_TIE_A3, GP_TIE_B2);
    __builtin_amdgcn_sched_barrier(0);
    GP_STAMP(0);
    __builtin_amdgcn_s_barrier();
    GP_STAMP(1);
    // ======== multiply half ========
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int f = 0; f < FM; ++f)
          acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, b[kc][i]),
                                                              __builtin_bit_cast(bf16x8_t, a[kc][f]), acc[i][f], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    GP_STAMP(2);
    if (--left == 0) {
      epilogue(cur);
      if (jc < n_my) { cur = item(jc++); left = cur.nst; }
    }
    GP_STAMP(3);
    __builtin_amdgcn_s_barrier();
    GP_STAMP(4);
  }
  if (!wn) __builtin_amdgcn_s_barrier();
  GP_STAMP_DUMP;
}

// ---------------------------------------------------------------------------------------------
#ifndef GPIPE
#define GPIPE 1        // 1: eligible launches with at least GPIPE_MIN_STEPS K steps; 0: only where csmri_gconv_desc.flags carries CSMRI_GCONV_USE_GPIPE
#endif
// Measured against gconv_glds on the discriminator's fifteen launches (profiles/r06_gpipe_vs_glds.log): gpipe wins where an
// item's K loop is long enough to amortise its epilogue (the persistent workgroup cannot hide it behind another
// workgroup's K loop as three resident gconv_glds workgroups do): from 32 steps on (layers 3-6 forward, 4-6 data gradient).
#ifndef GPIPE_MIN_STEPS
#define GPIPE_MIN_STEPS 32
#endif
#ifndef GPIPE_CUS
#define GPIPE_CUS 256
#endif

struct GpPlan { int fm, bn, sk, grid, mtiles, ntiles; };

// Tile height and K split: the plan with the smallest modelled time.  A work item costs its K steps at the stage intake
// rate (L2 -> LDS gather, ~70 GB/s per CU) or the MFMA rate, whichever is slower, plus a fixed epilogue; items are dealt
// to GPIPE_CUS persistent workgroups in rounds; a split pays its slabs (written, read back by csmri_gconv_reduce).
static GpPlan gp_plan(const csmri_gconv_desc* d, int forced_sk = 0) {
  GpPlan best{}; double best_t = 1e30;
  const int bn = d->Cout % 128 == 0 ? 128 : 64, nclass = d->nclass > 0 ? d->nclass : 1;
  const long long M = (long long)d->B * d->Ho * d->Wo;
  const int nsteps = d->TH * d->TW * d->Cin / 64, ntiles = d->Cout / bn;
  const bool stats = d->stats_partial != nullptr;
  const int fm_hi = (d->flags & 8) ? 3 : 4, fm_lo = (d->flags & 4) ? 4 : (stats ? 4 : 3);      // (tools: flags 4 / 8 force the tile height)
  for (int fm = fm_hi; fm >= fm_lo; --fm) {
    const int bm = 64 * fm, mtiles = cdiv(M, bm);
    const long long base = (long long)mtiles * ntiles * nclass;
    const double t_intake = (bm + bn) * 128.0 / 70e3, t_mfma = 2.0 * fm * (bn / 32) * 2 * 16 / 2.2e3;   // us per step
    const double t_step = std::max(t_intake, t_mfma) + 0.08;
    const int max_sk = stats ? 1 : std::max(1, std::min(32, nsteps / 8));
    for (int sk = forced_sk > 0 ? forced_sk : 1; sk <= (forced_sk > 0 ? forced_sk : max_sk); ++sk) {
      const int sps = cdiv(nsteps, sk);
      if ((long long)(sk - 1) * sps >= nsteps) continue;             // every slice owns at least one step
      const long long items = base * sk;
      const long long rounds = (items + GPIPE_CUS - 1) / GPIPE_CUS;
      double t = rounds * (sps * t_step + 1.5) + 2.0;
      if (sk > 1) t += 3.0 + (double)M * d->Cout * nclass * sk * 4.0 * 2.0 / 4.5e6;
      if (t < best_t) { best_t = t; best = GpPlan{fm, bn, sk, (int)std::min<long long>(items, GPIPE_CUS), mtiles, ntiles}; }
    }
  }
  return best;
}

int gpipe_eligible(const csmri_gconv_desc* d) {
  if (d->flags & CSMRI_GCONV_NO_GPIPE) return 0;
  if (!(d->flags & CSMRI_GCONV_USE_GPIPE) && (!GPIPE || d->TH * d->TW * d->Cin / 64 < GPIPE_MIN_STEPS || d->stats_partial)) return 0;
  if (d->dtype != CSMRI_BF16) return 0;
  if (d->Cin % 64 || d->Cout % 64) return 0;
  if (d->in1 || d->upsample) return 0;
  if (d->in0_pix_stride % 8) return 0;
  if (d->TH * d->TW > 16 || d->TW > 4) return 0;        // rows of the source-pixel table; tap / TW by multiplication
  { const long long howo = (long long)d->Ho * d->Wo;     // m -> (b, oy, ox): both divisors powers of two, or neither (magic numbers)
    auto p2 = [](long long v) { return (v & (v - 1)) == 0; };
    if (!(p2(howo) && p2(d->Wo)) && (p2(howo) || p2(d->Wo) || d->Wo < 3)) return 0; }
  if ((long long)d->B * d->Hin * d->Win * d->in0_pix_stride * 2 >= (1ll << 31)) return 0;   // 32-bit input byte offsets
  if (d->stats_partial && d->g_src) return 0;           // the BatchNorm-sums instance of the epilogue carries no gate
  if (d->stats_partial && ((long long)d->B * d->Ho * d->Wo) % 64) return 0;
  if (!(d->act_slope >= 0.f && d->act_slope <= 1.f)) return 0;
  // 32-bit byte offsets into the output / halo / gate tensors (buffer descriptors) and 32-bit row indices
  const long long out_px = (long long)d->B * d->Hout_t * d->Wout_t;
  const long long widest = std::max((long long)d->out_pix_stride, std::max((long long)(d->out_halo ? d->halo_pix_stride : 0),
                                                                           (long long)(d->g_src ? d->g_pix_stride : 0)));
  if (out_px * widest * 4 >= (1ll << 31)) return 0;
  if ((long long)d->B * d->Ho * d->Wo + 256 >= (1ll << 31)) return 0;
  return 1;
}

int gpipe_splitk(const csmri_gconv_desc* d) { return gp_plan(d).sk; }
int gpipe_stats_rows(const csmri_gconv_desc* d) { return gp_plan(d, 1).mtiles * 4; }     // (asked with stats_partial set: fm = 4)

template <int FM, int BN>
static int launch_gpipe(const GParams& p, int grid, hipStream_t st) {
  constexpr int lds = 3 * (64 * FM + BN) * 128 + 16 * 64 * FM * 4;     // ring of three stages + the (tap, row) source-pixel table
  auto kern = gpipe_kernel<FM, BN>;
  CSMRI_SET_MAX_LDS(kern, lds);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(768), lds, st, p);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

int gpipe_launch(const GParams& p0, const csmri_gconv_desc* d, hipStream_t st) {
  GParams p = p0;
  // the caller's split decides (csmri_gconv_suggest_splitk = the free plan's; 1 with BatchNorm sums; the tools override it);
  // the tile height is the best one for that split
  const GpPlan pl = gp_plan(d, p.splitk);
  if (pl.fm == 0) return CSMRI_E_ARG;
  p.nsteps = d->TH * d->TW * d->Cin / 64;
  p.steps_per_split = cdiv(p.nsteps, pl.sk);
  if ((long long)(pl.sk - 1) * p.steps_per_split >= p.nsteps) return CSMRI_E_ARG;
  p.mtiles = pl.mtiles; p.ntiles = pl.ntiles;
  const long long items = (long long)pl.mtiles * pl.ntiles * p.nclass * pl.sk;
  const int grid = (int)std::min<long long>(items, GPIPE_CUS);
  const long long w_elems = (long long)d->Cout * d->TH * d->TW * d->Cin * p.nclass;
  const long long x_elems = (long long)d->B * d->Hin * d->Win * d->Cin;
  p.nt_major = w_elems > x_elems;
  auto magic = [](unsigned dd, unsigned* m, unsigned* sh) {            // round-up magic number of n / dd, dd >= 2 and not a power of two
    unsigned l = 0; while ((1ull << l) < dd) ++l;
    *m = (unsigned)((((1ull << l) - dd) << 32) / dd + 1); *sh = l - 1;
  };
  p.dv_howo_m = p.dv_howo_s = p.dv_wo_m = p.dv_wo_s = 0;
  if (p.howo_shift < 0) {
    auto pow2 = [](unsigned v) { return (v & (v - 1)) == 0; };
    // (a power-of-two divisor beside a non-power-of-two one: the magic form needs l >= 1; handle through shift-only magic)
    const unsigned howo = (unsigned)d->Ho * d->Wo, wo = (unsigned)d->Wo;
    if (pow2(howo) || pow2(wo) || howo < 2 || wo < 2) return CSMRI_E_UNSUPPORTED;
    magic(howo, &p.dv_howo_m, &p.dv_howo_s); magic(wo, &p.dv_wo_m, &p.dv_wo_s);
  }
  if (pl.fm == 4) return pl.bn == 128 ? launch_gpipe<4, 128>(p, grid, st) : launch_gpipe<4, 64>(p, grid, st);
  return pl.bn == 128 ? launch_gpipe<3, 128>(p, grid, st) : launch_gpipe<3, 64>(p, grid, st);
}

void gpipe_kernel_name(const csmri_gconv_desc* d, char* buf, int n) {
  const GpPlan pl = gp_plan(d, d->splitk > 0 ? d->splitk : 0);
  snprintf(buf, n, "gpipe_kernel<%d, %d>", pl.fm, pl.bn);
}
