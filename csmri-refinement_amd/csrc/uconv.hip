// uconv: the U-Net conv class (and VGG conv1_2) -- stride-1 4 x 4 / 3 x 3 layers with 32 or 64 input channels on large
// maps (reference models/unet.py:48,100,241; models/vgg.py:35) -- as a persistent, role-split kernel.  It replaces
// tconv.hip for these shapes: tconv is one-shot (patch DMA -> vmcnt(0) -> barrier -> K loop -> epilogue, one tile per
// workgroup); its stamps (profiles/r05_tconv_stamps_before.log) put 25-46 % of a workgroup's cycles into staging and
// 14-21 % into the epilogue, and its K loop reads 6-8 fragments from LDS per 8-16 MFMAs.
//
// Structure
//   * One workgroup of 8 waves per CU, persistent.  Waves 0..3 multiply, waves 4..7 only move data (an LDS-DMA piece
//     costs its issuing wave 60-185 cycles: MI355X_MICROARCH.md cycle constants).
//   * Work unit = a STRIP of 8 output rows x 16 output columns x NF*16 output channels, one per compute wave; a PASS =
//     4 strips.  Strips are independent (each has its own input patch in LDS), so ragged extents (the 259 x 259 /
//     131 x 131 outputs of the reflection-padded data gradients) cost 8-row, not 16- or 32-row, granularity.
//   * K order inside a pass: 32-channel chunk (outer), filter COLUMN tx, then the rows of the strip's patch.  For one
//     (chunk, tx) "iteration" a wave holds the TH x NF weight fragments of that filter column in registers and slides
//     down the patch: the fragment of patch row R (16 pixels x 32 channels, ONE ds_read_b128) feeds the MFMAs of output
//     rows R, R-1, .., R-TH+1 (filter rows 0..TH-1) x NF channel fragments.  LDS reads per MFMA: (ROWS+TH-1 + TH*NF) /
//     (ROWS*TH*NF) = 0.21 (4x4, NF=4), 0.30 (NF=2) against tconv's / pconv2's 0.5-0.75 -- the bound measured on pconv2.
//     The weight registers are refilled IN PLACE for the next iteration: filter row ty is dead after patch row
//     ROWS-1+ty and first needed again at patch row ty of the next iteration, 16-24 MFMAs later.
//   * Weights stream through a ring of 3 stages (one stage = one iteration's TH x NF*16 x 32 block, 8-16 KiB); the
//     patch of 32-channel chunk c of the next pass is loaded into the buffer chunk c of this pass has left (one chunk:
//     two pass buffers).  ONE workgroup barrier per iteration = per 64-128 MFMAs of a wave (pconv2: two per 32).
//   * Loaders wait with counted vmcnt; a buffer is read after the barrier behind the wait that retired it and refilled
//     after the barrier behind its last read (cdna_hip_programming.md, "Read a staged buffer one phase AFTER ...").
//   * Epilogue of output row r (bias, slope, activation-derivative gate, BatchNorm partial sums, bf16 store) is emitted
//     between the MFMAs of the pass's last iteration, as soon as the row's last filter row has been added; the
//     BatchNorm partial sums live in registers for the whole launch (one stats row per wave).
// LDS images: patch = tconv's plane-major image per strip ([8-channel plane][pixel][16 B], PLANE % 256 == 0: conflict-
// free ds_read_b128 at every tap shift); weights = mma_core.h's 64-byte rows, swizzled at the DMA source.
#include <utility>
#include "mma_core.h"
#include "gconv_params.h"

__device__ __attribute__((aligned(16))) char u_zero_page[16];
typedef __attribute__((address_space(1))) const void* ug_t;
typedef __attribute__((address_space(3))) void* ul_t;

#ifndef UCONV_ABLATE
#define UCONV_ABLATE 0     // diagnostic builds only (tools/run/r05_ablate.sh): 1 no epilogue, 2 no DMA, 4 no LDS reads in the loop
#endif
#define U_ROWS 8
#define U_PITCH 20         // pixels per patch row in LDS (16 + taps - 1 = 18 or 19 used)
#define U_STRIPS 4
#define U_RING 3
#define U_STATS 1
#define U_BIAS 2
#define U_GATE 4
#define U_WIN 8

template <int N> __device__ __forceinline__ void u_vmwait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
#define U_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

template <int... I, class F>
__device__ __forceinline__ void u_unroll(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }

// s_waitcnt lgkmcnt(N) that hands the registers it has retired to the compiler (the asm "modifies" them, so no consumer is
// scheduled above the wait: cdna_hip_programming.md 5.7 form (ii))
template <int N, int NF>
__device__ __forceinline__ void u_lgkm(u32x4_t& f, u32x4_t* w, bool) {
  if constexpr (NF == 4)
    asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(f), "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]) : "n"(N));
  else
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(f), "+v"(w[0]), "+v"(w[1]) : "n"(N));
}
template <int N>
__device__ __forceinline__ void u_lgkm(u32x4_t& f) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N)); }

#ifdef CSMRI_DBG_STAMPS
#define U_NOW(t) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define U_STAMP(i) do { unsigned long long t_; U_NOW(t_); stp[i] += t_ - last_t; last_t = t_; } while (0)
#define U_STAMP_DECL unsigned long long stp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_t; U_NOW(last_t)
#define U_STAMP_DUMP do { if (lane == 0 && p.slab) { unsigned long long* dbg_ = (unsigned long long*)p.slab + ((size_t)blockIdx.x * 8 + wv) * 8; \
    for (int i_ = 0; i_ < 8; ++i_) dbg_[i_] = stp[i_]; } } while (0)
#else
#define U_STAMP(i) do {} while (0)
#define U_STAMP_DECL do {} while (0)
#define U_STAMP_DUMP do {} while (0)
#endif

template <int TH, int TW, int NCH, int NF, int MODE>
__global__ __launch_bounds__(512, 2) void uconv_kernel(const GParams p) {
  constexpr int ROWS = U_ROWS, NR = ROWS + TH - 1, PITCH = U_PITCH, NPIX = NR * PITCH;
  constexpr int NPC = (NPIX + 15) / 16;             // LDS-DMA pieces (16 pixels x 64 B) per strip and 32-channel chunk
  constexpr int SBUF = NPC * 1024, PBUF = U_STRIPS * SBUF;
  constexpr int WST = TH * NF * 1024, WOFF = 2 * PBUF;
  constexpr int NIT = NCH * TW;                     // iterations (chunk, tx) per pass
  constexpr int WPI = TH * NF / 4;                  // weight pieces per loader and stage
  constexpr int PPS0 = (NPC + 1) / 2, PPS1 = NPC - PPS0;   // patch pieces per loader in a phase's slots 0 and 1
  constexpr bool STATS = (MODE & U_STATS) != 0, BIAS = (MODE & U_BIAS) != 0, GATE = (MODE & U_GATE) != 0, WIN = (MODE & U_WIN) != 0;
  static_assert((TH * NF) % 4 == 0 && TW >= 3 && TH >= 3 && NCH >= 1 && NCH <= 2, "shape");
  static_assert(16 + TW - 1 <= PITCH && 2 * PBUF + U_RING * WST <= 160 * 1024, "LDS");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nstrips = p.us_n, SX = p.us_x, SY = p.us_y, NB = p.ntiles;
  const int npass = (nstrips + U_STRIPS - 1) / U_STRIPS;
  // which output-channel block, which worker (pconv2's mapping: an XCD keeps ONE channel block's weights in its L2)
  int nblk, worker, workers;
  {
    const int id = blockIdx.x, G = gridDim.x;
    if ((8 % NB) == 0 && (G & 7) == 0) {
      const int per = 8 / NB, xcd = id & 7;
      nblk = xcd / per; worker = (id >> 3) * per + (xcd % per); workers = (G >> 3) * per;
    } else { nblk = id % NB; worker = id / NB; workers = G / NB; }
  }
  const int n0 = nblk * NF * 16;
  const int my_passes = worker < npass ? (npass - worker + workers - 1) / workers : 0;
  const int NTOT = my_passes * NIT;
  if (NTOT == 0) return;

  if (wv >= 4) {
    // =================================================== loader waves ===================================================
    const int L = wv - 4;                              // loader L moves strip L's patch and weight pieces j = L (mod 4)
    const int wrow = lane >> 2;
    const int kc = (lane & 3) ^ tile_swz(wrow);        // (tile_swz depends on (row >> 2) & 3 only: the same for row + 16 pn)
    const char* wsrc = p.w + ((size_t)(n0 + wrow) * p.Kp + kc * 8) * 2;
    int w_i = 0, w_tx = 0, w_c = 0;
    unsigned w_ring = 0;
    auto issue_w = [&]() {
      char* dst = smem + WOFF + w_ring + L * 1024;
      const unsigned koff = (unsigned)((w_tx * p.Cin + w_c * 32) * 2);
#pragma unroll
      for (int k = 0; k < WPI; ++k) {
        const int j = L + 4 * k, ty = j / NF, pn = j % NF;
        const char* src = wsrc + ((size_t)pn * 16 * p.Kp + (size_t)ty * TW * p.Cin) * 2 + koff;
        if (!(UCONV_ABLATE & 2)) __builtin_amdgcn_global_load_lds((ug_t)src, (ul_t)(dst + 4096 * k), 16, 0, 0);
      }
      ++w_i;
      w_ring = w_ring + WST == U_RING * WST ? 0u : w_ring + WST;
      if (++w_tx == TW) { w_tx = 0; if (++w_c == NCH) w_c = 0; }
    };
    const int Hv = p.ups ? 2 * p.Hin : p.Hin, Wv = p.ups ? 2 * p.Win : p.Win;
    // Patch image of a strip: pixel-major, 64 B (one 32-channel chunk) per pixel, pixel P = patch row * PITCH + column.  An
    // LDS-DMA piece is 16 consecutive pixels x 64 B: every lane quad fetches ONE pixel's 64 contiguous bytes (16 segments
    // of 64 B per instruction; the plane-major image of tconv / pconv2 fetches 64 segments of 16 B, and the loaders were
    // issue-bound: 160-700 cycles per piece, profiles/r05_uconv_stamps.log).  The four 16-byte slots of a pixel are
    // XOR-swizzled by (P >> 1) & 3 -- on the SOURCE side, the LDS-DMA destination is linear -- which makes the fragment reads
    // (16 consecutive pixels per 16-lane group, slot = channel group) conflict-free at every tap shift.
    int spix[NPC];
    const int lpix = lane >> 2, lplane = (lane & 3) ^ ((lane >> 3) & 3);   // (P >> 1) & 3 = (lane >> 3) & 3: 16 j is 0 mod 8
    auto strip_pixels = [&](int pass) {                // pass = global pass index of this worker's next pass
      const int sid = pass * U_STRIPS + L;
      const bool sv = sid < nstrips;
      const int sx = sid % SX, t_ = sid / SX, sy = t_ % SY, b = t_ / SY;
      const int y0 = sy * ROWS + p.dy0, x0 = sx * 16 + p.dx0;
#pragma unroll
      for (int j = 0; j < NPC; ++j) {
        const int P = j * 16 + lpix;
        const int py = P / PITCH, px = P - py * PITCH;
        int u = y0 + py, w = x0 + px;
        if (p.border == CSMRI_BORDER_REFLECT) {
          u = u < 0 ? -u : u; u = min(u, 2 * (Hv - 1) - u);
          w = w < 0 ? -w : w; w = min(w, 2 * (Wv - 1) - w);
          u = max(u, 0); w = max(w, 0);                // (filler pixels past the patch may reflect twice: any valid pixel will do)
        }
        const bool ok = sv & ((unsigned)u < (unsigned)Hv) & ((unsigned)w < (unsigned)Wv);
        if (p.ups) { u >>= 1; w >>= 1; }
        spix[j] = ok ? (b * p.Hin + u) * p.Win + w : -1;
      }
    };
    const char* zero_page = u_zero_page;
    auto patch_slice = [&](auto hc, int c, int buf) {  // pieces [0, PPS0) or [PPS0, NPC) of chunk c into pass/chunk buffer `buf`
      constexpr int h = decltype(hc)::value;
      const int ch = c * 32;
      const bool second = ch >= p.c0;                  // wave-uniform (c0 % 32 == 0)
      const char* cb = (second ? p.in1 + (size_t)(ch - p.c0) * 2 : p.in0 + (size_t)ch * 2) + lplane * 16;
      const unsigned psb = (unsigned)(second ? p.ps1 : p.ps0) * 2u;
      char* dbase = smem + buf * PBUF + L * SBUF;
#pragma unroll
      for (int j = h * PPS0; j < (h ? NPC : PPS0); ++j) {
        const char* s = spix[j] >= 0 ? cb + (size_t)((unsigned)spix[j] * psb) : zero_page;
        if (!(UCONV_ABLATE & 2)) __builtin_amdgcn_global_load_lds((ug_t)s, (ul_t)(dbase + j * 1024), 16, 0, 0);
      }
    };
    // prologue: patch of phase 0, stages 0..2
    U_STAMP_DECL;
    strip_pixels(worker);
    patch_slice(std::integral_constant<int, 0>{}, 0, 0);
    patch_slice(std::integral_constant<int, 1>{}, 0, 0);
    issue_w(); issue_w(); issue_w();
    U_STAMP(0);                                        // prologue issue
    u_vmwait<2 * WPI>();                               // patch 0 and stage 0
    U_STAMP(1);                                        // prologue landing
    __builtin_amdgcn_s_barrier();                      // B_init
    U_STAMP(2);
    int i = 0;
    const int nphase = my_passes * NCH;
    for (int ph = 0; ph < nphase; ++ph) {
      const bool next_phase = ph + 1 < nphase;
      const int cn = NCH == 1 ? 0 : ((ph + 1) & 1);    // chunk of the next phase
      u_unroll(std::make_integer_sequence<int, TW>{}, [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        // everything up to stage i+1 (and, in a phase's last slot, the next phase's patch) has landed
        if (i + 3 >= NTOT) u_vmwait<0>();
        else if (j == 0 || j == TW - 1 || !next_phase) u_vmwait<WPI>();
        else if (j == 1) u_vmwait<PPS0 + WPI>();
        else u_vmwait<PPS1 + WPI>();
        U_STAMP(3);                                    // waiting for DMA
        __builtin_amdgcn_s_barrier();                  // B_i
        U_STAMP(4);                                    // waiting for the compute waves
        if constexpr (j < 2) {
          if (next_phase) {
            if (j == 0 && cn == 0) strip_pixels(worker + ((ph + 1) / NCH) * workers);
            patch_slice(std::integral_constant<int, j>{}, cn, (ph + 1) & 1);
          }
        }
        if (w_i < NTOT) issue_w();
        U_STAMP(5);                                    // issuing
        ++i;
      });
    }
    U_STAMP_DUMP;
    return;
  }

  // ===================================================== compute waves =====================================================
  __builtin_amdgcn_s_setprio(2);
  const int r16 = lane & 15, g = lane >> 4;
  const unsigned lds0 = (unsigned)(size_t)(ul_t)smem;
  // fragment address of patch row 0 for filter column tx (pixel c = tx + r16, slot = channel group g ^ swizzle); patch row R
  // adds R * PITCH * 64 (an immediate) and, for odd R, flips slot bit 1: (P >> 1) & 3 = (2 R + (c >> 1)) & 3
  unsigned abase[TW];
#pragma unroll
  for (int t = 0; t < TW; ++t) {
    const int c = t + r16;
    abase[t] = lds0 + wv * SBUF + c * 64 + ((g ^ ((c >> 1) & 3)) << 4);
  }
  const unsigned wbase = lds0 + WOFF + tile_off(r16, g);
  f32x4_t acc[ROWS][NF];
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int i = 0; i < NF; ++i) acc[r][i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float s1[STATS ? NF : 1][4], s2[STATS ? NF : 1][4];
  if constexpr (STATS) {
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) { s1[i][r] = 0.f; s2[i][r] = 0.f; }
  }
  f32x4_t bias[BIAS ? NF : 1];
  if constexpr (BIAS) {
#pragma unroll
    for (int i = 0; i < NF; ++i) bias[i] = *(const f32x4_t*)(p.bias + n0 + i * 16 + g * 4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < NF; ++i) asm volatile("" : "+v"(bias[i]));
  }

  // epilogue arguments, pinned in SGPRs.  (Written as gconv_out_pos(p, ..) the per-lane choice between p.out / p.out2
  // compiled into vector loads of the KERNEL ARGUMENTS themselves -- select of two kernarg addresses, global_load,
  // s_waitcnt vmcnt(0) -- in front of every store.)  The outputs leave through buffer stores: an invalid lane (ragged
  // edge of the strip, the other tensor of the windowed form) carries an offset past the descriptor's range and is
  // dropped by the hardware, so the epilogue is straight-line code the scheduler can place between the MFMAs.
  int e_ops = p.ops, e_o2ps = p.o2ps, e_gps = p.gps, e_Ht = p.Hout_t, e_Wt = p.Wout_t, e_Ho = p.Ho, e_Wo = p.Wo;
  int e_ooy = p.ooy, e_oox = p.oox, e_wy0 = p.win_y0, e_wx0 = p.win_x0, e_wh = p.win_h, e_ww = p.win_w;
  float e_slope = p.slope, e_gslope = p.gslope;
  asm volatile("" : "+s"(e_ops), "+s"(e_o2ps), "+s"(e_gps), "+s"(e_Ht), "+s"(e_Wt));
  asm volatile("" : "+s"(e_Ho), "+s"(e_Wo), "+s"(e_ooy), "+s"(e_oox), "+s"(e_wy0), "+s"(e_wx0), "+s"(e_wh), "+s"(e_ww));
  asm volatile("" : "+s"(e_slope), "+s"(e_gslope));
  const unsigned e_obytes = (unsigned)p.B * (WIN ? p.win_h * p.win_w : p.Hout_t * p.Wout_t) * (unsigned)p.ops * 2u;
  const unsigned e_hbytes = (unsigned)p.B * p.Hout_t * p.Wout_t * (unsigned)p.o2ps * 2u;
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)e_obytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_halo = __builtin_amdgcn_make_buffer_rsrc(WIN ? p.out2 : p.out, 0, (int)(WIN ? e_hbytes : 0u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_gate = __builtin_amdgcn_make_buffer_rsrc((void*)(GATE ? p.gsrc : p.out), 0,
      (int)(GATE ? (unsigned)p.B * (WIN ? p.win_h * p.win_w : p.Hout_t * p.Wout_t) * (unsigned)p.gps * 2u : 0u), 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  u32x4_t W[TH][NF], f[3];
  U_STAMP_DECL;
  __builtin_amdgcn_s_barrier();                        // B_init: patch of phase 0 and stage 0 have landed
  U_STAMP(0);                                          // waiting for the first patch
  u_unroll(std::make_integer_sequence<int, TH * NF>{}, [&](auto kc_) {
    constexpr int k = decltype(kc_)::value;
    auto& W_ = W; const unsigned wb_ = wbase;           // (asm operands alone do not capture)
    U_READ(W_[k / NF][k % NF], wb_, k * 1024);
  });
  {
    const unsigned a0 = abase[0], a1 = abase[0] ^ 32u;
    U_READ(f[0], a0, 0);
    U_READ(f[1], a1, PITCH * 64);
  }
  {
    // (everything issued so far is retired here; the counted waits of the first iteration then simply pass)
    u_unroll(std::make_integer_sequence<int, TH>{}, [&](auto tc) {
      constexpr int ty = decltype(tc)::value;
      u_lgkm<0, NF>(f[0], W[ty], true);
    });
    u_lgkm<0>(f[1]);
  }
  unsigned r_next = WST;                               // ring offset of the stage the NEXT iteration reads

  for (int ps = 0; ps < my_passes; ++ps) {
    // this pass's strip (epilogue coordinates)
    const int sid = (worker + ps * workers) * U_STRIPS + wv;
    const bool sv = sid < nstrips;
    const int sx = sid % SX, t_ = sid / SX, sy = t_ % SY, sb = t_ / SY;
    const int oy0 = sy * ROWS, ox = sx * 16 + r16;
    // output offsets of the strip (bf16 tensors < 2 GiB: uconv_eligible): what does not depend on the row, once per pass
    const int tyb = oy0 + e_ooy, txl = ox + e_oox;                 // tensor coordinates of row 0 (out_sy = out_sx = 1)
    const bool colv = sv && ox < e_Wo;
    const unsigned fpix0 = (unsigned)((sb * e_Ht + tyb) * e_Wt + txl);          // position in the [B, Hout_t, Wout_t] tensor
    const int cx = txl - e_wx0, cy0 = tyb - e_wy0;
    const bool cin_ = !WIN || (unsigned)cx < (unsigned)e_ww;
    const unsigned wpix0 = (unsigned)((sb * e_wh + cy0) * e_ww + cx);          // position in the dense window tensor
    const unsigned lch = (unsigned)(n0 + g * 4) * 2u;                          // this lane's channel quad of fragment 0
    const unsigned ob0 = (WIN ? wpix0 : fpix0) * (unsigned)(e_ops * 2) + lch, obr = (unsigned)((WIN ? e_ww : e_Wt) * e_ops * 2);
    const unsigned hb0 = fpix0 * (unsigned)(e_o2ps * 2) + lch, hbr = (unsigned)(e_Wt * e_o2ps * 2);
    const unsigned gb0 = (WIN ? wpix0 : fpix0) * (unsigned)(e_gps * 2) + lch, gbr = (unsigned)((WIN ? e_ww : e_Wt) * e_gps * 2);
    auto epilogue_row = [&](auto rc_) {
      constexpr int r = decltype(rc_)::value;
      const bool mv = colv && oy0 + r < e_Ho;
      const bool inside = cin_ && (!WIN || (unsigned)(cy0 + r) < (unsigned)e_wh);
      const unsigned offo = (mv && inside) ? ob0 + r * obr : OOB;
      const unsigned offh = (WIN && mv && !inside) ? hb0 + r * hbr : OOB;
      u32x2_t gate[GATE ? NF : 1];
      if constexpr (GATE) {
#pragma unroll
        for (int i = 0; i < NF; ++i)
          gate[i] = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(rs_gate, (int)((mv && inside) ? gb0 + r * gbr + i * 32 : OOB), 0, 0));
      }
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        f32x4_t v = acc[r][i];
        acc[r][i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if (UCONV_ABLATE & 1) { asm volatile("" :: "v"(v)); continue; }
        if constexpr (BIAS) {
          v += bias[i];
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], v[q] * e_slope);     // 0 <= slope <= 1 (uconv_eligible)
        }
        if constexpr (STATS) {
#pragma unroll
          for (int q = 0; q < 4; ++q) { const float vq = mv ? v[q] : 0.f; s1[i][q] += vq; s2[i][q] += vq * vq; }
        }
        if constexpr (GATE) {
          const f32x4_t gs = (f32x4_t){__uint_as_float(gate[i][0] << 16), __uint_as_float(gate[i][0] & 0xffff0000u),
                                       __uint_as_float(gate[i][1] << 16), __uint_as_float(gate[i][1] & 0xffff0000u)};
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = gs[q] > 0.f ? v[q] : v[q] * e_gslope;
        }
        const u32x2_t pk = pack4_bf16(v);
        __builtin_amdgcn_raw_buffer_store_b64(pk, rs_out, (int)(offo + i * 32), 0, 0);
        if constexpr (WIN) __builtin_amdgcn_raw_buffer_store_b64(pk, rs_halo, (int)(offh + i * 32), 0, 0);
      }
    };

    u_unroll(std::make_integer_sequence<int, NIT>{}, [&](auto itc) {
      constexpr int IT = decltype(itc)::value, tx = IT % TW, c = IT / TW;
      constexpr bool LAST = IT == NIT - 1;
      constexpr int txn = (IT + 1) % NIT % TW, cn = (IT + 1) % NIT / TW;
      const int buf = NCH == 1 ? (ps & 1) : c;
      const int bufn = NCH == 1 ? (LAST ? ((ps + 1) & 1) : (ps & 1)) : cn;
      // (the buffer offsets are made opaque here: the compiler otherwise hoists every iteration's four addresses to the top
      //  of the pass and spills them)
      unsigned boff = buf * PBUF, boffn = bufn * PBUF;
      asm volatile("" : "+s"(boff), "+s"(boffn));
      const unsigned pa = abase[tx] + boff, pan = abase[txn] + boffn;
      const unsigned pao = pa ^ 32u, pano = pan ^ 32u;
      const unsigned wn = wbase + r_next;
      r_next = r_next + WST == U_RING * WST ? 0u : r_next + WST;
      u_unroll(std::make_integer_sequence<int, NR>{}, [&](auto rc) {
        constexpr int R = decltype(rc)::value, G = IT * NR + R;
        // (1) the fragment two patch rows ahead (the last two: rows 0, 1 of the next iteration)
        if constexpr (UCONV_ABLATE & 4) { asm volatile("" : "+v"(f[(G + 2) % 3]) : "v"(pa), "v"(pan), "v"(pao), "v"(pano)); }
        else if constexpr (R + 2 < NR) U_READ(f[(G + 2) % 3], ((R & 1) ? pao : pa), (R + 2) * PITCH * 64);
        else U_READ(f[(G + 2) % 3], (((R + 2 - NR) & 1) ? pano : pan), (R + 2 - NR) * PITCH * 64);
        // (2) patch row R (and filter row R, refilled at the end of the previous iteration) have landed; younger in
        //     issue order: rows R+1, R+2 and the refills at the end of steps R-2, R-1
        constexpr int nw1 = (R - 1 + NR) % NR >= ROWS - 1 ? NF : 0, nw2 = (R - 2 + NR) % NR >= ROWS - 1 ? NF : 0;
        if constexpr (R < TH) u_lgkm<2 + nw1 + nw2, NF>(f[G % 3], W[R], true);
        else u_lgkm<2 + nw1 + nw2>(f[G % 3]);
        // (3) output row R-TH of the pass is complete (its last filter row was added a step ago): epilogue between the MFMAs
        if constexpr (LAST && R >= TH) epilogue_row(std::integral_constant<int, R - TH>{});
        // (4) MFMAs: filter row ty adds patch row R into output row R - ty
#pragma unroll
        for (int ty = 0; ty < TH; ++ty) {
          const int r = R - ty;
          if (r < 0 || r >= ROWS) continue;
#pragma unroll
          for (int i = 0; i < NF; ++i)
            acc[r][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, W[ty][i]),
                                                                __builtin_bit_cast(bf16x8_t, f[G % 3]), acc[r][i], 0, 0, 0);
        }
        // (5) B_i: stage i+1 (and, in a phase's last iteration, the next phase's patch) are visible from here on
        if constexpr (R == ROWS - 1) { U_STAMP(1); __builtin_amdgcn_s_barrier(); U_STAMP(2); }   // 1: multiplying, 2: barrier
        // (6) filter row R-(ROWS-1) is dead: refill it for the next iteration
        if constexpr (R >= ROWS - 1) {
          constexpr int ty = R - (ROWS - 1);
          u_unroll(std::make_integer_sequence<int, NF>{}, [&](auto ic) {
            constexpr int i = decltype(ic)::value;
            auto& W_ = W; const unsigned wn_ = wn;
            if constexpr (UCONV_ABLATE & 4) { asm volatile("" : "+v"(W_[ty][i]) : "v"(wn_)); }
            else U_READ(W_[ty][i], wn_, (ty * NF + i) * 1024);
          });
        }
        if constexpr (!LAST) __builtin_amdgcn_sched_barrier(0);
      });
    });
    epilogue_row(std::integral_constant<int, ROWS - 1>{});
    U_STAMP(3);                                        // tail of the pass (from its last barrier): MFMAs + epilogue
    // pass boundary: nothing in flight across the loop's back edge; next pass's rows 0, 1 sit in f[NIT*NR % 3], f[.. + 1]
    u_unroll(std::make_integer_sequence<int, TH>{}, [&](auto tc) {
      constexpr int ty = decltype(tc)::value;
      u_lgkm<0, NF>(f[0], W[ty], true);
    });
    u_lgkm<0>(f[1]); u_lgkm<0>(f[2]);
    constexpr int rot = (NIT * NR) % 3;
    if constexpr (rot == 1) { const u32x4_t t0 = f[0]; f[0] = f[1]; f[1] = f[2]; f[2] = t0; }
    if constexpr (rot == 2) { const u32x4_t t0 = f[0]; f[0] = f[2]; f[2] = f[1]; f[1] = t0; }
  }

  U_STAMP_DUMP;
  if constexpr (STATS) {
    if (p.stats) {
      const size_t RW = (size_t)workers * 4, row = (size_t)worker * 4 + wv;      // [2][Cout][rows]
#pragma unroll
      for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float a1 = s1[i][q], a2 = s2[i][q];
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) { a1 += __shfl_xor(a1, o); a2 += __shfl_xor(a2, o); }
          const int n = n0 + i * 16 + g * 4 + q;
          if (r16 == 0) { p.stats[(size_t)n * RW + row] = a1; p.stats[((size_t)p.Cout + n) * RW + row] = a2; }
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------
#ifndef UCONV_MIN_HW
#define UCONV_MIN_HW (64 * 64)
#endif
#ifndef UCONV_CUS
#define UCONV_CUS 256
#endif
#ifndef UCONV_NF4
#define UCONV_NF4 1      // 64 output channels per workgroup (4 fragments per wave) where no BatchNorm sums are kept
#endif
static int u_mode(const csmri_gconv_desc* d) {
  return (d->stats_partial ? U_STATS : 0) | (d->bias ? U_BIAS : 0) | (d->g_src ? U_GATE : 0) | (d->out_halo ? U_WIN : 0);
}
// channel fragments per wave: 4 (64 channels per workgroup) or 2.  With the BatchNorm partial sums (32 more registers)
// the 4-fragment wave spills 61 registers: those layers take 32-channel blocks (the patch is then read by Cout/32 workgroups)
static int u_nf(const csmri_gconv_desc* d) {
  if (d->Cout % 64) return 2;
  if (d->TH == 3) return 4;
  return (UCONV_NF4 && !d->stats_partial) ? 4 : 2;
}
static void u_grid(const csmri_gconv_desc* d, int* nstrips, int* sx, int* sy, int* nb, int* workers) {
  *sx = (d->Wo + 15) / 16; *sy = (d->Ho + U_ROWS - 1) / U_ROWS;
  *nstrips = d->B * *sx * *sy;
  *nb = d->Cout / (16 * u_nf(d));
  const int npass = (*nstrips + U_STRIPS - 1) / U_STRIPS;
  int maxw = UCONV_CUS / *nb; if (maxw < 1) maxw = 1;
  const int rounds = (npass + maxw - 1) / maxw;
  *workers = (npass + rounds - 1) / rounds;
}

int uconv_eligible(const csmri_gconv_desc* d) {
  if (d->dtype != CSMRI_BF16 || d->in_s != 1 || d->dy_step != 1 || d->dx_step != 1) return 0;
  if (d->nclass > 1 || d->splitk > 1 || d->out_sy != 1 || d->out_sx != 1) return 0;
  if (!(d->Cin == 32 || d->Cin == 64)) return 0;
  if (d->in1 && (d->c0 % 32)) return 0;
  if (!(d->Cout == 32 || d->Cout % 64 == 0)) return 0;
  if ((long long)d->Ho * d->Wo < UCONV_MIN_HW) return 0;
  if ((long long)d->B * d->Hin * d->Win * (d->in0_pix_stride > d->in1_pix_stride ? d->in0_pix_stride : d->in1_pix_stride) * 2 >= (1ll << 31)) return 0;
  const long long out_px = (long long)d->B * d->Hout_t * d->Wout_t;      // 32-bit byte offsets in the epilogue
  if (out_px * d->out_pix_stride * 4 >= (1ll << 31) || out_px * (d->out_halo ? d->halo_pix_stride : 0) * 4 >= (1ll << 31) ||
      out_px * (d->g_src ? d->g_pix_stride : 0) * 4 >= (1ll << 31)) return 0;
  if (d->out_dtype != CSMRI_BF16 || (d->g_src && d->g_dtype != CSMRI_BF16)) return 0;
  const int mode = u_mode(d);
  // the activation slope is applied with the bias only (VGG: bias + ReLU; the U-Net layers are followed by BatchNorm)
  if (!(mode & U_BIAS) && d->act_slope != 1.f) return 0;
  if ((mode & U_BIAS) && !(d->act_slope >= 0.f && d->act_slope <= 1.f)) return 0;
  if (d->TH == 4 && d->TW == 4) return mode == 0 || mode == U_STATS || mode == U_WIN;
  if (d->TH == 3 && d->TW == 3) return d->Cin == 64 && d->Cout % 64 == 0 && (mode == U_BIAS || mode == 0);
  return 0;
}

int uconv_stats_rows(const csmri_gconv_desc* d0) {
  csmri_gconv_desc t = *d0;                          // (asked before the caller has allocated the rows)
  if (!t.stats_partial) t.stats_partial = (float*)16;
  const csmri_gconv_desc* d = &t;
  int nstrips, sx, sy, nb, workers;
  u_grid(d, &nstrips, &sx, &sy, &nb, &workers);
  return workers * 4;
}

template <int TH, int TW, int NCH, int NF, int MODE>
static int launch_uconv(const GParams& p, int grid, hipStream_t st) {
  constexpr int NR = U_ROWS + TH - 1, NPC = (NR * U_PITCH + 15) / 16;
  constexpr int lds = 2 * U_STRIPS * NPC * 1024 + U_RING * TH * NF * 1024;
  auto kern = uconv_kernel<TH, TW, NCH, NF, MODE>;
  CSMRI_SET_MAX_LDS(kern, lds);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, p);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

int uconv_launch(const GParams& p0, const csmri_gconv_desc* d, hipStream_t st) {
  GParams p = p0;
  int nstrips, sx, sy, nb, workers;
  u_grid(d, &nstrips, &sx, &sy, &nb, &workers);
  p.us_n = nstrips; p.us_x = sx; p.us_y = sy; p.ntiles = nb;
  const int grid = workers * nb, mode = u_mode(d), nch = d->Cin / 32, nf = u_nf(d);
#define UC(TH_, NCH_, NF_, MODE_) return launch_uconv<TH_, TH_, NCH_, NF_, MODE_>(p, grid, st)
  if (d->TH == 4) {
    if (mode == U_STATS) { if (nch == 1) UC(4, 1, 2, U_STATS); UC(4, 2, 2, U_STATS); }
    if (mode == U_WIN) {
      if (nch == 1 && nf == 2) UC(4, 1, 2, U_WIN);
      if (nch == 1 && nf == 4) UC(4, 1, 4, U_WIN);
      if (nch == 2 && nf == 2) UC(4, 2, 2, U_WIN);
      UC(4, 2, 4, U_WIN);
    }
    if (nch == 1 && nf == 2) UC(4, 1, 2, 0);
    if (nch == 1 && nf == 4) UC(4, 1, 4, 0);
    if (nch == 2 && nf == 2) UC(4, 2, 2, 0);
    UC(4, 2, 4, 0);
  }
  if (mode == U_BIAS) UC(3, 2, 4, U_BIAS);
  UC(3, 2, 4, 0);
#undef UC
}

void uconv_kernel_name(const csmri_gconv_desc* d, char* buf, int n) {
  snprintf(buf, n, "uconv_kernel<%d, %d, %d, %d, %d>", d->TH, d->TW, d->Cin / 32, u_nf(d), u_mode(d));
}
