// uconv: the U-Net conv class (and VGG conv1_2) -- stride-1 4 x 4 / 3 x 3 layers with 32 or 64 input channels on large
// maps (reference models/unet.py:48,100,241; models/vgg.py:35) -- as a persistent, role-split kernel.  It replaces
// tconv.hip for these shapes: tconv is one-shot (patch DMA -> vmcnt(0) -> barrier -> K loop -> epilogue, one tile per
// workgroup); its stamps (profiles/r05_tconv_stamps_before.log) put 25-46 % of a workgroup's cycles into staging and
// 14-21 % into the epilogue, and its K loop reads 6-8 fragments from LDS per 8-16 MFMAs.
//
// Structure
//   * One workgroup of 8 waves per CU, persistent.  Waves 0..3 multiply, waves 4..7 only move data (an LDS-DMA piece
//     costs its issuing wave 60-185 cycles: MI355X_MICROARCH.md cycle constants).
//   * A PASS = one output tile of 16 rows x 32 columns x NF*16 channels; compute wave (sr, sc) owns the STRIP of 8 rows x
//     16 columns at (8 sr, 16 sc).  The tile's input patch ((16+TH-1) x (32+TW-1) pixels, one 32-channel chunk = 64 B per
//     pixel) is staged ONCE per chunk and shared by the four waves.
//   * K order inside a pass: 32-channel chunk (outer), filter COLUMN tx, then the rows of the strip's patch.  For one
//     (chunk, tx) "iteration" a wave holds the TH x NF weight fragments of that filter column in registers and slides
//     down the patch: the fragment of patch row R (16 pixels x 32 channels, ONE ds_read_b128) feeds the MFMAs of output
//     rows R, R-1, .., R-TH+1 (filter rows 0..TH-1) x NF channel fragments.  LDS reads per MFMA: (ROWS+TH-1 + TH*NF) /
//     (ROWS*TH*NF) = 0.21 (4x4, NF=4), 0.30 (NF=2) against tconv's / pconv2's 0.5-0.75 -- the bound measured on pconv2.
//     The weight registers are refilled IN PLACE for the next iteration: filter row ty is dead after patch row
//     ROWS-1+ty and first needed again at patch row ty of the next iteration, 16-24 MFMAs later.
//   * Weights: one STAGE = one iteration's TH x NF*16 x 32 block (8-16 KiB).  Where all NCH*TW stages fit beside the two
//     patch buffers they are loaded once per workgroup (RESIDENT: every instance but 64 -> 64 channels 4 x 4); otherwise
//     they stream through a ring of 4 stages.  The patch of chunk c of the next pass goes into the buffer chunk c of this
//     pass has left (one chunk: two pass buffers).  Barriers: RESIDENT two per chunk (buffer hand-over), else one per
//     iteration = per 128 MFMAs of a wave (pconv2: two per 32).
//   * Loaders wait with counted vmcnt; a buffer is read after the barrier behind the wait that retired it and refilled
//     after the barrier behind its last read (cdna_hip_programming.md, "Read a staged buffer one phase AFTER ...").
//   * Epilogue of output row r (bias, slope, BatchNorm partial sums, bf16 store) sits between the MFMAs of the pass's
//     last iteration, as soon as the row's last filter row has been added: straight-line code -- the outputs leave through
//     buffer stores whose out-of-range offsets drop invalid lanes -- with fragment pairs exchanged between lane rows
//     (v_permlane16_swap) so that a lane stores 16 B and a pixel 64 contiguous bytes per instruction.  The BatchNorm
//     partial sums live in registers for the whole launch (one stats row per wave).
// Why these choices (measured, profiles/r05_uconv_*): the first version (per-strip patches in tconv's plane-major image,
// weights re-streamed every pass, gconv_out_pos + 8-byte stores) ran at 30-36 cycles per MFMA: its epilogue compiled into
// vector loads of the kernel arguments and exec branches around every store (14,000 cycles per pass), its 64 x 16-byte
// DMA pieces cost the loaders 160-700 cycles each, and the per-CU LDS-DMA intake (patch with a 1.7x halo + all weights
// per pass = 14-21 B/clk of the ~30 B/clk a CU can take) kept the compute waves at the barriers.
// LDS images: patch = pixel-major, 64 B per pixel, the four 16-byte slots XOR-swizzled by (pixel >> 1) & 3 at the DMA
// source (conflict-free ds_read_b128 at every tap shift; an LDS-DMA piece is 16 pixels x 64 contiguous bytes); weights =
// mma_core.h's 64-byte rows, swizzled at the DMA source.
#include <utility>
#include "mma_core.h"
#include "gconv_params.h"

__device__ __attribute__((aligned(16))) char u_zero_page[16];
typedef __attribute__((address_space(1))) const void* ug_t;
typedef __attribute__((address_space(3))) void* ul_t;

#ifndef UCONV_ABLATE
#define UCONV_ABLATE 0     // diagnostic builds only (tools/run/r05_ablate.sh): 1 no epilogue, 2 no DMA, 4 no LDS reads in the loop
#endif
#ifndef UCONV_PRIO
#define UCONV_PRIO 1       // 1: compute waves at priority 2; 0: none; 2: loader waves at priority 3
#endif
#define U_ROWS 8           // output rows of a strip (one compute wave)
#define U_TR 16            // tile = 2 x 2 strips
#define U_TC 32
#define U_PITCH 36         // pixels per patch row in LDS (32 + taps - 1 = 34 or 35 used; even: see the swizzle)
#define U_NPC 44           // LDS-DMA pieces (16 pixels x 64 B) per patch buffer: 11 per loader wave
#define U_LDS (160 * 1024)
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
// (slots 6, 7: the wave's whole life in shader cycles (s_memtime) and in 100 MHz ticks (s_memrealtime): their ratio is
//  the clock the chip held, MI355X_MICROARCH.md 'DVFS give-back' item 6)
#define U_STAMP_DECL unsigned long long stp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_t, c0_, r0_; U_NOW(last_t); c0_ = last_t; \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0_) :: "memory")
#define U_STAMP_DUMP do { unsigned long long c1_, r1_; U_NOW(c1_); asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1_) :: "memory"); \
    stp[6] = c1_ - c0_; stp[7] = r1_ - r0_; \
    if (lane == 0 && p.slab) { unsigned long long* dbg_ = (unsigned long long*)p.slab + ((size_t)blockIdx.x * 8 + wv) * 8; \
    for (int i_ = 0; i_ < 8; ++i_) dbg_[i_] = stp[i_]; } } while (0)
#else
#define U_STAMP(i) do {} while (0)
#define U_STAMP_DECL do {} while (0)
#define U_STAMP_DUMP do {} while (0)
#endif

template <int TH, int TW, int NCH, int NF, int MODE>
__global__ __launch_bounds__(512, 2) void uconv_kernel(const GParams p) {
  constexpr int ROWS = U_ROWS, NR = ROWS + TH - 1, PITCH = U_PITCH, NPC = U_NPC, PBUF = NPC * 1024;
  constexpr int WST = TH * NF * 1024, WOFF = 2 * PBUF;
  constexpr int NIT = NCH * TW;                     // iterations (chunk, tx) per pass
  constexpr bool GATE = (MODE & U_GATE) != 0;
  // (the gated form keeps 8 KiB of LDS for the gate bits: its weights stream)
  constexpr int RFIT = (U_LDS - 2 * PBUF) / WST, RINGN = (RFIT >= NIT && !GATE) ? NIT : 4;
  constexpr int GOFF = 2 * PBUF + RINGN * WST;      // gate bits: [strip][row][lane] words
  constexpr bool RESIDENT = RINGN == NIT;
  constexpr bool RTC = NCH > 2;                     // runtime chunk loop (code size): a block = one chunk's TW iterations
  constexpr int NITU = RTC ? TW : NIT;
  constexpr int WPI = TH * NF / 4, PPL = NPC / 4;   // weight pieces per loader and stage; patch pieces per loader and phase
  // patch pieces a loader issues in slot j of a phase (streaming form; RESIDENT: all in slot 0)
  constexpr int PP0 = RESIDENT ? PPL : (TW == 3 ? 6 : 4), PP1 = RESIDENT ? 0 : (TW == 3 ? 5 : 4), PP2 = PPL - PP0 - PP1;
  constexpr bool STATS = (MODE & U_STATS) != 0, BIAS = (MODE & U_BIAS) != 0, WIN = (MODE & U_WIN) != 0;
  static_assert((TH * NF) % 4 == 0 && (NF % 2) == 0 && TW >= 3 && TW <= 4 && TH >= 3 && NCH >= 1 && NCH <= 16, "shape");
  static_assert((U_TR + TH - 1) * PITCH <= NPC * 16 && U_TC + TW - 1 <= PITCH && (PITCH % 4) == 0 && NPC % 4 == 0, "patch");
  static_assert(2 * PBUF + RINGN * WST + (GATE ? 8192 : 0) <= U_LDS && RINGN >= 3, "LDS");
  static_assert(!GATE || (!RESIDENT && TW == 3 && NF == 4), "gated form: streaming weights, 3 x 3, 64 channels");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int npass = p.us_n, TX = p.us_x, TY = p.us_y, NB = p.ntiles;
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

  // weight stage (tx, chunk c) into ring offset `ring`: wave L_ of its role moves pieces j = L_ (mod 4)
  const int wrow = lane >> 2;
  const int kc = (lane & 3) ^ tile_swz(wrow);          // (tile_swz depends on (row >> 2) & 3 only: the same for row + 16 pn)
  const char* wsrc = p.w + ((size_t)(n0 + wrow) * p.Kp + kc * 8) * 2;
  auto issue_stage = [&](int L_, int tx_, int c_, unsigned ring) {
    char* dst = smem + WOFF + ring + L_ * 1024;
    const unsigned koff = (unsigned)((tx_ * p.Cin + c_ * 32) * 2);
#pragma unroll
    for (int k = 0; k < WPI; ++k) {
      const int j = L_ + 4 * k, ty = j / NF, pn = j % NF;
      const char* src = wsrc + ((size_t)pn * 16 * p.Kp + (size_t)ty * TW * p.Cin) * 2 + koff;
      if (!(UCONV_ABLATE & 2)) __builtin_amdgcn_global_load_lds((ug_t)src, (ul_t)(dst + 4096 * k), 16, 0, 0);
    }
  };

  if (wv >= 4) {
    // =================================================== loader waves ===================================================
    const int L = wv - 4;                              // loader L moves patch pieces and weight pieces j = L (mod 4)
    if (UCONV_PRIO == 2) __builtin_amdgcn_s_setprio(3);
    // (the first RINGN stages are issued by the compute waves, idle in the prologue)
    int w_i = RINGN, w_tx = RINGN % TW, w_c = (RINGN / TW) % NCH;
    unsigned w_ring = 0;
    auto issue_w = [&]() {
      issue_stage(L, w_tx, w_c, w_ring);
      ++w_i;
      w_ring = w_ring + WST == RINGN * WST ? 0u : w_ring + WST;
      if (++w_tx == TW) { w_tx = 0; if (++w_c == NCH) w_c = 0; }
    };
    const int Hv = p.ups ? 2 * p.Hin : p.Hin, Wv = p.ups ? 2 * p.Win : p.Win;
    // piece j = L + 4 k covers pixels 16 j .. 16 j + 15 of the patch (pixel P = patch row * PITCH + column); a lane quad
    // fetches ONE pixel's 64 contiguous bytes, slot s of the quad the channel group s ^ ((P >> 1) & 3)
    int spix[PPL];
    const int lpix = lane >> 2, lplane = (lane & 3) ^ ((lane >> 3) & 3);   // (P >> 1) & 3 = (lane >> 3) & 3: 16 j is 0 mod 8
    auto tile_pixels = [&](int pass) {                 // pass = global pass (= tile) index
      const bool sv = pass < npass;
      const int tx_ = pass % TX, t_ = pass / TX, ty_ = t_ % TY, b = t_ / TY;
      const int y0 = ty_ * U_TR + p.dy0, x0 = tx_ * U_TC + p.dx0;
#pragma unroll
      for (int k = 0; k < PPL; ++k) {
        const int P = (L + 4 * k) * 16 + lpix;
        const int py = P / PITCH, px = P - py * PITCH;
        int u = y0 + py, w = x0 + px;
        if (p.border == CSMRI_BORDER_REFLECT) {
          u = u < 0 ? -u : u; u = min(u, 2 * (Hv - 1) - u);
          w = w < 0 ? -w : w; w = min(w, 2 * (Wv - 1) - w);
          u = max(u, 0); w = max(w, 0);                // (filler pixels past the patch may reflect twice: any valid pixel will do)
        }
        const bool ok = sv & ((unsigned)u < (unsigned)Hv) & ((unsigned)w < (unsigned)Wv);
        if (p.ups) { u >>= 1; w >>= 1; }
        spix[k] = ok ? (b * p.Hin + u) * p.Win + w : -1;
      }
    };
    const char* zero_page = u_zero_page;
    auto patch_pieces = [&](auto k0c, auto k1c, int c, int buf) {   // pieces k0 <= k < k1 of chunk c into buffer `buf`
      constexpr int k0 = decltype(k0c)::value, k1 = decltype(k1c)::value;
      const int ch = c * 32;
      const bool second = ch >= p.c0;                  // wave-uniform (c0 % 32 == 0)
      const char* cb = (second ? p.in1 + (size_t)(ch - p.c0) * 2 : p.in0 + (size_t)ch * 2) + lplane * 16;
      const unsigned psb = (unsigned)(second ? p.ps1 : p.ps0) * 2u;
      char* dbase = smem + buf * PBUF + L * 1024;
#pragma unroll
      for (int k = k0; k < k1; ++k) {
        const char* s = spix[k] >= 0 ? cb + (size_t)((unsigned)spix[k] * psb) : zero_page;
        if (!(UCONV_ABLATE & 2)) __builtin_amdgcn_global_load_lds((ug_t)s, (ul_t)(dbase + k * 4096), 16, 0, 0);
      }
    };
    // Gate bits (GATE): the activation-derivative gate of a data gradient -- out = g_src > 0 ? out : 0 (ReLU: VGG conv1_2,
    // reference models/vgg.py:35 backward) -- costs the compute waves a global load per fragment (8 rows x 4 fragments per
    // pass) exactly where they have no MFMAs left to hide it.  Loader L fetches the gate values of strip L one chunk ahead
    // and leaves ONE 16-bit mask per (row, lane) in LDS; the epilogue reads a word and selects.
    const __amdgpu_buffer_rsrc_t rs_gate = __builtin_amdgcn_make_buffer_rsrc((void*)(GATE ? p.gsrc : p.in0), 0,
        (int)(GATE ? (unsigned)p.B * p.Hout_t * p.Wout_t * (unsigned)p.gps * 2u : 0u), 0x00020000);
    u32x2_t graw[GATE ? U_ROWS : 1][GATE ? NF : 1];
    auto gate_loads = [&](int pass) {                  // the gate values of this loader's strip of tile `pass`
      const int g_ = lane >> 4, r16_ = lane & 15;
      const bool sv = pass < npass;
      const int tx_ = pass % TX, t_ = pass / TX, ty_ = t_ % TY, b = t_ / TY;
      const int oy0 = ty_ * U_TR + (L >> 1) * ROWS, ox = tx_ * U_TC + (L & 1) * 16 + r16_;
      const bool colv = sv && ox < p.Wo;
      const unsigned pix0 = (unsigned)((b * p.Hout_t + oy0 + p.ooy) * p.Wout_t + ox + p.oox);
      const unsigned lch = (unsigned)(n0 + g_ * 4) * 2u;
#pragma unroll
      for (int r = 0; r < U_ROWS; ++r) {
        const unsigned off = (colv && oy0 + r < p.Ho) ? (pix0 + (unsigned)(r * p.Wout_t)) * (unsigned)(p.gps * 2) + lch : 0x80000000u;
#pragma unroll
        for (int i = 0; i < NF; ++i)
          graw[r][i] = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(rs_gate, (int)(off + i * 32), 0, 0));
      }
    };
    auto gate_bits = [&]() {                           // 16 bits per (row, lane): bit 4 i + q = gate of channel 16 i + 4 g + q
      unsigned* gb = (unsigned*)(smem + GOFF) + L * U_ROWS * 64 + lane;
#pragma unroll
      for (int r = 0; r < U_ROWS; ++r) {
        unsigned m = 0;
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const unsigned h = (q & 1) ? (graw[r][i][q >> 1] >> 16) : (graw[r][i][q >> 1] & 0xffffu);
            m |= (((h - 1u) & 0xffffu) < 0x7f80u ? 1u : 0u) << (4 * i + q);      // bf16 value > 0 (not NaN)
          }
        gb[r * 64] = m;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
#define U_IC(v_) std::integral_constant<int, (v_)>{}
    // prologue: patch of phase 0, the first stages (RESIDENT: all of them)
    U_STAMP_DECL;
    tile_pixels(worker);
    patch_pieces(U_IC(0), U_IC(PPL), 0, 0);
    U_STAMP(0);                                        // prologue issue
    u_vmwait<0>();                                     // patch 0
    U_STAMP(1);                                        // prologue landing
    __builtin_amdgcn_s_barrier();                      // B_init
    U_STAMP(2);
    int i = 0;
    const int nphase = my_passes * NCH;
    for (int ph = 0; ph < nphase; ++ph) {
      const bool next_phase = ph + 1 < nphase;
      const int cn = (ph + 1) % NCH;                   // chunk of the next phase
      u_unroll(std::make_integer_sequence<int, TW>{}, [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (RESIDENT) {
          // two hand-overs per phase: after slot 0's barrier the other buffer is free; before the last slot's barrier the
          // next phase's patch has landed
          if constexpr (j == TW - 1) { u_vmwait<0>(); U_STAMP(3); __builtin_amdgcn_s_barrier(); U_STAMP(4); }
          if constexpr (j == 0) {
            U_STAMP(3);
            __builtin_amdgcn_s_barrier();
            U_STAMP(4);
            if (next_phase) {
              if (cn == 0) tile_pixels(worker + ((ph + 1) / NCH) * workers);
              patch_pieces(U_IC(0), U_IC(PPL), cn, (ph + 1) & 1);
            }
            U_STAMP(5);
          }
        } else {
          // everything up to stage i+1 (and, in a phase's last slot, the next phase's patch) has landed.  Younger in issue
          // order than stage i+1 (issued in slot i-3): all of slots i-2 and i-1 (patch pieces first, then the stage)
          if (i + RINGN >= NTOT) u_vmwait<0>();
          else if (j == TW - 1) u_vmwait<WPI>();
          else if (j == 0) { if (ph == 0) u_vmwait<2 * WPI>(); else u_vmwait<PP2 + 2 * WPI>(); }
          else if (j == 1) {
            // (GATE: the 8 x NF gate loads of chunk 0's slot 0 are younger than stage i+1 too)
            if (GATE && (ph % NCH) == 0 && next_phase) u_vmwait<PP0 + 2 * WPI + (GATE ? U_ROWS * NF : 0)>();
            else if (next_phase) u_vmwait<PP0 + 2 * WPI>();
            else u_vmwait<2 * WPI>();
          }
          else { if (next_phase) u_vmwait<PP0 + PP1 + 2 * WPI>(); else u_vmwait<2 * WPI>(); }
          U_STAMP(3);                                  // waiting for DMA
          __builtin_amdgcn_s_barrier();                // B_i
          U_STAMP(4);                                  // waiting for the compute waves
          if constexpr (GATE) {
            // chunk 0, slot 0: this pass's gate values leave; chunk 1, slot 0: their masks go to LDS (published by the
            // next barrier, one iteration before the pass's last, whose epilogue reads them)
            if (j == 0 && (ph % NCH) == 0) gate_loads(worker + (ph / NCH) * workers);
            if (j == 0 && (ph % NCH) == NCH - 1) gate_bits();
          }
          if (next_phase) {
            if constexpr (j == 0) {
              if (cn == 0) tile_pixels(worker + ((ph + 1) / NCH) * workers);
              patch_pieces(U_IC(0), U_IC(PP0), cn, (ph + 1) & 1);
            }
            if constexpr (j == 1) patch_pieces(U_IC(PP0), U_IC(PP0 + PP1), cn, (ph + 1) & 1);
            if constexpr (j == 2 && PP2 > 0 && TW == 4) patch_pieces(U_IC(PP0 + PP1), U_IC(PPL), cn, (ph + 1) & 1);
          }
          if (w_i < NTOT) issue_w();
          U_STAMP(5);                                  // issuing
        }
        ++i;
      });
    }
    U_STAMP_DUMP;
    return;
  }

  // ===================================================== compute waves =====================================================
  if (UCONV_PRIO == 1) __builtin_amdgcn_s_setprio(2);
  const int r16 = lane & 15, g = lane >> 4;
  const int sr = wv >> 1, sc = wv & 1;                 // this wave's strip of the tile
  const unsigned lds0 = (unsigned)(size_t)(ul_t)smem;
  // fragment address of the strip's patch row 0 for filter column tx (pixel column c = 16 sc + tx + r16, slot = channel
  // group g ^ swizzle); patch row R adds R * PITCH * 64 (an immediate) and, for odd R, flips slot bit 1:
  // (P >> 1) & 3 = (2 R + (c >> 1)) & 3 for P = (8 sr + R) * PITCH + c with PITCH = 36
  unsigned abase[TW];
#pragma unroll
  for (int t = 0; t < TW; ++t) {
    const int c = 16 * sc + t + r16;
    abase[t] = lds0 + (8 * sr * PITCH + c) * 64 + ((g ^ ((c >> 1) & 3)) << 4);
  }
  const unsigned wbase = lds0 + WOFF + tile_off(r16, g);
  f32x4_t acc[ROWS][NF];
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
  // (the accumulators start from the bias: no add in the epilogue)
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int i = 0; i < NF; ++i) acc[r][i] = BIAS ? bias[i] : (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // epilogue arguments, pinned in SGPRs.  (Written as gconv_out_pos(p, ..) the per-lane choice between p.out / p.out2
  // compiled into vector loads of the KERNEL ARGUMENTS themselves -- select of two kernarg addresses, global_load,
  // s_waitcnt vmcnt(0) -- in front of every store.)  The outputs leave through buffer stores: an invalid lane (ragged
  // edge of the tile, the other tensor of the windowed form) carries an offset past the descriptor's range and is
  // dropped by the hardware, so the epilogue is straight-line code the scheduler can place between the MFMAs.
  int e_ops = p.ops, e_o2ps = p.o2ps, e_Ht = p.Hout_t, e_Wt = p.Wout_t, e_Ho = p.Ho, e_Wo = p.Wo;
  int e_ooy = p.ooy, e_oox = p.oox, e_wy0 = p.win_y0, e_wx0 = p.win_x0, e_wh = p.win_h, e_ww = p.win_w;
  asm volatile("" : "+s"(e_ops), "+s"(e_o2ps), "+s"(e_Ht), "+s"(e_Wt));
  asm volatile("" : "+s"(e_Ho), "+s"(e_Wo), "+s"(e_ooy), "+s"(e_oox), "+s"(e_wy0), "+s"(e_wx0), "+s"(e_wh), "+s"(e_ww));
  const unsigned e_obytes = (unsigned)p.B * (WIN ? p.win_h * p.win_w : p.Hout_t * p.Wout_t) * (unsigned)p.ops * 2u;
  const unsigned e_hbytes = (unsigned)p.B * p.Hout_t * p.Wout_t * (unsigned)p.o2ps * 2u;
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)e_obytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_halo = __builtin_amdgcn_make_buffer_rsrc(WIN ? p.out2 : p.out, 0, (int)(WIN ? e_hbytes : 0u), 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  u32x4_t W[TH][NF], f[3];
  U_STAMP_DECL;
  // prologue: the first RINGN weight stages (RESIDENT: all of them) while the loaders fetch the first patch
#pragma unroll
  for (int s_ = 0; s_ < RINGN; ++s_) issue_stage(wv, s_ % TW, (s_ / TW) % NCH, (unsigned)(s_ * WST));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                        // B_init: patch of phase 0 and the first stages have landed
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
    // this pass's tile and this wave's strip (epilogue coordinates)
    const int tile = worker + ps * workers;
    const bool sv = tile < npass;
    const int tx_ = tile % TX, t_ = tile / TX, ty_ = t_ % TY, sb = t_ / TY;
    const int oy0 = ty_ * U_TR + sr * ROWS, ox = tx_ * U_TC + sc * 16 + r16;
    // output offsets of the strip (bf16 tensors < 2 GiB: uconv_eligible): what does not depend on the row, once per pass.
    // After the exchange of fragment pairs a lane stores 8 consecutive channels: lane row g holds channels
    // 8 (g >> 1) + 16 (g & 1) .. + 7 of the pair's 32
    const int tyb = oy0 + e_ooy, txl = ox + e_oox;                 // tensor coordinates of row 0 (out_sy = out_sx = 1)
    const bool colv = sv && ox < e_Wo;
    const unsigned fpix0 = (unsigned)((sb * e_Ht + tyb) * e_Wt + txl);          // position in the [B, Hout_t, Wout_t] tensor
    const int cx = txl - e_wx0, cy0 = tyb - e_wy0;
    const bool cin_ = !WIN || (unsigned)cx < (unsigned)e_ww;
    const unsigned wpix0 = (unsigned)((sb * e_wh + cy0) * e_ww + cx);          // position in the dense window tensor
    const unsigned lch = (unsigned)(n0 + 8 * (g >> 1) + 16 * (g & 1)) * 2u;
    const unsigned ob0 = (WIN ? wpix0 : fpix0) * (unsigned)(e_ops * 2) + lch, obr = (unsigned)((WIN ? e_ww : e_Wt) * e_ops * 2);
    const unsigned hb0 = fpix0 * (unsigned)(e_o2ps * 2) + lch, hbr = (unsigned)(e_Wt * e_o2ps * 2);
    auto epilogue_row = [&](auto rc_) {
      constexpr int r = decltype(rc_)::value;
      const bool mv = colv && oy0 + r < e_Ho;
      const bool inside = cin_ && (!WIN || (unsigned)(cy0 + r) < (unsigned)e_wh);
      const unsigned offo = (mv && inside) ? ob0 + r * obr : OOB;
      const unsigned offh = (WIN && mv && !inside) ? hb0 + r * hbr : OOB;
      unsigned gmask = 0xffffu;
      if constexpr (GATE) gmask = ((const unsigned*)(smem + GOFF))[(wv * ROWS + r) * 64 + lane];
      u32x2_t pk[NF];
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        f32x4_t v = acc[r][i];
        acc[r][i] = BIAS ? bias[i] : (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if (UCONV_ABLATE & 1) { asm volatile("" :: "v"(v)); pk[i] = (u32x2_t){0u, 0u}; continue; }
        if constexpr (BIAS) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);               // bias + ReLU (uconv_eligible: slope 0)
        }
        if constexpr (STATS) {
#pragma unroll
          for (int q = 0; q < 4; ++q) { const float vq = mv ? v[q] : 0.f; s1[i][q] += vq; s2[i][q] += vq * vq; }
        }
        if constexpr (GATE) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = (gmask >> (4 * i + q)) & 1u ? v[q] : 0.f;
        }
        pk[i] = pack4_bf16(v);
      }
      if (UCONV_ABLATE & 1) return;
      // fragments i, i+1: lane rows 1, 3 of fragment i change places with lane rows 0, 2 of fragment i+1
#pragma unroll
      for (int i = 0; i < NF; i += 2) {
        const auto x0 = __builtin_amdgcn_permlane16_swap(pk[i][0], pk[i + 1][0], false, false);
        const auto x1 = __builtin_amdgcn_permlane16_swap(pk[i][1], pk[i + 1][1], false, false);
        const u32x4_t d = (u32x4_t){x0[0], x1[0], x0[1], x1[1]};
        __builtin_amdgcn_raw_buffer_store_b128(d, rs_out, (int)(offo + i * 32), 0, 0);
        if constexpr (WIN) __builtin_amdgcn_raw_buffer_store_b128(d, rs_halo, (int)(offh + i * 32), 0, 0);
      }
    };

    // A BLOCK = the iterations unrolled in one piece: the whole pass (NCH <= 2) or one 32-channel chunk (more chunks:
    // the chunk loop is a runtime loop, the epilogue sits in the last chunk's copy of the block)
    auto block = [&](auto lastc, int ph0) {            // ph0: phase (chunk) index of the block's first iteration
    constexpr bool LASTBLOCK = decltype(lastc)::value;
    u_unroll(std::make_integer_sequence<int, NITU>{}, [&](auto itc) {
      constexpr int IT = decltype(itc)::value, tx = IT % TW;
      constexpr bool LAST = LASTBLOCK && IT == NITU - 1;
      constexpr int txn = (tx + 1) % TW;
      // does this iteration carry a workgroup barrier?  streaming: always (stage i+1); RESIDENT: the phase's hand-overs
      constexpr bool BAR = !RESIDENT || tx == 0 || tx == TW - 1;
      const int buf = (ph0 + IT / TW) & 1;             // patch buffer of this phase; the next iteration's after a phase's last
      const int bufn = tx == TW - 1 ? (buf ^ 1) : buf;
      // (the buffer offsets are made opaque here: the compiler otherwise hoists every iteration's four addresses to the top
      //  of the pass and spills them)
      unsigned boff = buf * PBUF, boffn = bufn * PBUF;
      asm volatile("" : "+s"(boff), "+s"(boffn));
      const unsigned pa = abase[tx] + boff, pan = abase[txn] + boffn;
      const unsigned pao = pa ^ 32u, pano = pan ^ 32u;
      const unsigned wn = wbase + r_next;              // (RESIDENT: the ring is the whole pass, RINGN == NIT)
      r_next = r_next + WST == RINGN * WST ? 0u : r_next + WST;
      u_unroll(std::make_integer_sequence<int, NR>{}, [&](auto rc) {
        constexpr int R = decltype(rc)::value, G = IT * NR + R;   // (step index inside the block)
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
        // (5) barrier: the next stage (and, in a phase's last iteration, the next phase's patch) are visible from here on
        if constexpr (R == ROWS - 1 && BAR) { U_STAMP(1); __builtin_amdgcn_s_barrier(); U_STAMP(2); }   // 1: multiplying, 2: barrier
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
    if constexpr (LASTBLOCK) epilogue_row(std::integral_constant<int, ROWS - 1>{});
    U_STAMP(3);                                        // tail of the pass (from its last barrier): MFMAs + epilogue
    // block boundary: nothing in flight across a loop's back edge; the next block's rows 0, 1 sit in f[NITU*NR % 3], f[.. + 1]
    u_unroll(std::make_integer_sequence<int, TH>{}, [&](auto tc) {
      constexpr int ty = decltype(tc)::value;
      u_lgkm<0, NF>(f[0], W[ty], true);
    });
    u_lgkm<0>(f[1]); u_lgkm<0>(f[2]);
    constexpr int rot = (NITU * NR) % 3;
    if constexpr (rot == 1) { const u32x4_t t0 = f[0]; f[0] = f[1]; f[1] = f[2]; f[2] = t0; }
    if constexpr (rot == 2) { const u32x4_t t0 = f[0]; f[0] = f[2]; f[2] = f[1]; f[1] = t0; }
    };
    if constexpr (RTC) {
      for (int c = 0; c < NCH - 1; ++c) block(std::false_type{}, ps * NCH + c);
      block(std::true_type{}, ps * NCH + NCH - 1);
    } else {
      block(std::true_type{}, ps * NCH);
    }
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
  *sx = (d->Wo + U_TC - 1) / U_TC; *sy = (d->Ho + U_TR - 1) / U_TR;
  *nstrips = d->B * *sx * *sy;                       // tiles (= passes)
  *nb = d->Cout / (16 * u_nf(d));
  const int npass = *nstrips;
  int maxw = UCONV_CUS / *nb; if (maxw < 1) maxw = 1;
  const int rounds = (npass + maxw - 1) / maxw;
  *workers = (npass + rounds - 1) / rounds;
}

int uconv_eligible(const csmri_gconv_desc* d) {
  if (d->dtype != CSMRI_BF16 || d->in_s != 1 || d->dy_step != 1 || d->dx_step != 1) return 0;
  if (d->nclass > 1 || d->splitk > 1 || d->out_sy != 1 || d->out_sx != 1) return 0;
  if (!(d->Cin == 32 || d->Cin == 64 || d->Cin == 128 || (d->TH == 3 && (d->Cin == 256 || d->Cin == 512)))) return 0;
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
  if ((mode & U_BIAS) && d->act_slope != 0.f) return 0;
  if (d->TH == 4 && d->TW == 4) return mode == 0 || mode == U_STATS || mode == U_WIN;
  if ((mode & U_GATE) && d->g_slope != 0.f) return 0;           // gate bits: out = g_src > 0 ? out : 0
  if (d->TH == 3 && d->TW == 3) return d->Cin >= 64 && d->Cout % 64 == 0 && (mode == U_BIAS || mode == U_GATE || mode == 0);
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
  constexpr bool GATE = (MODE & U_GATE) != 0;
  constexpr int WST = TH * NF * 1024, NIT = NCH * TW, RFIT = (U_LDS - 2 * U_NPC * 1024) / WST, RINGN = (RFIT >= NIT && !GATE) ? NIT : 4;
  constexpr int lds = 2 * U_NPC * 1024 + RINGN * WST + (GATE ? 8192 : 0);
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
    if (mode == U_STATS) { if (nch == 1) UC(4, 1, 2, U_STATS); if (nch == 2) UC(4, 2, 2, U_STATS); UC(4, 4, 2, U_STATS); }
    if (mode == U_WIN) {
      if (nch == 1 && nf == 2) UC(4, 1, 2, U_WIN);
      if (nch == 1 && nf == 4) UC(4, 1, 4, U_WIN);
      if (nch == 2 && nf == 2) UC(4, 2, 2, U_WIN);
      if (nch == 2) UC(4, 2, 4, U_WIN);
      if (nf == 2) UC(4, 4, 2, U_WIN);
      UC(4, 4, 4, U_WIN);
    }
    if (nch == 1 && nf == 2) UC(4, 1, 2, 0);
    if (nch == 1 && nf == 4) UC(4, 1, 4, 0);
    if (nch == 2 && nf == 2) UC(4, 2, 2, 0);
    if (nch == 2) UC(4, 2, 4, 0);
    if (nf == 2) UC(4, 4, 2, 0);
    UC(4, 4, 4, 0);
  }
#define UC3(NCH_) do { if (mode == U_BIAS) UC(3, NCH_, 4, U_BIAS); if (mode == U_GATE) UC(3, NCH_, 4, U_GATE); UC(3, NCH_, 4, 0); } while (0)
  if (nch == 2) UC3(2);
  if (nch == 4) UC3(4);
  if (nch == 8) UC3(8);
  UC3(16);
#undef UC3
#undef UC
}

void uconv_kernel_name(const csmri_gconv_desc* d, char* buf, int n) {
  snprintf(buf, n, "uconv_kernel<%d, %d, %d, %d, %d>", d->TH, d->TW, d->Cin / 32, u_nf(d), u_mode(d));
}
