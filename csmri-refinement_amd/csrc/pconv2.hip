// pconv2: patch-structured convolution (tconv.hip's idea carried to many channels) as a persistent, role-split, fully
// pipelined kernel -- stride-1 3 x 3 layers with Cin % 64 == 0 and Cout % 128 == 0 from 64 tile blocks on (VGG19
// conv2_1 .. conv5_4 at batch 16 and their data gradients at batch 8; see P2_MIN_BLOCKS).  It replaced round 2's pconv (128 input
// channels) and gconv_glds256 (256 x 256 tiles), which no benchmarked layer reached any more.
//
// Why: the 128 x 128 implicit-GEMM tile of gconv_glds.hip moves 32 KiB from L2 into LDS per 2.1 MFLOP (64 FLOP/B); at
// the 66-73 GB/s per CU an LDS gather from L2 sustains (MI355X_MICROARCH.md, "Indexed rows") that caps it near
// 1.1 PFLOP/s, and the best layers sit at 0.96.  A 16 x 16 output tile x 128 output channels that stages its 18 x 18
// input patch once per 64-channel chunk and streams only the weights per tap moves 185 KiB per 37.7 MFLOP
// (203 FLOP/B): the L2 -> LDS stream stops being the limit.
//
// One workgroup of 12 waves per CU owns one 128-channel block of the output and walks the spatial tiles worker,
// worker + workers, ...; the loop over (tile, 64-channel chunk, tap) is ONE flat sequence of steps (the operand stream
// never drains at a tile boundary).  Roles (an LDS-DMA piece costs its issuing wave 60-185 cycles, MI355X_MICROARCH.md
// cycle constants: waves that multiply must not issue them):
//   waves 8..11  loaders: every step 4 pieces (1 KiB each) of the weight stage three steps ahead into a ring of 4
//                stages (16 KiB: 128 channels x 64 K), plus the next chunk's patch (8 planes of 16 B x patch pixels, two
//                buffers) spread over the chunk's first taps; counted vmcnt retires exactly the stage read next.
//   waves 0..7   4 position groups x 2 channel halves; the waves of channel half 1 run half a step behind those of half
//                0 (ping-pong): on every SIMD one wave is in its 32-MFMA block while the other reads its 16 fragments.
// Two workgroup barriers per step; a buffer is read one phase after the barrier behind the wait that retired it and
// refilled one phase after the barrier behind its last read.
// LDS images: patch = tconv's plane-major image (16-B plane p of pixel q at p * PLANE + 16 q, PLANE % 256 == 0);
// weights = gconv_glds's 128-B rows with the XOR swizzle applied on the source side.  Epilogue = tconv's (bias, leaky slope,
// activation-derivative gate of the data gradient); no BatchNorm partial sums, one input tensor.
#include <utility>
#include "mma_core.h"
#include "gconv_params.h"

__device__ __attribute__((aligned(16))) char p2_zero_page[16];
typedef __attribute__((address_space(1))) const void* p2g_t;
typedef __attribute__((address_space(3))) void* p2l_t;

__device__ __forceinline__ int p2_woff(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int N> __device__ __forceinline__ void p2_vmwait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
#define P2_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

template <int... I, class F>
__device__ __forceinline__ void p2_unroll(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }

#ifdef CSMRI_DBG_STAMPS
#define P2_STAMP(i) do { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
    ph[i] += t_ - last_t; last_t = t_; } while (0)
#define P2_STAMP_DECL unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_t; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last_t) :: "memory")
#define P2_STAMP_DUMP do { if (lane == 0 && p.slab) { unsigned long long* dbg_ = (unsigned long long*)p.slab + ((size_t)blockIdx.x * 12 + wv) * 8; \
    for (int i_ = 0; i_ < 8; ++i_) dbg_[i_] = ph[i_]; } } while (0)
#else
#define P2_STAMP(i) do {} while (0)
#define P2_STAMP_DECL do {} while (0)
#define P2_STAMP_DUMP do {} while (0)
#endif

// F8: operands are fp8 (OCP e4m3fn) -- a 128-byte row of either LDS image then holds 128 K elements instead of 64 and ONE
// v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales) consumes what two bf16 MFMAs did: the same bytes, the same LDS reads,
// the same MFMA cycles per step for twice the K (the instruction contracts the 32 bytes lane group g holds of A with the 32
// bytes it holds of B: the two 16-byte slots (g, 4 + g) a lane read for the bf16 half-steps, concatenated -- gconv_fp8.hip).
// The accumulators are multiplied by *dq0 * *dq1 (the operands' dequantisation scales) before the epilogue.
typedef __attribute__((ext_vector_type(8))) int p2_i32x8_t;
template <int TH, int TW, int BN, bool F8>
__global__ __launch_bounds__(768, 1) void pconv2_kernel(const GParams p) {
  constexpr int ES = F8 ? 1 : 2;                               // bytes per operand element
  constexpr int NT = TH * TW, TPW = 16 + TW - 1, TPH = 16 + TH - 1, NPIX = TPH * TPW;
  constexpr int PLANE = (NPIX * 16 + 255) & ~255, PBUF = 8 * PLANE, WST = BN * 128, NG = 6;
  constexpr int BOFF = 2 * PBUF + 4 * WST;                     // bias of the workgroup's channel block (BN floats)
  constexpr int FN = BN / 32;                                   // 16-channel fragments per compute wave
  constexpr int WP = BN / 32;                                   // weight pieces (8 rows x 128 B) per loader and stage
  static_assert(BN == 128 || BN == 64, "output-channel block");
  constexpr int PPT = NT >= 12 ? 1 : 2, PTAPS = 12 / PPT;      // patch pieces per loader and tap; taps that carry them
  static_assert(NT >= 4 && PTAPS + 3 <= NT && NPIX >= 64 && NPIX <= NG * 64 && 2 * PBUF + 4 * WST + BN * 4 <= 160 * 1024, "patch / ring do not fit");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_x = (p.Wo + 15) >> 4, tiles_y = (p.Ho + 15) >> 4, tiles_img = tiles_x * tiles_y;
  const int ntile = p.mtiles, NB = p.ntiles;
  const int ncc = p.Cin >> (F8 ? 7 : 6);                         // 128-byte channel chunks

  // which output-channel block, which spatial worker: an XCD (workgroup id % 8) keeps ONE channel block's weights in its L2
  int nblk, worker, workers;
  {
    const int id = blockIdx.x, G = gridDim.x;
    if ((8 % NB) == 0 && (G & 7) == 0) {
      const int per = 8 / NB, xcd = id & 7;
      nblk = xcd / per; worker = (id >> 3) * per + (xcd % per); workers = (G >> 3) * per;
    } else { nblk = id % NB; worker = id / NB; workers = G / NB; }
  }
  const int n0 = nblk * BN;
  const int my_tiles = worker < ntile ? (ntile - worker + workers - 1) / workers : 0;
  const int n_chunks = my_tiles * ncc, S = n_chunks * NT;
  if (S == 0) return;

  if (wv >= 8) {
    // =================================================== loader waves ===================================================
    const int L = wv - 8;
    // weights: piece k of a stage = rows 8 (L + 4 k) .. + 7 (8 rows x 128 B), 16-B slots XOR-swizzled at the source
    const int lrow = lane >> 3;
    const int wchunk = (lane & 7) ^ ((4 * (L & 1) + (lane >> 4)) & 7);
    const char* wbase = p.w + (size_t)n0 * p.Kp * ES;
    const size_t wstep = (size_t)32 * p.Kp * ES;
    const char* w0p = wbase; const char* w1p = wbase + wstep; const char* w2p = wbase + 2 * wstep; const char* w3p = wbase + 3 * wstep;
    (void)w2p; (void)w3p;
    const unsigned wvoff = (unsigned)((L * 8 + lrow) * p.Kp * ES + wchunk * 16);
    unsigned cin2 = (unsigned)p.Cin * (unsigned)ES;
    asm volatile("" : "+s"(w0p), "+s"(w1p), "+s"(w2p), "+s"(w3p), "+s"(cin2));
    int w_s = 0, w_tap = 0, w_cc = 0;
    unsigned w_koff = 0, w_ring = 0;
    auto issue_w = [&]() {
      char* dst = smem + 2 * PBUF + w_ring + L * 1024;
      const size_t off = (size_t)(wvoff + w_koff);
      __builtin_amdgcn_global_load_lds((p2g_t)(w0p + off), (p2l_t)dst, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((p2g_t)(w1p + off), (p2l_t)(dst + 4096), 16, 0, 0);
      if constexpr (WP == 4) {
        __builtin_amdgcn_global_load_lds((p2g_t)(w2p + off), (p2l_t)(dst + 8192), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((p2g_t)(w3p + off), (p2l_t)(dst + 12288), 16, 0, 0);
      }
      ++w_s;
      w_ring = (w_ring + WST) & (4 * WST - 1);
      w_koff += cin2;
      if (++w_tap == NT) { w_tap = 0; if (++w_cc == ncc) w_cc = 0; w_koff = (unsigned)w_cc * 128u; }
    };
    // patch: piece k (0..11) of a chunk = plane L + 4 (k / 6), pixel group k % 6
    int p_q = 0, p_cc = 0, p_tile = worker;
    int spix[NG];
    auto patch_pixels = [&](int T) {
      const int b = T / tiles_img, t = T - b * tiles_img;
      const int tyi = t / tiles_x, txi = t - tyi * tiles_x;
      const int y0 = tyi * 16 + p.dy0, x0 = txi * 16 + p.dx0;
#pragma unroll
      for (int grp = 0; grp < NG; ++grp) {
        const int P = (grp * 64 < NPIX - 64 ? grp * 64 : NPIX - 64) + lane;
        const int py = P / TPW, px = P - py * TPW;
        int u = y0 + py, w = x0 + px;
        if (p.border == CSMRI_BORDER_REFLECT) {
          u = u < 0 ? -u : u; u = min(u, 2 * (p.Hin - 1) - u);
          w = w < 0 ? -w : w; w = min(w, 2 * (p.Win - 1) - w);
        }
        const bool ok = ((unsigned)u < (unsigned)p.Hin) & ((unsigned)w < (unsigned)p.Win);
        spix[grp] = ok ? (b * p.Hin + u) * p.Win + w : -1;
      }
    };
    patch_pixels(p_tile);
    // (kept in registers: the compiler would re-load kernel arguments it finds no SGPR for inside the step loop)
    const char* in0 = p.in0 + L * 16;
    unsigned ps0b = (unsigned)p.ps0 * (unsigned)ES;
    const char* zero_page = p2_zero_page;
    asm volatile("" : "+s"(in0), "+s"(ps0b), "+s"(zero_page));
    auto patch_piece = [&](auto kc_) {                 // piece k of chunk p_cc of tile p_tile
      constexpr int k = decltype(kc_)::value, hi = k / 6, grp = k % 6;
      constexpr int px0 = grp * 64 < NPIX - 64 ? grp * 64 : NPIX - 64;
      const char* base = in0 + p_cc * 128 + hi * 64;
      const unsigned ps = ps0b;
      char* dst = smem + (p_q & 1) * PBUF + (L + 4 * hi) * PLANE + px0 * 16;
      const char* s = spix[grp] >= 0 ? base + (size_t)((unsigned)spix[grp] * ps) : zero_page;
      __builtin_amdgcn_global_load_lds((p2g_t)s, (p2l_t)dst, 16, 0, 0);
    };
    auto patch_done = [&]() {
      ++p_q;
      if (++p_cc == ncc) { p_cc = 0; p_tile += workers; if (p_tile < ntile) patch_pixels(p_tile); }
    };
    // prologue: patch 0, stages 0..2
    p2_unroll(std::make_integer_sequence<int, 12>{}, patch_piece);
    patch_done();
    issue_w(); issue_w(); issue_w();
    p2_vmwait<2 * WP>();                               // patch 0 and stage 0 (older than stages 1, 2)
    __builtin_amdgcn_s_barrier();
    int s = 0;
    P2_STAMP_DECL;
    for (int q = 0; q < n_chunks; ++q) {
      const bool patch_here = q + 1 < n_chunks;
      p2_unroll(std::make_integer_sequence<int, NT>{}, [&](auto tc) {
        constexpr int t = decltype(tc)::value;
        // ---- first half of step s: stage s+3 into the buffer stage s-1 left; a slice of the next chunk's patch
        if (w_s < S) issue_w();
        if constexpr (t < PTAPS) {
          if (patch_here) {
            p2_unroll(std::make_integer_sequence<int, PPT>{}, [&](auto jc) { patch_piece(std::integral_constant<int, t * PPT + decltype(jc)::value>{}); });
            if constexpr (t == PTAPS - 1) patch_done();
          }
        }
        P2_STAMP(0);
        __builtin_amdgcn_s_barrier();
        P2_STAMP(1);
        // ---- second half: retire stage s+1 (read from the next first half on).  Younger in issue order: the patch
        // pieces of step s-2, all of steps s-1 and s (WP weight pieces each + that tap's patch pieces)
        constexpr int pp = (t - 2 >= 0 && t - 2 < PTAPS ? PPT : 0) + (t - 1 >= 0 && t - 1 < PTAPS ? PPT : 0) + (t < PTAPS ? PPT : 0);
        if (s + 3 >= S) p2_vmwait<0>();
        else if (patch_here) p2_vmwait<2 * WP + pp>();
        else p2_vmwait<2 * WP>();
        P2_STAMP(2);
        __builtin_amdgcn_s_barrier();
        P2_STAMP(3);
        P2_STAMP(4);
        ++s;
      });
    }
    __builtin_amdgcn_s_barrier();
    P2_STAMP_DUMP;
    return;
  }

  // ===================================================== compute waves =====================================================
  const int wm = wv & 3, wn = wv >> 2;                // wn is also the ping-pong group (one wave of each per SIMD)
  const int r16 = lane & 15, g = lane >> 4;
  const unsigned lds0 = (unsigned)(size_t)(p2l_t)smem;
  f32x4_t acc[FN][4];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int f = 0; f < 4; ++f) acc[i][f] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  // fragment addresses: one VGPR per operand and K half; fragment index and tap are immediate offsets
  const unsigned abase = lds0 + g * PLANE + ((4 * wm * TPW + r16) << 4);
  const unsigned wb0 = lds0 + 2 * PBUF + p2_woff(wn * (BN / 2) + r16, g), wb1 = lds0 + 2 * PBUF + p2_woff(wn * (BN / 2) + r16, 4 + g);
  // Epilogue of a tile: bf16 outputs through range-checked buffer stores (an invalid lane carries an offset past the
  // descriptor's range: no exec branches), fragment pairs exchanged between lane rows so that a lane stores 16 B and a pixel
  // 64 contiguous bytes per instruction, the bias read from LDS, and ALL gate loads of the tile in flight before the first
  // is used.  (The common epilogue it replaced -- gconv_out_pos, one global load of the bias and of the gate per fragment,
  // each behind its own s_waitcnt vmcnt(0), 8-byte stores -- cost the VGG layers 6-26 us per launch with the bias and
  // 3-15 us with the gate, profiles/r05_epilogue_cost.log: 16-32 serial L2 / HBM round trips per tile with all twelve
  // waves of the workgroup waiting at the next barrier.)
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0,
      (int)((unsigned)p.B * p.Hout_t * p.Wout_t * (unsigned)p.ops * 2u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_gate = __builtin_amdgcn_make_buffer_rsrc((void*)(p.gsrc ? p.gsrc : p.out), 0,
      (int)(p.gsrc ? (unsigned)p.B * p.Hout_t * p.Wout_t * (unsigned)p.gps * 2u : 0u), 0x00020000);
  const bool has_bias = p.bias != nullptr, has_gate = p.gsrc != nullptr, has_act = p.slope != 1.f;
  const float e_slope = p.slope, e_gslope = p.gslope;
  const float e_dq = F8 ? (p.dq0 ? *p.dq0 : 1.f) * (p.dq1 ? *p.dq1 : 1.f) : 1.f;
  // fp8 copy of the output (csmri_gconv_desc.out_q): the stored bf16 value times *out_q_scale, rounded to e4m3 by
  // v_cvt_pk_fp8_f32 -- bit for bit what csmri_quantize_fp8 makes of the bf16 tensor -- 8 channels = 8 bytes per lane
  const bool has_q = p.outq != nullptr, has_amax = p.oamax != nullptr;
  const float e_qs = has_q ? *p.oqs : 1.f;
  const __amdgpu_buffer_rsrc_t rs_q = __builtin_amdgcn_make_buffer_rsrc(has_q ? (void*)p.outq : (void*)p.out, 0,
      (int)(has_q ? (unsigned)p.B * p.Hout_t * p.Wout_t * (unsigned)p.oqps : 0u), 0x00020000);
  unsigned amax_bits = 0;
  constexpr unsigned OOB = 0x80000000u;
  const unsigned nown = (unsigned)(n0 + wn * (BN / 2));
  const unsigned lch_st = (nown + 8 * (g >> 1) + 16 * (g & 1)) * 2u;     // store: 8 consecutive channels after the exchange
  const unsigned lch_ld = (nown + g * 4) * 2u;                            // gate: this lane's own channel quad
  auto epilogue = [&](int T) {
    const int b = T / tiles_img, t = T - b * tiles_img;
    const int tyi = t / tiles_x, txi = t - tyi * tiles_x;
    const int y0 = tyi * 16, x0 = txi * 16;
    unsigned opix[4]; bool mvv[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int oy = y0 + 4 * wm + f, ox = x0 + r16;
      mvv[f] = oy < p.Ho && ox < p.Wo;
      opix[f] = (unsigned)((b * p.Hout_t + oy + p.ooy) * p.Wout_t + ox + p.oox);
    }
    u32x2_t gt[4][FN];
    if (has_gate) {
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int i = 0; i < FN; ++i)
          gt[f][i] = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(
              rs_gate, (int)(mvv[f] ? opix[f] * (unsigned)(p.gps * 2) + lch_ld + i * 32 : OOB), 0, 0));
    }
    f32x4_t bb[FN];
#pragma unroll
    for (int i = 0; i < FN; ++i) bb[i] = has_bias ? *(const f32x4_t*)(smem + BOFF + (wn * (BN / 2) + i * 16 + g * 4) * 4) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const unsigned off = mvv[f] ? opix[f] * (unsigned)(p.ops * 2) + lch_st : OOB;
      u32x2_t pk[FN];
#pragma unroll
      for (int i = 0; i < FN; ++i) {
        f32x4_t v = F8 ? acc[i][f] * e_dq + bb[i] : acc[i][f] + bb[i];
        acc[i][f] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if (has_act) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], v[r] * e_slope);      // 0 <= slope <= 1 (pconv2_eligible)
        }
        if (has_gate) {
          const f32x4_t gs = (f32x4_t){__uint_as_float(gt[f][i][0] << 16), __uint_as_float(gt[f][i][0] & 0xffff0000u),
                                       __uint_as_float(gt[f][i][1] << 16), __uint_as_float(gt[f][i][1] & 0xffff0000u)};
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = gs[r] > 0.f ? v[r] : v[r] * e_gslope;
        }
        pk[i] = pack4_bf16(v);
      }
#pragma unroll
      for (int i = 0; i < FN; i += 2) {
        const auto s0 = __builtin_amdgcn_permlane16_swap(pk[i][0], pk[i + 1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(pk[i][1], pk[i + 1][1], false, false);
        const u32x4_t d = (u32x4_t){s0[0], s1[0], s0[1], s1[1]};
        __builtin_amdgcn_raw_buffer_store_b128(d, rs_out, (int)(off + i * 32), 0, 0);
        if (has_q | has_amax) {
          float e[8];
#pragma unroll
          for (int k = 0; k < 4; ++k) { e[2 * k] = __uint_as_float(d[k] << 16); e[2 * k + 1] = __uint_as_float(d[k] & 0xffff0000u); }
          if (has_amax && mvv[f]) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {                   // NaNs are skipped, as csmri_absmax does
              const unsigned ab = __float_as_uint(e[k]) & 0x7fffffffu;
              amax_bits = (ab <= 0x7f800000u && ab > amax_bits) ? ab : amax_bits;
            }
          }
          if (has_q) {
            int q0 = 0, q1 = 0;
            q0 = __builtin_amdgcn_cvt_pk_fp8_f32(sat_e4m3(e[0] * e_qs), sat_e4m3(e[1] * e_qs), q0, false);
            q0 = __builtin_amdgcn_cvt_pk_fp8_f32(sat_e4m3(e[2] * e_qs), sat_e4m3(e[3] * e_qs), q0, true);
            q1 = __builtin_amdgcn_cvt_pk_fp8_f32(sat_e4m3(e[4] * e_qs), sat_e4m3(e[5] * e_qs), q1, false);
            q1 = __builtin_amdgcn_cvt_pk_fp8_f32(sat_e4m3(e[6] * e_qs), sat_e4m3(e[7] * e_qs), q1, true);
            const unsigned qoff = mvv[f] ? opix[f] * (unsigned)p.oqps + (lch_st >> 1) + i * 16 : OOB;
            __builtin_amdgcn_raw_buffer_store_b64((u32x2_t){(unsigned)q0, (unsigned)q1}, rs_q, (int)qoff, 0, 0);
          }
        }
      }
    }
  };

  if (has_bias && tid < BN) ((float*)(smem + BOFF))[tid] = p.bias[n0 + tid];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                       // patch 0 and stage 0 have landed
  if (wn) __builtin_amdgcn_s_barrier();               // channel half 1: half a step behind
  int c_cc = 0, c_tile = worker;
  unsigned r_ring = 0;
  P2_STAMP_DECL;
  for (int q = 0; q < n_chunks; ++q) {
    const unsigned pa = abase + (q & 1) * PBUF;
    p2_unroll(std::make_integer_sequence<int, NT>{}, [&](auto tc) {
      constexpr int t = decltype(tc)::value;
      // ======== read half: this step's 16 fragments ========
      const unsigned w0 = wb0 + r_ring, w1 = wb1 + r_ring;
      r_ring = (r_ring + WST) & (4 * WST - 1);
      u32x4_t a[2][4], b[2][FN];
      constexpr int TAP = ((t / TW) * TPW + (t % TW)) * 16;
#pragma unroll
      for (int i = 0; i < FN; ++i) P2_READ(b[0][i], w0, i * 2048);
#pragma unroll
      for (int f = 0; f < 4; ++f) P2_READ(a[0][f], pa, TAP + f * TPW * 16);
#pragma unroll
      for (int i = 0; i < FN; ++i) P2_READ(b[1][i], w1, i * 2048);
#pragma unroll
      for (int f = 0; f < 4; ++f) P2_READ(a[1][f], pa, TAP + 4 * PLANE + f * TPW * 16);
      // (the reads are complete before the barrier behind which a loader may refill what they read)
      if constexpr (FN == 4)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]), "+v"(a[1][3]),
                       "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[0][2]), "+v"(b[0][3]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[1][2]), "+v"(b[1][3]));
      else
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]), "+v"(a[1][3]),
                       "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]));
      P2_STAMP(0);
      __builtin_amdgcn_s_barrier();
      P2_STAMP(1);
      // ======== multiply half ========
      __builtin_amdgcn_s_setprio(1);
      if constexpr (F8) {
        p2_i32x8_t a8[4], b8[FN];
#pragma unroll
        for (int f = 0; f < 4; ++f)
          a8[f] = (p2_i32x8_t){(int)a[0][f][0], (int)a[0][f][1], (int)a[0][f][2], (int)a[0][f][3],
                               (int)a[1][f][0], (int)a[1][f][1], (int)a[1][f][2], (int)a[1][f][3]};
#pragma unroll
        for (int i = 0; i < FN; ++i)
          b8[i] = (p2_i32x8_t){(int)b[0][i][0], (int)b[0][i][1], (int)b[0][i][2], (int)b[0][i][3],
                               (int)b[1][i][0], (int)b[1][i][1], (int)b[1][i][2], (int)b[1][i][3]};
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
          for (int f = 0; f < 4; ++f)   // cbsz = blgp = 0: both operands e4m3; block scales 2^0 (E8M0 127)
            acc[i][f] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b8[i], a8[f], acc[i][f], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      } else {
#pragma unroll
      for (int kc = 0; kc < 2; ++kc)
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
          for (int f = 0; f < 4; ++f)
            acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, b[kc][i]),
                                                                __builtin_bit_cast(bf16x8_t, a[kc][f]), acc[i][f], 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
      P2_STAMP(2);
      if constexpr (t == NT - 1) {
        if (++c_cc == ncc) { c_cc = 0; epilogue(c_tile); c_tile += workers; }
      }
      P2_STAMP(3);
      __builtin_amdgcn_s_barrier();
      P2_STAMP(4);
      P2_STAMP(5);                                     // (nothing in between: the cost of a stamp)
    });
  }
  if (!wn) __builtin_amdgcn_s_barrier();
  if (has_amax) {
    // One atomic per WAVE on one word was 2,048 same-address atomics per launch: ~11 ns each, 20-40 us on a 60 us kernel.
    // The maximum only grows: a wave first reads the word and skips the atomic when it cannot raise it (a stale read
    // costs an unnecessary atomic, never a wrong result).
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned t_ = __shfl_xor(amax_bits, o); amax_bits = t_ > amax_bits ? t_ : amax_bits; }
    if (lane == 0 && amax_bits) {
      const unsigned cur = __hip_atomic_load(p.oamax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (amax_bits > cur) atomicMax(p.oamax, amax_bits);
    }
  }
  P2_STAMP_DUMP;
}

// ---------------------------------------------------------------------------------------------
static int p2_plane(const csmri_gconv_desc* d) {
  const int npix = (16 + d->TH - 1) * (16 + d->TW - 1);
  return (npix * 16 + 255) & ~255;
}

#ifndef P2_BN64_BELOW
#define P2_BN64_BELOW 256      // fewer 128-channel tile blocks than CUs: 64-channel blocks, twice the workgroups (conv5 at batch 16: 51 -> 30 us, conv4 data gradients at batch 8: 50 -> 37 us; profiles/r05_pconv2_bn64.log)
#endif
static int p2_bn(const csmri_gconv_desc* d) {
  const long long blocks128 = (long long)d->B * ((d->Ho + 15) / 16) * ((d->Wo + 15) / 16) * (d->Cout / 128);
  return blocks128 < P2_BN64_BELOW ? 64 : 128;
}
#ifndef P2_CUS
#define P2_CUS 256
#endif
static void p2_grid(const csmri_gconv_desc* d, int* ntile, int* nb, int* workers) {
  *ntile = d->B * ((d->Ho + 15) / 16) * ((d->Wo + 15) / 16);
  *nb = d->Cout / p2_bn(d);
  int maxw = P2_CUS / *nb; if (maxw < 1) maxw = 1;
  const int rounds = (*ntile + maxw - 1) / maxw;
  *workers = (*ntile + rounds - 1) / rounds;
}

// Smallest grid (tile blocks) that takes this kernel.  Alone, layers with less than a full chip of blocks lose to
// gconv_glds with split-K (half-empty chip, one workgroup per CU); INSIDE the step the VGG branch runs beside the
// discriminator's kernels and the idle CUs are not idle: same-box A/B of the bench step (slices/s resident) 256 -> 128
// +1.1 %, -> 64 +1.9 %, -> 32 +1.9 %, 512 -2 % (tools/ab_old_new.sh, profiles/r04_same_box_ab.json)
#ifndef P2_MIN_BLOCKS
#define P2_MIN_BLOCKS 64
#endif
int pconv2_bn(const csmri_gconv_desc* d) { return p2_bn(d); }
int pconv2_eligible(const csmri_gconv_desc* d) {
  if ((d->dtype != CSMRI_BF16 && d->dtype != CSMRI_FP8) || d->in_s != 1 || d->dy_step != 1 || d->dx_step != 1) return 0;
  const bool f8 = d->dtype == CSMRI_FP8;
  if (f8 && (d->Cin % 128 || d->in0_pix_stride % 16 || d->border != CSMRI_BORDER_ZERO)) return 0;
  if (d->out_q && (long long)d->B * d->Hout_t * d->Wout_t * d->out_q_pix_stride >= (1ll << 31)) return 0;
  if (d->nclass > 1 || d->splitk > 1 || d->upsample || d->in1 || d->stats_partial || d->out_halo) return 0;
  // epilogue: bf16 tensors addressed with 32-bit byte offsets, leaky slope applied as max(v, slope v)
  if (d->out_dtype != CSMRI_BF16 || (d->g_src && d->g_dtype != CSMRI_BF16)) return 0;
  if (!(d->act_slope >= 0.f && d->act_slope <= 1.f)) return 0;
  { const long long px = (long long)d->B * d->Hout_t * d->Wout_t;
    if (px * d->out_pix_stride * 2 >= (1ll << 31) || px * (d->g_src ? d->g_pix_stride : 0) * 2 >= (1ll << 31)) return 0; }
  if (d->TH != 3 || d->TW != 3 || d->out_sy != 1 || d->out_sx != 1) return 0;
  // Measured (profiles/r03_pconv2_layers.log): against gconv_glds / gconv_glds256 / pconv / tconv, alone with warm caches
  // +15 % at 512 input channels (1.20 vs 1.05 PFLOP/s on VGG conv4_x, batch 16), +27 % at 64 -> 128 (VGG conv2_1),
  // +-3 % at 128 / 256; inside the training step every class with a full chip of tile blocks is worth 0.01-0.04 ms
  // (5.61 vs 5.65 ms with all of them), layers with fewer blocks lose (half-empty chip, one workgroup per CU).
  // (64-channel output blocks -- VGG conv1_2 -- were built and measured too: 117 vs tconv's 122 us alone, the step 0.1 ms
  // SLOWER: a wave's 64 x 32 tile reads 12 fragments per 16 MFMAs)
  if (d->Cin % 64 || d->Cout % 128) return 0;
  if ((long long)d->B * d->Hin * d->Win * d->in0_pix_stride * (f8 ? 1 : 2) >= (1ll << 32)) return 0;
  if ((long long)d->Cout * d->TH * d->TW * d->Cin * (f8 ? 1 : 2) >= (1ll << 31)) return 0;
  int ntile, nb, workers;
  p2_grid(d, &ntile, &nb, &workers);
  // (the fp8 form and the fp8-copy / maximum outputs exist in this kernel only: no minimum grid for them)
  if (f8 || d->out_q || d->out_amax) return 1;
  return (long long)ntile * nb >= P2_MIN_BLOCKS;
}

int pconv2_launch(const GParams& p0, const csmri_gconv_desc* d, hipStream_t st) {
  GParams p = p0;
  int ntile, nb, workers;
  p2_grid(d, &ntile, &nb, &workers);
  p.mtiles = ntile; p.ntiles = nb;
  const int bn = p2_bn(d);
  const int lds = 2 * 8 * p2_plane(d) + 4 * bn * 128 + bn * 4;
#define P2_GO(BN_, F8_) do { CSMRI_SET_MAX_LDS((pconv2_kernel<3, 3, BN_, F8_>), 160 * 1024); \
    hipLaunchKernelGGL((pconv2_kernel<3, 3, BN_, F8_>), dim3(workers * nb), dim3(768), lds, st, p); } while (0)
  if (d->dtype == CSMRI_FP8) { if (bn == 64) P2_GO(64, true); else P2_GO(128, true); }
  else { if (bn == 64) P2_GO(64, false); else P2_GO(128, false); }
#undef P2_GO
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}
