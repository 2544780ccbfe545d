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

__device__ __attribute__((aligned(16))) char t_zero_page[16];
typedef __attribute__((address_space(1))) const void* tgptr_t;
typedef __attribute__((address_space(3))) void* tlptr_t;

// LDS image of the input patch: one PLANE per 16-byte channel chunk, pixels 16 bytes apart inside a
// plane ([chunk][pixel][8 ch]).  A ds_read_b128 fragment read (16 consecutive pixels per 16-lane
// k-group) is then conflict-free for EVERY tap shift (planes are a multiple of 256 B apart, and a
// lane group's two k-groups cover complementary pixels of one 16-pixel run), and a tap shift is a
// plain byte offset -- no per-tap swizzle arithmetic.  Patch and weights are staged by LDS-DMA.
#ifndef TCONV_CPS64
#define TCONV_CPS64 4      // 64 MFMAs per wave between barriers in the 64-channel, 64-output streamed loop
#endif
template <int CIN, int FN, bool STREAM>
__global__ __launch_bounds__(256, (CIN <= 32 && FN < 4) ? 4 : 2) void tconv_kernel(const GParams p) {
  constexpr int VPP = CIN / 8;                      // planes (16-byte chunks per pixel)
  constexpr int TPC = CIN >= 32 ? 1 : 32 / CIN;     // taps per K chunk
  constexpr int KCH = CIN >= 32 ? CIN / 32 : 1;     // K chunks per tap
  constexpr int BN = FN * 16;
  constexpr int WT = BN * 64;                       // bytes of one chunk's weight tile
  constexpr int CPS = (STREAM && CIN == 64 && FN == 4) ? TCONV_CPS64 : 2;   // chunks per streamed weight stage
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
#ifdef CSMRI_DBG_STAMPS
  unsigned long long ts[4];
#define T_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts[i]) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define T_STAMP(i) do {} while (0)
#endif
  T_STAMP(0);
  const int tiles_x = (p.Wo + 15) >> 4, tiles_y = (p.Ho + 15) >> 4;
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int b = t / (tiles_x * tiles_y);
  t -= b * tiles_x * tiles_y;
  const int tyi = t / tiles_x, txi = t - tyi * tiles_x;
  const int y0 = tyi * 16, x0 = txi * 16, n0 = blockIdx.y * BN;
  const int S = p.S;                                // 1, or 2 for the 8-channel first layers (patch = 15 S + taps wide)
  const int TPW = 15 * S + p.TW, TPH = 15 * S + p.TH;
  const int Hv = p.ups ? 2 * p.Hin : p.Hin, Wv = p.ups ? 2 * p.Win : p.Win;
  const int npix = TPH * TPW, NG = (npix + 63) >> 6, PLANE = NG << 10;
  char* wl = smem + p.nsteps;                       // weight tiles follow the patch planes

  // ---- patch: global -> LDS by LDS-DMA, once ----------------------------------------------
  {
    const unsigned magic = 0xFFFFFFFFu / (unsigned)TPW + 1u;     // P / TPW for P < 2^16
    for (int grp = 0; grp < NG; ++grp) {
      // instruction (grp, plane) belongs to wave (grp*VPP + plane) & 3
      if (VPP < 4) { const int first = (grp * VPP) & 3; if (wv < first || wv >= first + VPP) continue; }
      const int P = (grp << 6) + lane;
      const int py = (int)__umulhi((unsigned)P, magic), px = P - py * TPW;
      int u = y0 * S + p.dy0 + py, w = x0 * S + p.dx0 + px;
      if (p.border == CSMRI_BORDER_REFLECT) {
        u = u < 0 ? -u : u; u = min(u, 2 * (Hv - 1) - u);
        w = w < 0 ? -w : w; w = min(w, 2 * (Wv - 1) - w);
      }
      const bool ok = (P < npix) & ((unsigned)u < (unsigned)Hv) & ((unsigned)w < (unsigned)Wv);
      if (p.ups) { u >>= 1; w >>= 1; }
      const size_t pix = ((size_t)b * p.Hin + u) * p.Win + w;
      const char* s0 = p.in0 + pix * p.ps0 * 2;
      const char* s1 = p.in1 + pix * p.ps1 * 2;
#pragma unroll
      for (int pl = 0; pl < VPP; ++pl) {
        if (((grp * VPP + pl) & 3) != wv) continue;              // wave-uniform round robin
        const int c = pl * 8;
        const char* src = c < p.c0 ? s0 + c * 2 : s1 + (c - p.c0) * 2;
        src = ok ? src : t_zero_page;
        __builtin_amdgcn_global_load_lds((tgptr_t)src, (tlptr_t)(smem + pl * PLANE + (grp << 10)), 16, 0, 0);
      }
    }
  }
  // ---- weights: [BN][64 B] tiles per 32-wide K chunk (mma_core swizzle applied at the source) --
  const char* wsrc = p.w + (size_t)n0 * p.Kp * 2;
  const int wrow_l = lane >> 2, wslot = lane & 3;
  auto wload = [&](int chunk, char* dst) {            // one chunk: BN/16 instructions, round robin
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      if (((chunk * FN + i) & 3) != wv) continue;
      const int row = i * 16 + wrow_l;
      const int kc = wslot ^ tile_swz(row);
      __builtin_amdgcn_global_load_lds((tgptr_t)(wsrc + ((size_t)row * p.Kp + (size_t)chunk * 32 + kc * 8) * 2),
                                       (tlptr_t)(dst + i * 1024), 16, 0, 0);
    }
  };
  const int groups_x = p.TW / TPC;
  const int nq = p.TH * groups_x * KCH;
  if (!STREAM) {
    for (int q = 0; q < nq; ++q) wload(q, wl + q * WT);
  } else {
#pragma unroll
    for (int c = 0; c < CPS; ++c)
      if (c < nq) wload(c, wl + c * WT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  T_STAMP(1);

  // ---- K loop out of LDS -----------------------------------------------------------------
  f32x4_t acc[FN][4];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int f = 0; f < 4; ++f) acc[i][f] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  // bias of this lane's channel quads: in flight while the patch is staged (the epilogue adds it without a load)
  f32x4_t bb[FN];
#pragma unroll
  for (int i = 0; i < FN; ++i)
    bb[i] = (p.bias && n0 + i * 16 + g * 4 < p.Cout) ? *(const f32x4_t*)(p.bias + n0 + i * 16 + g * 4) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
  // lane constants: fragment f = output row 4*wv+f, pixel r16; k-group g -> (plane, pixel shift)
  const int gplane = CIN >= 32 ? g : (g % VPP), gshift = CIN >= 32 ? 0 : g / VPP;
  int abase[4], wbase[FN];
#pragma unroll
  for (int f = 0; f < 4; ++f) abase[f] = gplane * PLANE + (((4 * wv + f) * S * TPW + r16 * S + gshift) << 4);
#pragma unroll
  for (int i = 0; i < FN; ++i) wbase[i] = tile_off(i * 16 + r16, g);
  int ty = 0, txg = 0, cb = 0;
  // fragment reads of chunk q+1 are issued BEFORE the MFMAs of chunk q (two register sets): in-kernel stamps showed
  // the K loop at a third of the MFMA rate with read -> wait -> MFMA per chunk (tools/stamp_tconv.py)
  auto load_frags = [&](const char* wt, u32x4_t* a, u32x4_t* bw) {   // wt: the chunk's [BN][64 B] weight tile
    const int soff = ((ty * TPW + txg * TPC) << 4) + (CIN >= 32 ? cb * 4 * PLANE : 0);
#pragma unroll
    for (int i = 0; i < FN; ++i) bw[i] = *(const u32x4_t*)(wt + wbase[i]);
#pragma unroll
    for (int f = 0; f < 4; ++f) a[f] = *(const u32x4_t*)(smem + abase[f] + soff);
    if (++cb == KCH) { cb = 0; if (++txg == groups_x) { txg = 0; ++ty; } }
  };
  auto mma = [&](const u32x4_t* a, const u32x4_t* bw) {
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int f = 0; f < 4; ++f)
        acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bw[i]),
                                                            __builtin_bit_cast(bf16x8_t, a[f]), acc[i][f], 0, 0, 0);
  };
  // (the 4-workgroups-per-CU variants with 64 output channels have no registers for a second set: 128-VGPR cap)
  // Measured: +2..6 % on the streamed 64-channel variants, -3 % on the resident-weight ones (more waves per CU hide
  // the read latency there already), nothing on the step.
  constexpr int PF = (STREAM && !(CIN <= 32 && FN == 4)) ? 1 : 0;
  u32x4_t fa[1 + PF][4], fb[1 + PF][FN];
  if (!STREAM) {
    if (!PF) {
      for (int q = 0; q < nq; ++q) { load_frags(wl + q * WT, fa[0], fb[0]); mma(fa[0], fb[0]); }
    } else if (nq > 0) {
      // two register sets alternate with STATIC indices (a runtime index would put the fragments in scratch)
      load_frags(wl, fa[0], fb[0]);
      for (int q = 0;;) {
        if (q + 1 < nq) load_frags(wl + (q + 1) * WT, fa[PF], fb[PF]);
        mma(fa[0], fb[0]);
        if (++q >= nq) break;
        if (q + 1 < nq) load_frags(wl + (q + 1) * WT, fa[0], fb[0]);
        mma(fa[PF], fb[PF]);
        if (++q >= nq) break;
      }
    }
  } else {
    // CPS chunks per stage, two stage buffers: stage st+1 streams in under the MFMAs of stage st
    const int nst = (nq + CPS - 1) / CPS;
    for (int st = 0; st < nst; ++st) {
      char* cur = wl + (st & 1) * CPS * WT;
      char* nxt = wl + ((st & 1) ^ 1) * CPS * WT;
      const int nc = min(CPS, nq - CPS * st);          // chunks of this stage (wave-uniform)
      if (PF) load_frags(cur, fa[0], fb[0]);
#pragma unroll
      for (int c = 0; c < CPS; ++c) {
        const int qn = CPS * (st + 1) + c;             // next stage's chunk c, issued in front of this stage's chunk c
        if (qn < nq) wload(qn, nxt + c * WT);
        if (c < nc) {
          if (!PF) load_frags(cur + c * WT, fa[0], fb[0]);
          else if (c + 1 < nc) load_frags(cur + (c + 1) * WT, fa[(c + 1) & PF], fb[(c + 1) & PF]);
          mma(fa[c & PF], fb[c & PF]);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }

  T_STAMP(2);
  // ---- epilogue (same contract as gconv).  Straight-line: the outputs leave through range-checked buffer stores (an
  // invalid lane, or the other tensor of the windowed form, carries an offset past the descriptor's range), the bias was
  // loaded at the top of the kernel, all gate loads of the tile are in flight before the first is used, and bf16 fragment
  // pairs are exchanged between lane rows so that a lane stores 16 B.  (The common epilogue -- gconv_out_pos, one global load
  // of bias / gate per fragment behind its own vmcnt(0), 8-byte stores under exec branches -- cost the first VGG layer 25 us.)
  const int es = p.out_dt == CSMRI_F32 ? 4 : 2, ges = p.gdt == CSMRI_F32 ? 4 : 2;
  const unsigned opx = (unsigned)p.B * (p.out2 ? p.win_h * p.win_w : p.Hout_t * p.Wout_t);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)(opx * (unsigned)p.ops * es), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_halo = __builtin_amdgcn_make_buffer_rsrc(p.out2 ? p.out2 : p.out, 0,
      (int)(p.out2 ? (unsigned)p.B * p.Hout_t * p.Wout_t * (unsigned)p.o2ps * es : 0u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_gate = __builtin_amdgcn_make_buffer_rsrc((void*)(p.gsrc ? p.gsrc : p.out), 0,
      (int)(p.gsrc ? opx * (unsigned)p.gps * ges : 0u), 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  constexpr bool PAIRS = FN % 2 == 0;                  // 16-byte stores need two fragments (32 channels)
  const bool bf_out = p.out_dt != CSMRI_F32, has_gate = p.gsrc != nullptr, has_act = p.slope != 1.f, has_stats = p.stats != nullptr;
  // channel offset of this lane inside a fragment (pair): own quad, or 8 consecutive channels after the exchange
  const unsigned lch_own = (unsigned)(n0 + g * 4), lch_pair = (unsigned)(n0 + 8 * (g >> 1) + 16 * (g & 1));
  float s1[FN][4], s2[FN][4];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[i][r] = 0.f; s2[i][r] = 0.f; }
  unsigned offo[4], offh[4], offg[4]; bool mvv[4], gated[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const int oy = y0 + 4 * wv + f, ox = x0 + r16;
    const bool mv = oy < p.Ho && ox < p.Wo;
    const int ty_ = oy * p.osy + p.ooy, tx_ = ox * p.osx + p.oox;
    const unsigned fpix = (unsigned)((b * p.Hout_t + ty_) * p.Wout_t + tx_);
    bool inside = true; unsigned opix = fpix;
    if (p.out2) {
      const int cy = ty_ - p.win_y0, cx = tx_ - p.win_x0;
      inside = (unsigned)cy < (unsigned)p.win_h && (unsigned)cx < (unsigned)p.win_w;
      opix = (unsigned)((b * p.win_h + cy) * p.win_w + cx);
    }
    mvv[f] = mv;
    offo[f] = (mv && inside) ? opix * (unsigned)(p.ops * es) : OOB;
    offh[f] = (mv && !inside) ? fpix * (unsigned)(p.o2ps * es) : OOB;
    offg[f] = (mv && inside) ? opix * (unsigned)(p.gps * ges) : OOB;
    gated[f] = inside;            // halo positions leave ungated (csmri_fold_halo gates them where they land)
  }
  f32x4_t gt[4][FN];
  if (has_gate) {
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int i = 0; i < FN; ++i) {
        const unsigned go = offg[f] + (lch_own + i * 16) * ges;
        if (p.gdt == CSMRI_F32) gt[f][i] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rs_gate, (int)go, 0, 0));
        else {
          const u32x2_t u = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(rs_gate, (int)go, 0, 0));
          gt[f][i] = (f32x4_t){__uint_as_float(u[0] << 16), __uint_as_float(u[0] & 0xffff0000u),
                               __uint_as_float(u[1] << 16), __uint_as_float(u[1] & 0xffff0000u)};
        }
      }
  }
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    f32x4_t vv[FN];
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      f32x4_t v = acc[i][f] + bb[i];
      if (n0 + i * 16 + g * 4 >= p.Cout) v = (f32x4_t){0.f, 0.f, 0.f, 0.f};        // channel blocks past Cout: nothing to add up
      if (has_stats) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float q = mvv[f] ? v[r] : 0.f; s1[i][r] += q; s2[i][r] += q * q; }
      }
      if (has_act) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], v[r] * p.slope);            // 0 <= slope <= 1 (tconv_eligible)
      }
      if (has_gate) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (gt[f][i][r] > 0.f || !gated[f]) ? v[r] : v[r] * p.gslope;
      }
      vv[i] = v;
    }
    if (bf_out) {
      if constexpr (PAIRS) {
#pragma unroll
        for (int i = 0; i < FN; i += 2) {
          const u32x2_t a = pack4_bf16(vv[i]), c = pack4_bf16(vv[i + 1]);
          const auto x0_ = __builtin_amdgcn_permlane16_swap(a[0], c[0], false, false);
          const auto x1_ = __builtin_amdgcn_permlane16_swap(a[1], c[1], false, false);
          const u32x4_t d = (u32x4_t){x0_[0], x1_[0], x0_[1], x1_[1]};
          const unsigned ch = (lch_pair + i * 16) * 2u;
          const bool cok = n0 + i * 16 + 8 * (g >> 1) + 16 * (g & 1) < p.Cout;
          __builtin_amdgcn_raw_buffer_store_b128(d, rs_out, (int)(cok ? offo[f] + ch : OOB), 0, 0);
          if (p.out2) __builtin_amdgcn_raw_buffer_store_b128(d, rs_halo, (int)(cok ? offh[f] + ch : OOB), 0, 0);
        }
      } else {
        const u32x2_t a = pack4_bf16(vv[0]);
        const bool cok = n0 + g * 4 < p.Cout;
        __builtin_amdgcn_raw_buffer_store_b64(a, rs_out, (int)(cok ? offo[f] + lch_own * 2u : OOB), 0, 0);
        if (p.out2) __builtin_amdgcn_raw_buffer_store_b64(a, rs_halo, (int)(cok ? offh[f] + lch_own * 2u : OOB), 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < FN; ++i) {
        const unsigned ch = (lch_own + i * 16) * 4u;
        const bool cok = n0 + i * 16 + g * 4 < p.Cout;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, vv[i]), rs_out, (int)(cok ? offo[f] + ch : OOB), 0, 0);
        if (p.out2) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, vv[i]), rs_halo, (int)(cok ? offh[f] + ch : OOB), 0, 0);
      }
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
          const size_t R = (size_t)gridDim.x * 4, r = (size_t)blockIdx.x * 4 + wv;   // [2][Cout][rows]
          p.stats[(size_t)n * R + r] = a1; p.stats[((size_t)p.Cout + n) * R + r] = a2;
        }
      }
  }
#ifdef CSMRI_DBG_STAMPS
  T_STAMP(3);
  if (lane == 0 && p.slab && p.splitk == 1) {
    unsigned long long* dbg = (unsigned long long*)p.slab + ((size_t)blockIdx.x * 4 + wv) * 4;
    dbg[0] = ts[1] - ts[0]; dbg[1] = ts[2] - ts[1]; dbg[2] = ts[3] - ts[2]; dbg[3] = ts[0];
  }
#endif
}

// ---------------------------------------------------------------------------------------------
static int tc_npix(const csmri_gconv_desc* d) { return (15 * d->in_s + d->TH) * (15 * d->in_s + d->TW); }
#ifndef TCONV_MIN_HW
#define TCONV_MIN_HW (64 * 64)
#endif
int tconv_eligible(const csmri_gconv_desc* d) {
  if (d->dtype != CSMRI_BF16 || d->dy_step != 1 || d->dx_step != 1) return 0;
  // stride 2: the discriminator's first layer (one real input channel padded to 8; reference models/discriminators.py:137-150)
  if (!(d->in_s == 1 || (d->in_s == 2 && d->Cin == 8 && !d->upsample && !d->in1))) return 0;
  if ((d->nclass > 1) || d->splitk > 1) return 0;
  if (!(d->Cin == 8 || d->Cin == 16 || d->Cin == 32 || d->Cin == 64)) return 0;
  const int tpc = d->Cin >= 32 ? 1 : 32 / d->Cin;
  if (d->TW % tpc) return 0;
  if (d->out_sy != 1 || d->out_sx != 1) return 0;
  if ((long long)d->Ho * d->Wo < TCONV_MIN_HW) return 0;        // small maps: generic / split-K path
  if (!(d->act_slope >= 0.f && d->act_slope <= 1.f)) return 0;   // epilogue: max(v, slope v), 32-bit byte offsets
  { const long long px = (long long)d->B * d->Hout_t * d->Wout_t;
    if (px * d->out_pix_stride * 4 >= (1ll << 31) || px * (d->out_halo ? d->halo_pix_stride : 0) * 4 >= (1ll << 31) ||
        px * (d->g_src ? d->g_pix_stride : 0) * 4 >= (1ll << 31)) return 0; }
  const size_t lds = (size_t)(d->Cin / 8) * ((tc_npix(d) + 63) / 64) * 1024;
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
  const int npix_ = tc_npix(d);
  const int patch = (CIN / 8) * ((npix_ + 63) / 64) * 1024;      // planes of 64-pixel groups
  const int tpc = CIN >= 32 ? 1 : 32 / CIN, kch = CIN >= 32 ? CIN / 32 : 1;
  const int nq = d->TH * (d->TW / tpc) * kch;
  const int cps = (STREAM && CIN == 64 && FN == 4) ? TCONV_CPS64 : 2;
  const int lds = patch + (STREAM ? 2 * cps : nq) * FN * 16 * 64;
  p.nsteps = patch;                 // tconv reuses this field: byte offset of the weight tiles
  auto kern = tconv_kernel<CIN, FN, STREAM>;
  CSMRI_SET_MAX_LDS(kern, lds);
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
  const int patch_ = (d->Cin / 8) * ((tc_npix(d) + 63) / 64) * 1024;
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
  const int patch_ = (d->Cin / 8) * ((tc_npix(d) + 63) / 64) * 1024;
  const bool stream = patch_ + nq_ * fn * 16 * 64 > 64 * 1024;
  snprintf(buf, n, "tconv_kernel<%d, %d, %s>", d->Cin, fn, stream ? "true" : "false");
}
