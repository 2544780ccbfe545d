// pconv: stride-1 convolutions with 64 or 128 input channels (Cin % 64 == 0; more: see pconv_eligible) and at least 128
// output channels, bf16 -- the VGG19 conv2_2 .. conv4_4 layers and their data gradients.
//
// The 128-row implicit-GEMM kernels (gconv_glds.hip) re-fetch every input row once per filter tap
// and run at the L2 -> LDS bandwidth limit (in-kernel stamps).  This kernel is
// tconv.hip's idea carried to many channels: a workgroup owns a 16 x 16 output tile x 128 output
// channels; per 64-channel chunk it stages the tile's input PATCH (18 x 18 pixels for 3 x 3) once in
// LDS -- border rule and two-source concat applied while loading -- and runs all taps of that chunk
// out of LDS, the weights streaming through two small stage buffers.  L2 -> LDS bytes per FLOP:
// (48 KiB patch + 144 KiB weights) per 37.7 MFLOP against 32 KiB per 2.1 MFLOP: a third.
// LDS image, fragment addressing and epilogue are tconv's (plane-major patch, see there).
#include "mma_core.h"
#include "gconv_params.h"

__device__ __attribute__((aligned(16))) char p_zero_page[16];
typedef __attribute__((address_space(1))) const void* pgptr_t;
typedef __attribute__((address_space(3))) void* plptr_t;

template <int FN>
__global__ __launch_bounds__(256, 2) void pconv_kernel(const GParams p) {
  constexpr int VPP = 8;                            // 16-byte planes per 64-channel chunk
  constexpr int BN = FN * 16;
  constexpr int WT = BN * 64;                       // bytes of one 32-wide K chunk's weight tile
  constexpr int MAXG = 6;                           // 64-pixel groups of the patch (<= 19 x 19 pixels)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int tiles_x = (p.Wo + 15) >> 4, tiles_y = (p.Ho + 15) >> 4;
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int b = t / (tiles_x * tiles_y);
  t -= b * tiles_x * tiles_y;
  const int tyi = t / tiles_x, txi = t - tyi * tiles_x;
  const int y0 = tyi * 16, x0 = txi * 16, n0 = blockIdx.y * BN;
  const int TPW = 16 + p.TW - 1, TPH = 16 + p.TH - 1;
  const int npix = TPH * TPW, NG = (npix + 63) >> 6, PLANE = NG << 10;
  char* wl = smem + p.nsteps;                       // weight stage buffers follow the patch planes
  const int ncc = p.Cin >> 6;

  // this lane's patch pixels (one per 64-pixel group): source pixel index or -1 (zero page)
  int spix[MAXG];
  {
    const unsigned magic = 0xFFFFFFFFu / (unsigned)TPW + 1u;     // P / TPW for P < 2^16
#pragma unroll
    for (int grp = 0; grp < MAXG; ++grp) {
      const int P = (grp << 6) + lane;
      const int py = (int)__umulhi((unsigned)P, magic), px = P - py * TPW;
      int u = y0 + p.dy0 + py, w = x0 + p.dx0 + px;
      if (p.border == CSMRI_BORDER_REFLECT) {
        u = u < 0 ? -u : u; u = min(u, 2 * (p.Hin - 1) - u);
        w = w < 0 ? -w : w; w = min(w, 2 * (p.Win - 1) - w);
      }
      const bool ok = (P < npix) & ((unsigned)u < (unsigned)p.Hin) & ((unsigned)w < (unsigned)p.Win);
      spix[grp] = ok ? (b * p.Hin + u) * p.Win + w : -1;
    }
  }
  auto stage_patch = [&](int cc) {                  // planes wv and wv + 4 of every group
    const int c_lo = cc * 64;
    const bool second = c_lo >= p.c0;               // wave-uniform: c0 % 64 == 0 (host check)
    const char* base = second ? p.in1 + (size_t)(c_lo - p.c0) * 2 : p.in0 + (size_t)c_lo * 2;
    const unsigned ps = (unsigned)(second ? p.ps1 : p.ps0) * 2u;
#pragma unroll
    for (int grp = 0; grp < MAXG; ++grp) {
      if (grp >= NG) break;
      const char* src = base + (size_t)(unsigned)spix[grp] * ps;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int pl = wv + 4 * h;
        const char* s = spix[grp] >= 0 ? src + pl * 16 : p_zero_page;
        __builtin_amdgcn_global_load_lds((pgptr_t)s, (plptr_t)(smem + pl * PLANE + (grp << 10)), 16, 0, 0);
      }
    }
  };
  // weights: [BN][64 B] tile per 32-wide K chunk (mma_core swizzle applied at the source)
  const char* wsrc = p.w + (size_t)n0 * p.Kp * 2;
  const int wrow_l = lane >> 2, wslot = lane & 3;
  auto wload = [&](int kidx, int rr, char* dst) {   // kidx: 32-wide chunk index along K; rr: round-robin phase
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      if (((rr * FN + i) & 3) != wv) continue;
      const int row = i * 16 + wrow_l;
      const int kc = wslot ^ tile_swz(row);
      __builtin_amdgcn_global_load_lds((pgptr_t)(wsrc + ((size_t)row * p.Kp + (size_t)kidx * 32 + kc * 8) * 2),
                                       (plptr_t)(dst + i * 1024), 16, 0, 0);
    }
  };
  const int nq = p.TH * p.TW * 2;                   // 32-wide chunks per 64-channel patch
  const int kpt = p.Cin >> 5;                       // 32-wide chunks per tap along K
  auto kidx = [&](int q, int cc) { return (q >> 1) * kpt + cc * 2 + (q & 1); };

  f32x4_t acc[FN][4];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int f = 0; f < 4; ++f) acc[i][f] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  // lane constants: fragment f = output row 4*wv+f, pixel r16; k-group g -> plane g (+4 for the upper 32 channels)
  int abase[4], wbase[FN];
#pragma unroll
  for (int f = 0; f < 4; ++f) abase[f] = g * PLANE + (((4 * wv + f) * TPW + r16) << 4);
#pragma unroll
  for (int i = 0; i < FN; ++i) wbase[i] = tile_off(i * 16 + r16, g);
  int ty = 0, tx = 0, cb = 0;
  auto compute = [&](const char* wt) {              // wt: this chunk's [BN][64 B] weight tile
    const int soff = ((ty * TPW + tx) << 4) + cb * 4 * PLANE;
    u32x4_t a[4], bw[FN];
#pragma unroll
    for (int i = 0; i < FN; ++i) bw[i] = *(const u32x4_t*)(wt + wbase[i]);
#pragma unroll
    for (int f = 0; f < 4; ++f) a[f] = *(const u32x4_t*)(smem + abase[f] + soff);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int f = 0; f < 4; ++f)
        acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bw[i]),
                                                            __builtin_bit_cast(bf16x8_t, a[f]), acc[i][f], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    if (++cb == 2) { cb = 0; if (++tx == p.TW) { tx = 0; ++ty; } }
  };

  const int nst = nq >> 1;                          // stages of two chunks
  for (int cc = 0; cc < ncc; ++cc) {
    if (cc) __syncthreads();                        // every wave is done with the previous patch
    stage_patch(cc);
    wload(kidx(0, cc), 0, wl);
    wload(kidx(1, cc), 1, wl + WT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ty = 0; tx = 0; cb = 0;
    for (int st = 0; st < nst; ++st) {
      char* cur = wl + (st & 1) * 2 * WT;
      char* nxt = wl + ((st & 1) ^ 1) * 2 * WT;
      if (st + 1 < nst) wload(kidx(2 * st + 2, cc), 0, nxt);
      compute(cur);
      if (st + 1 < nst) wload(kidx(2 * st + 3, cc), 1, nxt + WT);
      compute(cur + WT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }

  // ---- epilogue (tconv's) ----------------------------------------------------------------------
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
    const OutPos op = gconv_out_pos(p, b, oy * p.osy + p.ooy, ox * p.osx + p.oox);
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
      if (p.gsrc && op.g_ok) {
        f32x4_t gs = load4(p.gsrc, op.gpix + n, p.gdt);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gs[r] > 0.f ? v[r] : v[r] * p.gslope;
      }
      store4(op.base, op.opix + n, p.out_dt, v);
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
          const size_t R = (size_t)gridDim.x * 4, r_ = (size_t)blockIdx.x * 4 + wv;   // [2][Cout][rows]
          p.stats[(size_t)n * R + r_] = a1; p.stats[((size_t)p.Cout + n) * R + r_] = a2;
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------
static long long pconv_blocks(const csmri_gconv_desc* d) {
  return (long long)d->B * ((d->Ho + 15) / 16) * ((d->Wo + 15) / 16) * (d->Cout / 128);
}

int pconv_eligible(const csmri_gconv_desc* d) {
  if (d->dtype != CSMRI_BF16 || d->in_s != 1 || d->dy_step != 1 || d->dx_step != 1) return 0;
  if (d->nclass > 1 || d->splitk > 1 || d->upsample) return 0;
  // (64 input channels: +12 % over tconv on VGG conv2_1 in isolation, nothing at step level: left with tconv)
  if (d->Cin % 64 || d->Cin < 128 || d->Cout % 128) return 0;
  // measured (tools/bench_conv.py): +17..25 % over the 128-row kernel at 128 input channels (VGG conv2_2 and
  // its data gradient); at 256+ channels the one-barrier-per-64-MFMA loop with an exposed patch re-stage per
  // chunk loses to gconv_glds / gconv_glds256 (-4..-20 %), so those stay there until this loop is pipelined
  if (d->Cin > 128) return 0;
  if (d->in1 && d->c0 % 64) return 0;
  if (d->out_sy != 1 || d->out_sx != 1) return 0;
  if (d->TH * d->TW < 4 || (16 + d->TH - 1) * (16 + d->TW - 1) > 6 * 64) return 0;
  if ((long long)d->B * d->Hin * d->Win * (d->in0_pix_stride > d->in1_pix_stride ? d->in0_pix_stride : d->in1_pix_stride) * 2 >= (1ll << 32)) return 0;
  return pconv_blocks(d) >= 512;                                // two workgroups per CU
}

int pconv_stats_rows(const csmri_gconv_desc* d) {
  return d->B * ((d->Ho + 15) / 16) * ((d->Wo + 15) / 16) * 4;
}

int pconv_launch(const GParams& p0, const csmri_gconv_desc* d, hipStream_t st) {
  GParams p = p0;
  const int npix_ = (16 + d->TH - 1) * (16 + d->TW - 1);
  const int patch = 8 * ((npix_ + 63) / 64) * 1024;
  const int lds = patch + 4 * 128 * 64;
  p.nsteps = patch;                 // byte offset of the weight stage buffers (as in tconv)
  CSMRI_SET_MAX_LDS(pconv_kernel<8>, 8 * 6 * 1024 + 4 * 128 * 64);
  const int tiles = d->B * ((d->Ho + 15) / 16) * ((d->Wo + 15) / 16);
  dim3 grid(tiles, d->Cout / 128, 1);
  hipLaunchKernelGGL(pconv_kernel<8>, grid, dim3(256), lds, st, p);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}
