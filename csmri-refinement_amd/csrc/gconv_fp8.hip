// gconv_fp8: the implicit-GEMM convolution with fp8 (OCP e4m3fn) operands on the block-scaled MFMA.
//
//   D[n][m] = dq * sum_k W8[n][k] * X8[m][k],   k = (tap, ci),   fp32 accumulate
//
// X8 / W8 are per-tensor-scaled fp8 copies made by fp8.hip (csmri_quantize_fp8); dq = 1/(scale_x*scale_w)
// is read from two device scalars in the epilogue, so the whole sequence is graph-capturable.
// Same skeleton as gconv_glds.hip -- 128 positions x BN channels, 256 threads = 2x2 waves, operand
// tiles staged by LDS-DMA with the XOR swizzle applied on the source side -- but a 128-byte tile row
// now holds 128 K elements instead of 64, and one v_mfma_scale_f32_16x16x128_f8f6f4 (unit block
// scales) consumes the whole row: per K step the same bytes, LDS reads and MFMA cycles as the bf16
// kernel for twice the K.  The instruction contracts the 32 bytes lane group g holds for A with the
// 32 bytes group g holds for B; which 32 K positions of the row those are is free as long as both
// operands agree, so each lane simply concatenates the two 16-byte slots (g, 4+g) it would read for
// the bf16 kernel's two half-steps.
// Needs Cin % 128 == 0 (a K step never straddles a tap), c0 % 16 == 0, Cout % 64 == 0.
#include "common.h"
#include "gconv_params.h"

__device__ __attribute__((aligned(16))) char gf8_zero_page[16];

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(8))) int i32x8_t;

__device__ __forceinline__ int f8_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int BN, int NST>
__global__ __launch_bounds__(256, (NST == 1 && BN == 64) ? 3 : 2) void gconv_fp8_kernel(const GParams p) {
  constexpr int BM = 128, WN = 2;
  constexpr int WTM = 64, WTN = BN / WN, FM = WTM / 16, FN = WTN / 16;
  constexpr int GA = 4, GB = BN / 32;
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
  const int chunk = (lane & 7) ^ ((4 * (wid & 1) + (lane >> 4)) & 7);

  int by[GA], bx[GA], ib[GA];
#pragma unroll
  for (int j = 0; j < GA; ++j) {
    const int m = m0 + (j * 4 + wid) * 8 + lrow;
    if (m < p.M) {
      int b, oy, ox;
      if (p.howo_shift >= 0) { b = m >> p.howo_shift; const int r = m & (HoWo - 1); oy = r >> p.wo_shift; ox = r & (p.Wo - 1); }
      else { b = m / HoWo; const int r = m - b * HoWo; oy = r / p.Wo; ox = r - oy * p.Wo; }
      by[j] = oy * p.S + p.dy0; bx[j] = ox * p.S + p.dx0; ib[j] = b * p.Hin * p.Win;
    } else { by[j] = 0; bx[j] = 0; ib[j] = -1; }
  }
  int k0 = s_begin * 128;
  int tap = k0 / p.Cin, ci = k0 - tap * p.Cin;
  int ty = tap / p.TW, tx = tap - ty * p.TW;
  const char* aptr[GA]; unsigned ainc[GA];
  const bool straddle = (p.c0 & 127) != 0;
  auto compute_ptrs = [&]() {
    const int oy_ = ty * p.dys, ox_ = tx * p.dxs;
    // c0 % 128 == 0: wave-uniform.  Otherwise (c0 % 16 == 0) the K step that contains c0 takes its
    // first 16-byte chunks from in0 and the rest from in1: the choice is per lane (chunk is lane-constant)
    const int cc = ci + chunk * 16;
    const bool second = (straddle ? cc : ci) >= p.c0;
    const char* src = second ? p.in1 + (size_t)(cc - p.c0) : p.in0 + (size_t)cc;
    const size_t ps = (size_t)(second ? p.ps1 : p.ps0);
#pragma unroll
    for (int j = 0; j < GA; ++j) {
      int u = by[j] + oy_, v = bx[j] + ox_;
      bool ok = ib[j] >= 0;
      if (p.border == CSMRI_BORDER_REFLECT) { u = reflect_idx(u, Hv); v = reflect_idx(v, Wv); }
      else ok = ok && (unsigned)u < (unsigned)Hv && (unsigned)v < (unsigned)Wv;
      if (p.ups) { u >>= 1; v >>= 1; }
      const int pix = ib[j] + u * p.Win + v;
      const char* g = p.off32 ? src + (unsigned)pix * (unsigned)ps : src + (size_t)pix * ps;
      aptr[j] = ok ? g : gf8_zero_page;
      ainc[j] = ok ? 128u : 0u;
    }
  };
  compute_ptrs();
  const char* wptr[GB];
#pragma unroll
  for (int j = 0; j < GB; ++j)
    wptr[j] = p.w + (size_t)cls * (size_t)p.wcs + (size_t)(n0 + (j * 4 + wid) * 8 + lrow) * p.Kp + chunk * 16 +
              (size_t)s_begin * 128;

  auto issue = [&](char* buf) {
#pragma unroll
    for (int j = 0; j < GA; ++j) {
      __builtin_amdgcn_global_load_lds((gptr_t)aptr[j], (lptr_t)(buf + (j * 4 + wid) * 1024), 16, 0, 0);
      aptr[j] += ainc[j];
    }
#pragma unroll
    for (int j = 0; j < GB; ++j) {
      __builtin_amdgcn_global_load_lds((gptr_t)wptr[j], (lptr_t)(buf + TILE_Q + (j * 4 + wid) * 1024), 16, 0, 0);
      wptr[j] += 128;
    }
    ci += 128;
    if (ci == p.Cin) { ci = 0; if (++tx == p.TW) { tx = 0; ++ty; } compute_ptrs(); }
    else if (ci == p.c0 || straddle) compute_ptrs();
  };

  f32x4_t acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int r16 = lane & 15, g = lane >> 4;

  auto load_frags = [&](const char* buf, i32x8_t* pf, i32x8_t* qf) {
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      const int row = wn * WTN + i * 16 + r16;
      const u32x4_t lo = *(const u32x4_t*)(buf + TILE_Q + f8_off(row, g));
      const u32x4_t hi = *(const u32x4_t*)(buf + TILE_Q + f8_off(row, 4 + g));
      pf[i] = (i32x8_t){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
    }
#pragma unroll
    for (int j = 0; j < FM; ++j) {
      const int row = wm * WTM + j * 16 + r16;
      const u32x4_t lo = *(const u32x4_t*)(buf + f8_off(row, g));
      const u32x4_t hi = *(const u32x4_t*)(buf + f8_off(row, 4 + g));
      qf[j] = (i32x8_t){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
    }
  };
  auto mma = [&](const i32x8_t* pf, const i32x8_t* qf) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int j = 0; j < FM; ++j)     // cbsz = blgp = 0: both operands e4m3; block scales 2^0 (E8M0 127)
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(pf[i], qf[j], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0,
                                                                    0x7f7f7f7f);
    __builtin_amdgcn_s_setprio(0);
  };

  if constexpr (NST == 2) {
    if (s_begin < s_end) issue(smem);
    for (int s = s_begin; s < s_end; ++s) {
      char* cur = smem + ((s - s_begin) & 1) * BUF;
      char* nxt = smem + (((s - s_begin) & 1) ^ 1) * BUF;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      i32x8_t pf[FN], qf[FM];
      load_frags(cur, pf, qf);
      if (s + 1 < s_end) issue(nxt);
      mma(pf, qf);
    }
  } else {
    for (int s = s_begin; s < s_end; ++s) {
      issue(smem);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      i32x8_t pf[FN], qf[FM];
      load_frags(smem, pf, qf);
      mma(pf, qf);
      __syncthreads();
    }
  }

  // ---- epilogue: dequantise, then the contract of gconv_kernel ---------------------------------
  const float dq = (p.dq0 ? *p.dq0 : 1.f) * (p.dq1 ? *p.dq1 : 1.f);
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
      if (p.dense_out) { op.opix = (size_t)m * p.ops; op.gpix = (size_t)m * p.gps; }
      else {
        int b, oy, ox;
        if (p.howo_shift >= 0) { b = m >> p.howo_shift; const int r = m & (HoWo - 1); oy = r >> p.wo_shift; ox = r & (p.Wo - 1); }
        else { b = m / HoWo; const int r = m - b * HoWo; oy = r / p.Wo; ox = r - oy * p.Wo; }
        op = gconv_out_pos(p, b, oy * p.osy + ooy, ox * p.osx + oox);
      }
    }
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      const int n = n0 + wn * WTN + i * 16 + g * 4;
      f32x4_t v = acc[i][j] * dq;
      if (p.splitk > 1) {
        if (mv) *(f32x4_t*)(p.slab + (((size_t)cls * p.splitk + ks) * p.M + m) * p.Cout + n) = v;
        continue;
      }
      if (!mv) continue;
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
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a = s1[i][r], b = s2[i][r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
        const int n = n0 + wn * WTN + i * 16 + g * 4 + r;
        if (r16 == 0) {
          const size_t R = (size_t)p.mtiles * 2, rr = (size_t)mt * 2 + wm;
          p.stats[(size_t)n * R + rr] = a; p.stats[((size_t)p.Cout + n) * R + rr] = b;
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------
int gconv_fp8_bn(const csmri_gconv_desc* d) { return d->Cout % 128 == 0 ? 128 : 64; }

int gconv_fp8_eligible(const csmri_gconv_desc* d) {
  if (d->dtype != CSMRI_FP8) return 0;
  if (d->Cin % 128 || d->Cout % 64) return 0;
  if (d->in1 && d->c0 % 16) return 0;
  if (d->in0_pix_stride % 16 || (d->in1 && d->in1_pix_stride % 16) || d->Kp % 16) return 0;
  return 1;
}

static int fp8_stages(long long blocks) { return blocks <= 512 ? 2 : 1; }

template <int BN, int NST>
static int launch_fp8(const GParams& p, hipStream_t st) {
  constexpr int lds = (128 + BN) * 128 * NST;
  dim3 grid(p.mtiles * p.ntiles, 1, p.nclass * p.splitk);
  auto kern = gconv_fp8_kernel<BN, NST>;
  CSMRI_SET_MAX_LDS(kern, lds);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

int gconv_fp8_launch(const GParams& p0, const csmri_gconv_desc* d, hipStream_t st) {
  GParams p = p0;
  const int bn = gconv_fp8_bn(d);
  p.nsteps = d->TH * d->TW * d->Cin / 128;
  p.steps_per_split = cdiv(p.nsteps, p.splitk);
  p.mtiles = cdiv(p.M, 128); p.ntiles = d->Cout / bn;
  const long long w_elems = (long long)d->Cout * d->TH * d->TW * d->Cin * p.nclass;
  const long long x_elems = (long long)d->B * d->Hin * d->Win * d->Cin;
  p.nt_major = w_elems > x_elems;
  const int nst = fp8_stages((long long)p.mtiles * p.ntiles * p.nclass * p.splitk);
  if (bn == 128) return nst == 2 ? launch_fp8<128, 2>(p, st) : launch_fp8<128, 1>(p, st);
  return nst == 2 ? launch_fp8<64, 2>(p, st) : launch_fp8<64, 1>(p, st);
}

void gconv_fp8_kernel_name(const csmri_gconv_desc* d, char* buf, int n) {
  const int bn = gconv_fp8_bn(d), nclass = d->nclass > 0 ? d->nclass : 1, sk = d->splitk > 0 ? d->splitk : 1;
  const long long blocks = (long long)cdiv((long long)d->B * d->Ho * d->Wo, 128) * (d->Cout / bn) * nclass * sk;
  snprintf(buf, n, "gconv_fp8_kernel<%d, %d>", bn, fp8_stages(blocks));
}
