// csmri_wgrad: weight gradient of a convolution as an implicit GEMM on MFMA.
//
//   dW[(tap,c)][n] = sum_{pixels m} Xg[m][(tap,c)] * dY[m][n]
//
// The reduction runs over output pixels, which is the NON-contiguous direction of
// both NHWC operands.  MFMA fragments need K-contiguous data, so each thread loads
// a VE x VE block (VE pixels x VE channels, 16 B per pixel), transposes it in
// registers (v_perm_b32 for bf16, free renaming for fp32) and writes VE rows of
// "channel-major, pixel-contiguous" data into the same swizzled LDS tile layout the
// forward kernel uses; the MFMA loop is shared (mma_core.h).
// Pixels are split over grid.z; every split writes an fp32 slab, a second kernel
// sums the slabs in fixed order (deterministic) and accumulates into the fp32
// gradient in the reference's [Cout][Cin][KH][KW] layout.
#include "mma_core.h"
#include "wgrad_params.h"


template <int DT>
__device__ __forceinline__ void transpose_block(const u32x4_t* in, u32x4_t* out) {
  if constexpr (DT == CSMRI_BF16) {
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int d = 0; d < 4; ++d)
        out[c][d] = __builtin_amdgcn_perm(in[2 * d + 1][c >> 1], in[2 * d][c >> 1],
                                          (c & 1) ? 0x07060302u : 0x05040100u);
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) out[c][e] = in[e][c];
  }
}

template <int DT, int BP, int BQ, int WP, int WQ>
__global__ __launch_bounds__(256) void wgrad_kernel(const WParams p) {
  using Tr = DTraits<DT>;
  constexpr int VE = Tr::VE, BKE = Tr::BKE, ES = Tr::ES;
  constexpr int KC = DT == CSMRI_BF16 ? 2 : 1;
  constexpr int PS = KC * BKE;             // pixels per K step
  constexpr int PVEC = BP / VE, QVEC = BQ / VE;
  constexpr int NBLK = BP + BQ;            // VE x VE blocks per step
  constexpr int NI = (NBLK + 255) / 256;
  constexpr int WTP = BP / WP, WTQ = BQ / WQ, FP = WTP / 16, FQ = WTQ / 16;
  constexpr int TILE_P = BP * 64, TILE_Q = BQ * 64;
  constexpr int BUF = KC * (TILE_P + TILE_Q);
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wp = wid / WQ, wq = wid % WQ;
  const int t = blockIdx.x;
  const int ptile = t / p.qtiles, qtile = t - ptile * p.qtiles;
  const int p0 = ptile * BP, q0 = qtile * BQ;
  const int ks = blockIdx.z;
  const int s_begin = ks * p.steps_per_split;
  const int s_end = min(p.nsteps, s_begin + p.steps_per_split);
  const int HoWo = p.Ho * p.Wo;
  const int Hv = p.ups ? 2 * p.Hin : p.Hin, Wv = p.ups ? 2 * p.Win : p.Win;

  // static block assignment
  bool isP[NI], act[NI];
  int cvec[NI], pg[NI], dy_[NI], dx_[NI], ci_[NI];
#pragma unroll
  for (int it = 0; it < NI; ++it) {
    const int v = tid + it * 256;
    act[it] = v < NBLK;
    isP[it] = v < BP;
    if (isP[it]) { cvec[it] = v % PVEC; pg[it] = v / PVEC; }
    else { const int w = v - BP; cvec[it] = w % QVEC; pg[it] = w / QVEC; }
    dy_[it] = 0; dx_[it] = 0; ci_[it] = 0;
    if (isP[it]) {
      const int col = p0 + cvec[it] * VE;
      if (col < p.NK) {
        const int tap = col / p.Cin;
        ci_[it] = col - tap * p.Cin;
        const int ky = tap / p.KW, kx = tap - ky * p.KW;
        dy_[it] = ky - p.pt; dx_[it] = kx - p.pl;
      } else act[it] = false;
    } else {
      if (q0 + cvec[it] * VE >= p.Cout) act[it] = false;
    }
  }

  // pixel coordinates of each block's first pixel, advanced incrementally (PS pixels per step)
  int sb[NI], soy[NI], sox[NI];
#pragma unroll
  for (int it = 0; it < NI; ++it) {
    const int mb = s_begin * PS + pg[it] * VE;
    sb[it] = mb / HoWo;
    const int r = mb - sb[it] * HoWo;
    soy[it] = r / p.Wo; sox[it] = r - soy[it] * p.Wo;
  }
  u32x4_t blk[NI][VE];
  auto load_step = [&](int s) {
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const int mb = s * PS + pg[it] * VE;
      int b = sb[it], oy = soy[it], ox = sox[it];
      sox[it] += PS;
      while (sox[it] >= p.Wo) { sox[it] -= p.Wo; if (++soy[it] == p.Ho) { soy[it] = 0; ++sb[it]; } }
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        u32x4_t v = (u32x4_t){0u, 0u, 0u, 0u};
        const bool mv = act[it] && (mb + e) < p.M;
        if (mv) {
          if (isP[it]) {
            int u = oy * p.S + dy_[it], w = ox * p.S + dx_[it];
            bool ok = true;
            if (p.border == CSMRI_BORDER_REFLECT) { u = reflect_idx(u, Hv); w = reflect_idx(w, Wv); }
            else ok = (unsigned)u < (unsigned)Hv && (unsigned)w < (unsigned)Wv;
            if (p.ups) { u >>= 1; w >>= 1; }
            if (ok) {
              const size_t px = (size_t)b * p.Hin * p.Win + (size_t)u * p.Win + w;
              const int c = ci_[it];
              const char* src = (c < p.c0) ? p.in0 + (px * p.ps0 + c) * ES
                                           : p.in1 + (px * p.ps1 + (c - p.c0)) * ES;
              v = *(const u32x4_t*)src;
            }
          } else {
            v = *(const u32x4_t*)(p.dy + ((size_t)(mb + e) * p.dyps + q0 + cvec[it] * VE) * ES);
          }
        }
        blk[it][e] = v;
        if (++ox == p.Wo) { ox = 0; if (++oy == p.Ho) { oy = 0; ++b; } }
      }
    }
  };
  auto store_step = [&](int buf) {
    char* base = smem + buf * BUF;
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      if (tid + it * 256 >= NBLK) continue;
      u32x4_t tr[VE];
      transpose_block<DT>(blk[it], tr);
      const int kc = pg[it] >> 2, chunk = pg[it] & 3;
      char* tile = isP[it] ? base + kc * TILE_P : base + KC * TILE_P + kc * TILE_Q;
#pragma unroll
      for (int c = 0; c < VE; ++c)
        *(u32x4_t*)(tile + tile_off(cvec[it] * VE + c, chunk)) = tr[c];
    }
  };

  MmaCore<DT, FP, FQ> core;
  core.zero();
  if (s_begin < s_end) {
    load_step(s_begin);
    store_step(0);
    __syncthreads();
    for (int s = s_begin; s < s_end; ++s) {
      const int cur = (s - s_begin) & 1;
      const bool more = s + 1 < s_end;
      if (more) load_step(s + 1);
      const char* base = smem + cur * BUF;
#pragma unroll
      for (int kc = 0; kc < KC; ++kc)
        core.step(base + kc * TILE_P, base + KC * TILE_P + kc * TILE_Q, wp * WTP, wq * WTQ, lane);
      if (more) store_step(cur ^ 1);
      __syncthreads();
    }
  }
  const int g = lane >> 4, r16 = lane & 15;
#pragma unroll
  for (int j = 0; j < FQ; ++j) {
    const int co = q0 + wq * WTQ + j * 16 + r16;
    if (co >= p.Cout) continue;
#pragma unroll
    for (int i = 0; i < FP; ++i) {
      const int col = p0 + wp * WTP + i * 16 + g * 4;
      if (col < p.NK)
        *(f32x4_t*)(p.slab + ((size_t)ks * p.Cout + co) * p.NK + col) = core.acc[i][j];
    }
  }
}

// Slab reductions.  The bodies are device functions over a virtual block index so that the single-layer kernels
// (one launch per csmri_wgrad call) and the multi-layer finish kernel (one launch for all layers of a backward pass,
// csmri_wgrad_finish_multi) run exactly the same arithmetic per output element.
__device__ __forceinline__ void wgrad_scatter_body(float (*red)[33], long long vb, long long nvb, const float* slab,
                                                   int splitk, int Cout, int NK, int Cin, int KH, int KW,
                                                   int Cout_real, int Cin_real, float* dw, int accumulate) {
  // 32 consecutive slab elements x 8 split groups per workgroup: the reads of a split coalesce
  // (128-byte runs), the 8 groups keep 8x more loads in flight and are combined in fixed order;
  // the scattered fp32 write into [Cout][Cin][KH][KW] is the small side
  const int e = threadIdx.x & 31, zg = threadIdx.x >> 5;
  const long long total = (long long)Cout_real * NK;
  const size_t zs = (size_t)Cout * NK;
  for (long long base = vb * 32; base < total; base += nvb * 32) {
    const long long i = base + e;
    float s0 = 0.f, s1 = 0.f;
    if (i < total) {
      const float* p = slab + i;
      int z = zg;
      for (; z + 8 < splitk; z += 16) { s0 += p[(size_t)z * zs]; s1 += p[(size_t)(z + 8) * zs]; }
      if (z < splitk) s0 += p[(size_t)z * zs];
    }
    red[zg][e] = s0 + s1;
    __syncthreads();
    if (zg == 0 && i < total) {
      float s = red[0][e];
#pragma unroll
      for (int g = 1; g < 8; ++g) s += red[g][e];
      const int co = (int)(i / NK), k = (int)(i - (long long)co * NK);
      const int tap = k / Cin, ci = k - tap * Cin;
      if (ci < Cin_real) {
        const size_t o = ((size_t)co * Cin_real + ci) * KH * KW + tap;
        dw[o] = accumulate ? dw[o] + s : s;
      }
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void wgrad_scatter_kernel(const float* slab, int splitk, int Cout, int NK, int Cin,
                                                            int KH, int KW, int Cout_real, int Cin_real, float* dw,
                                                            int accumulate) {
  __shared__ float red[8][33];
  wgrad_scatter_body(red, blockIdx.x, gridDim.x, slab, splitk, Cout, NK, Cin, KH, KW, Cout_real, Cin_real, dw,
                     accumulate);
}

// Split-K sum + layout change [co][tap][ci] -> [co][ci][tap] through an LDS transpose so that
// both the slab reads (64 consecutive channels) and the gradient writes (64*taps consecutive
// floats) are coalesced.  One block per (co, 64-channel chunk); the z (split) loop is spread
// over 4 thread groups and combined in a fixed order.
__device__ __forceinline__ void wgrad_scatter_t_body(float (*tile)[65], int co, int ci0, const float* slab, int splitk,
                                                     int Cout, int NK, int Cin, int taps, int Cin_real, float* dw,
                                                     int accumulate) {
  const size_t zs = (size_t)Cout * NK;
  for (int t0 = 0; t0 < taps; t0 += 16) {
    const int nt = min(16, taps - t0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = threadIdx.x + 256 * j, t = e >> 6, c = e & 63;
      if (t < nt && ci0 + c < Cin) {
        const float* q = slab + (size_t)co * NK + (size_t)(t0 + t) * Cin + ci0 + c;
        float s = 0.f;
        for (int z = 0; z < splitk; ++z) s += q[(size_t)z * zs];
        tile[t][c] = s;
      }
    }
    __syncthreads();
    const int nc = min(64, Cin_real - ci0);
    for (int idx = threadIdx.x; idx < nc * nt; idx += 256) {
      const int cc = idx / nt, t = idx - cc * nt;
      const size_t o = ((size_t)co * Cin_real + ci0 + cc) * taps + t0 + t;
      dw[o] = accumulate ? dw[o] + tile[t][cc] : tile[t][cc];
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void wgrad_scatter_t_kernel(const float* slab, int splitk, int Cout, int NK,
                                                              int Cin, int taps, int Cin_real, float* dw,
                                                              int accumulate) {
  __shared__ float tile[16][65];
  wgrad_scatter_t_body(tile, blockIdx.x, blockIdx.y * 64, slab, splitk, Cout, NK, Cin, taps, Cin_real, dw, accumulate);
}
// The same for ROWS output channels per workgroup, taps <= 16 (every layer of the step): all ROWS * 4 slab loads of a thread
// are in flight before the one barrier -- the single-row form moves 4 KiB in and 4 KiB out per workgroup behind two
// barriers and ran the discriminator's 67 MB layers at 2 TB/s (67 us on the tail of D's backward).
template <int ROWS>
__global__ __launch_bounds__(256) void wgrad_scatter_t_rows_kernel(const float* __restrict__ slab, int splitk, int Cout, int NK,
                                                                   int Cin, int taps, int Cin_real, int Cout_real,
                                                                   float* __restrict__ dw, int accumulate) {
  __shared__ float tile[ROWS][16][65];
  const int co0 = blockIdx.x * ROWS, ci0 = blockIdx.y * 64;
  const size_t zs = (size_t)Cout * NK;
  float v[ROWS][4];
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = threadIdx.x + 256 * j, t = e >> 6, c = e & 63;
      v[r][j] = 0.f;
      if (co0 + r < Cout_real && t < taps && ci0 + c < Cin) {
        const float* q = slab + (size_t)(co0 + r) * NK + (size_t)t * Cin + ci0 + c;
        float s = 0.f;
        for (int z = 0; z < splitk; ++z) s += q[(size_t)z * zs];
        v[r][j] = s;
      }
    }
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = threadIdx.x + 256 * j;
      tile[r][e >> 6][e & 63] = v[r][j];
    }
  __syncthreads();
  const int nc = min(64, Cin_real - ci0);
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    if (co0 + r >= Cout_real) break;
    for (int idx = threadIdx.x; idx < nc * taps; idx += 256) {
      const int cc = idx / taps, t = idx - cc * taps;
      const size_t o = ((size_t)(co0 + r) * Cin_real + ci0 + cc) * taps + t;
      dw[o] = accumulate ? dw[o] + tile[r][t][cc] : tile[r][t][cc];
    }
  }
}

// bias gradient: column sums of dY, two deterministic stages
#define DB_ROWS 2048     // partial rows of the bias-gradient column sums (one workgroup each)
__global__ __launch_bounds__(256) void colsum_partial_kernel(int dt, const char* dy, int dyps, long long npix,
                                                             int C, float* partial) {
  // block = pixel range; threads = (channel vector, pixel lane), LDS tree over lanes
  __shared__ float red[256][4];
  const int nv = C >> 2;
  const long long chunk = (npix + gridDim.x - 1) / gridDim.x;
  const long long a = blockIdx.x * chunk, b = min(npix, a + chunk);
  for (int v0 = 0; v0 < nv; v0 += 256) {
    const int nvv = min(256, nv - v0);          // vectors handled this round
    int lanes = 256 / nvv;
    const int cv = threadIdx.x % nvv, pl = threadIdx.x / nvv;
    f32x4_t s = (f32x4_t){0, 0, 0, 0};
    if (pl < lanes) {
#pragma unroll 4
      for (long long m = a + pl; m < b; m += lanes) s += load4(dy, m * dyps + (v0 + cv) * 4, dt);
    }
    for (int q = 0; q < 4; ++q) red[threadIdx.x][q] = s[q];
    __syncthreads();
    if (pl == 0 && cv < nvv) {
      for (int l = 1; l < lanes; ++l)
        for (int q = 0; q < 4; ++q) s[q] += red[l * nvv + cv][q];
      *(f32x4_t*)(partial + (size_t)blockIdx.x * C + (v0 + cv) * 4) = s;
    }
    __syncthreads();
  }
}
__device__ __forceinline__ void colsum_final_body(double (*red)[32], int vb, const float* partial, int rows, int C,
                                                  int C_real, float* db, int accumulate) {
  // 32 channels x 8 row lanes per block; lanes combined in fixed order
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5, c = vb * 32 + cl;
  double s = 0;
  if (c < C_real)
    for (int r = rl; r < rows; r += 8) s += partial[(size_t)r * C + c];
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < C_real) {
    for (int l = 1; l < 8; ++l) s += red[l][cl];
    db[c] = accumulate ? db[c] + (float)s : (float)s;
  }
  __syncthreads();
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* partial, int rows, int C, int C_real,
                                                           float* db, int accumulate) {
  __shared__ double red[8][32];
  colsum_final_body(red, blockIdx.x, partial, rows, C, C_real, db, accumulate);
}

// One launch for the slab reductions (+ in-kernel bias partials) of up to WFINISH_MAX weight-gradient calls whose
// main kernels ran with defer_finish: blockIdx.y = layer, blockIdx.x strides over that layer's virtual blocks.
#define WFINISH_MAX 24
struct WFinishItem {
  const float* slab; float* dw; float* db; const float* part;
  int splitk, Cout, NK, Cin, KH, KW, Cout_real, Cin_real, accumulate, part_rows;
};
struct WFinishTable { WFinishItem it[WFINISH_MAX]; int first[WFINISH_MAX + 1]; };   // first[i]: first block of item i
__global__ __launch_bounds__(256) void wgrad_finish_multi_kernel(const WFinishTable T, int n) {
  __shared__ double smem[(16 * 65 * sizeof(float) + sizeof(double) - 1) / sizeof(double)];
  int li = 0;
  while (li + 1 < n && (int)blockIdx.x >= T.first[li + 1]) ++li;        // (uniform per block)
  const WFinishItem& t = T.it[li];
  const int vb0 = blockIdx.x - T.first[li], nvb = T.first[li + 1] - T.first[li];
  if (t.splitk <= 8) {          // big layers: bandwidth-bound transposing copy (wgrad_scatter_t_kernel)
    const int nchunk = (t.Cin_real + 63) / 64;
    for (int vb = vb0; vb < t.Cout_real * nchunk; vb += nvb)
      wgrad_scatter_t_body((float (*)[65])smem, vb % t.Cout_real, (vb / t.Cout_real) * 64, t.slab, t.splitk, t.Cout,
                           t.NK, t.Cin, t.KH * t.KW, t.Cin_real, t.dw, t.accumulate);
  } else {                      // few outputs, many splits (wgrad_scatter_kernel)
    wgrad_scatter_body((float (*)[33])smem, vb0, nvb, t.slab, t.splitk, t.Cout, t.NK, t.Cin, t.KH, t.KW,
                       t.Cout_real, t.Cin_real, t.dw, t.accumulate);
  }
  if (t.db)
    for (int vb = vb0; vb < (t.Cout_real + 31) / 32; vb += nvb)
      colsum_final_body((double (*)[32])smem, vb, t.part, t.part_rows, t.Cout, t.Cout_real, t.db, t.accumulate);
}

// ---------------------------------------------------------------------------------------------
// bf16 path: the staged tiles keep the natural [pixel][channel] layout (16-byte global loads
// written straight to LDS, no register transposes, every thread loads both operands) and the
// K-contiguous MFMA fragments are produced by the LDS itself with ds_read_b64_tr_b16 (hardware
// transposed read: a 16-lane group reads a 4-pixel x 16-channel block and each lane receives
// one channel's 4 pixels).  16-byte chunks of a pixel row are XOR-swizzled per row so that the
// transposed reads of a 32-lane half (8 pixel rows x 2 chunks) hit 16 distinct slots.

// ---------------------------------------------------------------------------------------------
// wgrad_glds_kernel: the [pixel][channel] images are filled by LDS-DMA (global_load_lds_dwordx4): no staging
// registers, no ds_write pass (a ds_write_b128 costs 13 LDS cycles per wave-instruction, which made the
// register-staged round-1 kernel store-bound).  An LDS-DMA wave-instruction writes 64 consecutive
// 16-byte slots, so the chunk swizzle of img_off is applied on the source side: the lane that
// owns slot s of pixel row r fetches chunk s ^ f(r).  One LDS buffer, two barriers per 64-pixel
// step; 3-4 workgroups per CU overlap each other's loads and MFMAs.
__device__ __attribute__((aligned(16))) char w_zero_page[16];

//
// wgrad_glds_row_kernel: the same kernel for geometries whose 64-pixel K steps are aligned with
// output rows (Wo % 64 == 0, or 64 % Wo == 0 with Ho*Wo % 64 == 0 -- every power-of-two feature
// map).  The step's (b, oy, ox) origin is then wave-uniform (scalar unit) and a lane's pixel is
// origin + a per-lane constant, which cuts the address arithmetic from ~100 to ~15 vector
// instructions per 16-byte piece; the general kernel below is vector-ALU bound on exactly that.
// DIRECT (16-tap layers without split-K: the discriminator's 512 / 1024-channel layers): the tile's 128 columns are ONE
// block of 8 input channels x all 16 taps instead of 128 consecutive (tap, channel) indices -- the LDS-DMA source address
// is per lane and per 16-byte chunk anyway, so the order of the chunks is free -- arranged so that the four column
// fragments of a wave hold four CONSECUTIVE taps of the same channels: a lane then owns 16 contiguous bytes of the
// reference-layout gradient [Cout][Cin][4][4] and the kernel writes dW itself.  No fp32 slab (64 MB written and read
// back for the 1024 -> 1024 layer) and no transposing scatter launch behind it (47 + 28 us per step on the chain).
template <int BP, int BQ, int WP, int WQ, bool REFLECT, bool UPS, bool DIRECT = false>
__global__ __launch_bounds__(64 * WP * WQ, WP * WQ == 8 ? 2 : (((BP == 128 && BQ == 128) || (BP == 256 && BQ == 64)) ? 3 : 4)) void wgrad_glds_row_kernel(const WParams p) {
  static_assert(!DIRECT || (BP == 128 && WP == 2), "DIRECT: 16 chunks of 8 channels, four column fragments per wave");
  constexpr int PS = 64;
  constexpr int NW = WP * WQ;                         // waves: every wave issues 1/NW of a step's LDS-DMA instructions
  constexpr int CP = BP / 8, CQ = BQ / 8;
  constexpr int NXI = CP / NW, NYI = (CQ + NW - 1) / NW;
  constexpr int WTP = BP / WP, WTQ = BQ / WQ, FP = WTP / 16, FQ = WTQ / 16;
  constexpr int IMG_X = 64 * BP * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wid / WQ, wq = wid % WQ;
  const int ptile = blockIdx.x / p.qtiles, qtile = blockIdx.x - ptile * p.qtiles;
  const int p0 = ptile * BP, q0 = qtile * BQ;
  const int ks = blockIdx.z;
  const int s_begin = ks * p.steps_per_split;
  const int s_end = min(p.nsteps, s_begin + p.steps_per_split);
  const int Hv = UPS ? 2 * p.Hin : p.Hin, Wv = UPS ? 2 * p.Win : p.Win;
  const int rows_per_step = p.Wo >= 64 ? 1 : 64 / p.Wo;

  // per instruction: lane constants (tap shift + pixel-in-step shift), source base
  const char* xsrc[NXI]; int xps[NXI], cu[NXI], cw[NXI];
#pragma unroll
  for (int j = 0; j < NXI; ++j) {
    const int e = (j * NW + wid) * 64 + lane, row = e / CP, slot = e % CP;
    const int chunk = ((img_off<CP>(row, slot) >> 4) % CP);
    const int col = p0 + chunk * 8;
    const int ry = p.Wo >= 64 ? 0 : row / p.Wo, rx = p.Wo >= 64 ? row : row - ry * p.Wo;
    xsrc[j] = w_zero_page; xps[j] = 0; cu[j] = 0; cw[j] = 0;      // columns past NK read zeros
    if (col < p.NK) {
      int tap = col / p.Cin, ci = col - tap * p.Cin;
      if constexpr (DIRECT) {
        // chunk = 8 wp + 2 i + h (wave half wp, fragment i, lane-row pair h) holds tap 8 wp + 4 h + i of channels 8 ptile ..
        tap = 8 * (chunk >> 3) + 4 * (chunk & 1) + ((chunk >> 1) & 3);
        ci = ptile * 8;
      }
      const int ky = tap / p.KW, kx = tap - ky * p.KW;
      cu[j] = ry * p.S + ky - p.pt; cw[j] = rx * p.S + kx - p.pl;
      xsrc[j] = (ci < p.c0) ? p.in0 + (size_t)ci * 2 : p.in1 + (size_t)(ci - p.c0) * 2;
      xps[j] = ((ci < p.c0) ? p.ps0 : p.ps1) * 2;
    }
  }
  const char* ysrc[NYI]; unsigned ystep[NYI];
#pragma unroll
  for (int j = 0; j < NYI; ++j) {
    const int e = (j * NW + wid) * 64 + lane, row = e / CQ, slot = e % CQ;
    const int chunk = ((img_off<CQ>(row, slot) >> 4) % CQ);
    const bool yv = q0 + chunk * 8 < p.Cout;
    ysrc[j] = yv ? p.dy + ((size_t)(s_begin * PS + row) * p.dyps + q0 + chunk * 8) * 2 : w_zero_page;
    ystep[j] = yv ? (unsigned)PS * p.dyps * 2 : 0u;
  }
  // wave-uniform origin of the current step
  int m0 = s_begin * PS;
  int sb = m0 / (p.Ho * p.Wo);
  int rem = m0 - sb * p.Ho * p.Wo;
  int oy0 = rem / p.Wo, ox0 = rem - oy0 * p.Wo;
  sb = __builtin_amdgcn_readfirstlane(sb); oy0 = __builtin_amdgcn_readfirstlane(oy0);
  ox0 = __builtin_amdgcn_readfirstlane(ox0);

  auto issue = [&]() {
    const int su = oy0 * p.S, sw = ox0 * p.S, sbase = sb * p.Hin * p.Win;
#pragma unroll
    for (int j = 0; j < NXI; ++j) {
      int u = su + cu[j], w = sw + cw[j];
      bool ok = true;
      if (REFLECT) {                                   // branch-free mirror, |pad| < n
        u = u < 0 ? -u : u; u = min(u, 2 * (Hv - 1) - u);
        w = w < 0 ? -w : w; w = min(w, 2 * (Wv - 1) - w);
      } else ok = ((unsigned)u < (unsigned)Hv) & ((unsigned)w < (unsigned)Wv);
      if (UPS) { u >>= 1; w >>= 1; }
      const unsigned pix = (unsigned)sbase + __umul24(u, p.Win) + (unsigned)w;   // < 2^24 (host check)
      const char* g = xsrc[j] + __umul24(pix, xps[j]);
      if (!REFLECT) g = ok ? g : w_zero_page;
      __builtin_amdgcn_global_load_lds((wgptr_t)g, (wlptr_t)(smem + (j * NW + wid) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NYI; ++j) {
      if (CQ % NW == 0 || j * NW + wid < CQ) {
        __builtin_amdgcn_global_load_lds((wgptr_t)ysrc[j], (wlptr_t)(smem + IMG_X + (j * NW + wid) * 1024), 16, 0, 0);
        ysrc[j] += ystep[j];
      }
    }
    ox0 += PS;
    if (ox0 >= p.Wo) { ox0 = 0; oy0 += rows_per_step; if (oy0 >= p.Ho) { oy0 = 0; ++sb; } }
  };

  f32x4_t acc[FP][FQ];
#pragma unroll
  for (int i = 0; i < FP; ++i)
#pragma unroll
    for (int j = 0; j < FQ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int g = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;

  for (int s = s_begin; s < s_end; ++s) {
    issue();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      const int rlo = kc * 32 + 8 * g + tq, rhi = rlo + 4;
      bf16x8_t pf[FP], qf[FQ];
#pragma unroll
      for (int i = 0; i < FP; ++i) {
        const int ch = (wp * WTP + i * 16) / 8 + (tp >> 1);
        pf[i] = tr_frag(smem, img_off<CP>(rlo, ch) + (tp & 1) * 8, img_off<CP>(rhi, ch) + (tp & 1) * 8);
      }
#pragma unroll
      for (int j = 0; j < FQ; ++j) {
        const int ch = (wq * WTQ + j * 16) / 8 + (tp >> 1);
        qf[j] = tr_frag(smem + IMG_X, img_off<CQ>(rlo, ch) + (tp & 1) * 8, img_off<CQ>(rhi, ch) + (tp & 1) * 8);
      }
#pragma unroll
      for (int i = 0; i < FP; ++i)
#pragma unroll
        for (int j = 0; j < FQ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[i], qf[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  const int r16 = lane & 15;
  if constexpr (DIRECT) {
    // lane (r16, g): output channel co, input channels 8 ptile + 4 (g & 1) + r, taps 8 wp + 4 (g >> 1) + i (i = fragment)
#pragma unroll
    for (int j = 0; j < FQ; ++j) {
      const int co = q0 + wq * WTQ + j * 16 + r16;
      if (co >= p.Cout) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ci = ptile * 8 + (g & 1) * 4 + r;
        float* o = p.dwd + ((size_t)co * p.Cin + ci) * 16 + 8 * wp + 4 * (g >> 1);
        f32x4_t v = (f32x4_t){acc[0][j][r], acc[1][j][r], acc[2][j][r], acc[3][j][r]};
        if (p.dw_acc) v += *(const f32x4_t*)o;
        *(f32x4_t*)o = v;
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < FQ; ++j) {
    const int co = q0 + wq * WTQ + j * 16 + r16;
    if (co >= p.Cout) continue;
#pragma unroll
    for (int i = 0; i < FP; ++i) {
      const int cc = p0 + wp * WTP + i * 16 + g * 4;
      if (cc < p.NK) *(f32x4_t*)(p.slab + ((size_t)ks * p.Cout + co) * p.NK + cc) = acc[i][j];
    }
  }
}

template <int BP, int BQ, int WP, int WQ>
__global__ __launch_bounds__(256, 3) void wgrad_glds_kernel(const WParams p) {
  constexpr int PS = 64;
  constexpr int CP = BP / 8, CQ = BQ / 8;
  constexpr int NXI = CP / 4, NYI = (CQ + 3) / 4;    // LDS-DMA instructions per wave per step
  constexpr int WTP = BP / WP, WTQ = BQ / WQ, FP = WTP / 16, FQ = WTQ / 16;
  constexpr int IMG_X = 64 * BP * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wp = wid / WQ, wq = wid % WQ;
  const int ptile = blockIdx.x / p.qtiles, qtile = blockIdx.x - ptile * p.qtiles;
  const int p0 = ptile * BP, q0 = qtile * BQ;
  const int ks = blockIdx.z;
  const int s_begin = ks * p.steps_per_split;
  const int s_end = min(p.nsteps, s_begin + p.steps_per_split);
  const int HoWo = p.Ho * p.Wo;
  const int Hv = p.ups ? 2 * p.Hin : p.Hin, Wv = p.ups ? 2 * p.Win : p.Win;

  // X operand: per instruction, this lane's pixel row (within the step) and (tap, ci) chunk
  const char* xsrc[NXI]; int xps[NXI], dyo[NXI], dxo[NXI], sb[NXI], soy[NXI], sox[NXI], xm[NXI];
#pragma unroll
  for (int j = 0; j < NXI; ++j) {
    const int e = (j * 4 + wid) * 64 + lane, row = e / CP, slot = e % CP;
    const int chunk = ((img_off<CP>(row, slot) >> 4) % CP);
    const int col = p0 + chunk * 8;
    xsrc[j] = nullptr; xps[j] = 0; dyo[j] = 0; dxo[j] = 0;
    if (col < p.NK) {
      const int tap = col / p.Cin, ci = col - tap * p.Cin;
      const int ky = tap / p.KW, kx = tap - ky * p.KW;
      dyo[j] = ky - p.pt; dxo[j] = kx - p.pl;
      xsrc[j] = (ci < p.c0) ? p.in0 + (size_t)ci * 2 : p.in1 + (size_t)(ci - p.c0) * 2;
      xps[j] = ((ci < p.c0) ? p.ps0 : p.ps1) * 2;
    }
    const int m = s_begin * PS + row;
    xm[j] = m;
    sb[j] = m / HoWo;
    const int r = m - sb[j] * HoWo;
    soy[j] = r / p.Wo; sox[j] = r - soy[j] * p.Wo;
  }
  // dY operand
  const char* ysrc[NYI]; int ym[NYI];
#pragma unroll
  for (int j = 0; j < NYI; ++j) {
    const int e = (j * 4 + wid) * 64 + lane, row = e / CQ, slot = e % CQ;
    const int chunk = ((img_off<CQ>(row, slot) >> 4) % CQ);
    ym[j] = s_begin * PS + row;
    ysrc[j] = (q0 + chunk * 8 < p.Cout) ? p.dy + ((size_t)ym[j] * p.dyps + q0 + chunk * 8) * 2 : nullptr;
  }

  auto issue = [&]() {
#pragma unroll
    for (int j = 0; j < NXI; ++j) {
      const char* g = w_zero_page;
      if (xsrc[j] && xm[j] < p.M) {
        int u = soy[j] * p.S + dyo[j], w = sox[j] * p.S + dxo[j];
        bool ok = true;
        if (p.border == CSMRI_BORDER_REFLECT) { u = reflect_idx(u, Hv); w = reflect_idx(w, Wv); }
        else ok = (unsigned)u < (unsigned)Hv && (unsigned)w < (unsigned)Wv;
        if (p.ups) { u >>= 1; w >>= 1; }
        if (ok) g = xsrc[j] + (((size_t)sb[j] * p.Hin + u) * p.Win + w) * (size_t)xps[j];
      }
      __builtin_amdgcn_global_load_lds((wgptr_t)g, (wlptr_t)(smem + (j * 4 + wid) * 1024), 16, 0, 0);
      xm[j] += PS; sox[j] += PS;
      while (sox[j] >= p.Wo) { sox[j] -= p.Wo; if (++soy[j] == p.Ho) { soy[j] = 0; ++sb[j]; } }
    }
#pragma unroll
    for (int j = 0; j < NYI; ++j) {
      if (CQ % 4 == 0 || j * 4 + wid < CQ) {
        const char* g = (ysrc[j] && ym[j] < p.M) ? ysrc[j] : w_zero_page;
        __builtin_amdgcn_global_load_lds((wgptr_t)g, (wlptr_t)(smem + IMG_X + (j * 4 + wid) * 1024), 16, 0, 0);
        ym[j] += PS;
        if (ysrc[j]) ysrc[j] += (size_t)PS * p.dyps * 2;
      }
    }
  };

  f32x4_t acc[FP][FQ];
#pragma unroll
  for (int i = 0; i < FP; ++i)
#pragma unroll
    for (int j = 0; j < FQ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int g = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;

  for (int s = s_begin; s < s_end; ++s) {
    issue();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      const int rlo = kc * 32 + 8 * g + tq, rhi = rlo + 4;
      bf16x8_t pf[FP], qf[FQ];
#pragma unroll
      for (int i = 0; i < FP; ++i) {
        const int ch = (wp * WTP + i * 16) / 8 + (tp >> 1);
        pf[i] = tr_frag(smem, img_off<CP>(rlo, ch) + (tp & 1) * 8, img_off<CP>(rhi, ch) + (tp & 1) * 8);
      }
#pragma unroll
      for (int j = 0; j < FQ; ++j) {
        const int ch = (wq * WTQ + j * 16) / 8 + (tp >> 1);
        qf[j] = tr_frag(smem + IMG_X, img_off<CQ>(rlo, ch) + (tp & 1) * 8, img_off<CQ>(rhi, ch) + (tp & 1) * 8);
      }
#pragma unroll
      for (int i = 0; i < FP; ++i)
#pragma unroll
        for (int j = 0; j < FQ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[i], qf[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  const int r16 = lane & 15;
#pragma unroll
  for (int j = 0; j < FQ; ++j) {
    const int co = q0 + wq * WTQ + j * 16 + r16;
    if (co >= p.Cout) continue;
#pragma unroll
    for (int i = 0; i < FP; ++i) {
      const int cc = p0 + wp * WTP + i * 16 + g * 4;
      if (cc < p.NK) *(f32x4_t*)(p.slab + ((size_t)ks * p.Cout + co) * p.NK + cc) = acc[i][j];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// wpatch: weight gradient of the stride-1 layers with few channels on large maps (RecNet, the 256^2 /
// 128^2 U-Net levels: Cin in {32, 64}, Cout <= 64) from an LDS PATCH.  The kernels above gather X once
// per filter tap and per 256-row tile of the (tap, channel) axis; on these layers (16 taps, two such
// tiles) that is what the hardware counters showed: 2.8-3x the operand bytes fetched, and at C2's batch
// of 64, where the tensors exceed the Infinity Cache, all of it comes from HBM.  Here a workgroup owns
// 16 x 16 output pixels at a time: it stages the (16+KH-1) x (16+KW-1) input patch (border rule,
// upsampling, concat applied while loading, as in tconv.hip) and the 256 x Cout dY tile ONCE, and every
// tap reads its shifted window of the patch from LDS with ds_read_b64_tr_b16.  Each input pixel is
// fetched 1.27-1.4 times (the halo) and dY once.  Workgroups are persistent over tiles (grid.z-style
// split count = number of workgroups) so the fp32 accumulators leave as ONE slab per workgroup; the
// slabs go through the same deterministic reduce-scatter as before.
// Work split: wave w owns taps w, w+WAVES, ...; all (Cin/16) x (Cout/16) fragment pairs of a tap.
template <int CIN, int COUT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void wpatch_kernel(const WParams p) {
  constexpr int CPR = CIN / 8;                      // 16-byte chunks per patch pixel
  constexpr int CQ = COUT / 8;                      // chunks per dY pixel (COUT = 16 for 8 real channels)
  // CIN = 8 (the 2-channel first layers, padded to 8): a 16-wide fragment of the (tap, channel) axis is a PAIR
  // of taps x 8 channels; the work unit of a wave is such a pair instead of a tap
  constexpr int CF = CIN >= 16 ? CIN / 16 : 1, NF = COUT / 16;
  constexpr int UNITS = CIN >= 16 ? 16 : 8;         // taps, or tap pairs (KH * KW <= 16)
  constexpr int MAXT = UNITS / WAVES;               // units per wave
  constexpr int XROWS = 1024 / (CPR * 16), YROWS = 1024 / (CQ * 16);   // rows per LDS-DMA instruction
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = p.S;                                // 1, or 2 for the discriminator's first layer (CIN = 8)
  const int TPW = 15 * S + p.KW, TPH = 15 * S + p.KH, npix = TPH * TPW;
  const int xinstr = (npix + XROWS - 1) / XROWS;
  const int IMG_X = xinstr * 1024;                  // dY image follows the patch
  const int IMG_BOTH = IMG_X + 256 * COUT * 2;      // one stage = patch + dY tile; two stages (double buffer)
  const int taps = p.KH * p.KW;
  const int tiles_x = (p.Wo + 15) >> 4, tiles_y = (p.Ho + 15) >> 4, ntiles = p.B * tiles_x * tiles_y;
  const int Hv = p.ups ? 2 * p.Hin : p.Hin, Wv = p.ups ? 2 * p.Win : p.Win;
  const int g = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3, r16 = lane & 15;
  const int qoff = blockIdx.y * COUT;               // this workgroup's slice of the output channels (grid.y > 1: Cin = 128)

  // bias gradient = column sums of dY: one more MFMA per dY fragment against a fragment of ones (wave 0)
  f32x4_t bacc[NF];
#pragma unroll
  for (int n = 0; n < NF; ++n) bacc[n] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, (u32x4_t){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});
  f32x4_t acc[MAXT][CF][NF];
#pragma unroll
  for (int a = 0; a < MAXT; ++a)
#pragma unroll
    for (int c = 0; c < CF; ++c)
#pragma unroll
      for (int n = 0; n < NF; ++n) acc[a][c][n] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // lane constants of the staging instructions
  const int xr = lane / CPR, xslot = lane % CPR;
  const int yr = lane / CQ, yslot = lane % CQ;
  const unsigned magic = 0xFFFFFFFFu / (unsigned)TPW + 1u;
  // lane constants of the fragment reads: K chunk kc = tile rows 2kc, 2kc+1; this lane's pixel k = 8g + tq (+4)
  const int klo = 8 * g + tq;
  const int prow = S * ((klo >> 4) * TPW + (klo & 15));   // patch row of pixel k for tap (0,0), tile row 0

  // this wave's taps (tap pairs for CIN = 8, where the tap is a lane property) as patch-row offsets ty * TPW + tx:
  // decoded ONCE -- the integer division by the filter width sat inside the K loop (PMC: 9.5 VALU per MFMA)
  int toff[MAXT];
#pragma unroll
  for (int a = 0; a < MAXT; ++a) {
    const int unit = wid + a * WAVES;
    int tap = CIN >= 16 ? unit : 2 * unit + (tp >> 1);
    if (CIN < 16 && tap >= taps) tap = 2 * unit;      // odd tap count: the last pair repeats its first tap
    const int ty = tap / p.KW, tx = tap - ty * p.KW;
    toff[a] = ty * TPW + tx;
  }

  auto stage = [&](int t, char* buf) {
    char* ximg = buf; char* yimg = buf + IMG_X;
    const int b = t / (tiles_x * tiles_y);
    const int rem = t - b * tiles_x * tiles_y;
    const int y0 = (rem / tiles_x) * 16, x0 = (rem % tiles_x) * 16;
    for (int i = wid; i < xinstr; i += WAVES) {
      const int P = i * XROWS + xr;
      const int py = (int)__umulhi((unsigned)P, magic), px = P - py * TPW;
      int u = y0 * S - p.pt + py, w = x0 * S - p.pl + px;
      if (p.border == CSMRI_BORDER_REFLECT) {
        u = u < 0 ? -u : u; u = min(u, 2 * (Hv - 1) - u);
        w = w < 0 ? -w : w; w = min(w, 2 * (Wv - 1) - w);
      }
      const bool ok = (P < npix) & ((unsigned)u < (unsigned)Hv) & ((unsigned)w < (unsigned)Wv);
      if (p.ups) { u >>= 1; w >>= 1; }
      const size_t pix = ((size_t)b * p.Hin + u) * p.Win + w;
      const int chunk = (img_off<CPR>(P, xslot) >> 4) % CPR;      // source-side swizzle
      const int c = chunk * 8;
      const char* src = c < p.c0 ? p.in0 + (pix * p.ps0 + c) * 2 : p.in1 + (pix * p.ps1 + (c - p.c0)) * 2;
      src = ok ? src : w_zero_page;
      __builtin_amdgcn_global_load_lds((wgptr_t)src, (wlptr_t)(ximg + i * 1024), 16, 0, 0);
    }
    for (int i = wid; i < 256 / YROWS; i += WAVES) {
      const int k = i * YROWS + yr;                 // tile pixel: row k >> 4, column k & 15
      const int oy = y0 + (k >> 4), ox = x0 + (k & 15);
      const int chunk = (img_off<CQ>(k, yslot) >> 4) % CQ;
      const bool ok = oy < p.Ho && ox < p.Wo && qoff + chunk * 8 < p.Cout;
      const char* src = ok ? p.dy + ((((size_t)b * p.Ho + oy) * p.Wo + ox) * p.dyps + qoff + chunk * 8) * 2 : w_zero_page;
      __builtin_amdgcn_global_load_lds((wgptr_t)src, (wlptr_t)(yimg + i * 1024), 16, 0, 0);
    }
  };
  if (p.stages >= 2 && (int)blockIdx.x < ntiles) stage(blockIdx.x, smem);
  int it = 0;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x, ++it) {
    // ---- this tile's images were staged one iteration ahead (LDS-DMA, asynchronous): wait for my pieces, then the
    // barrier publishes all pieces AND retires every wave's reads of the other stage, which the next tile's DMA
    // overwrites from here on while this tile is multiplied -----------------------------------------------------
    if (p.stages < 2) {                       // single stage: every wave is done with the previous tile, then reload
      if (it) __syncthreads();
      stage(t, smem);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    char* ximg = smem + (p.stages < 2 ? 0 : (it & 1) * IMG_BOTH); char* yimg = ximg + IMG_X;
    if (p.stages >= 2 && t + (int)gridDim.x < ntiles) stage(t + gridDim.x, smem + ((it & 1) ^ 1) * IMG_BOTH);
    // ---- 8 K chunks of 32 pixels x my taps -----------------------------------------------------
#pragma unroll 1
    for (int kc = 0; kc < 8; ++kc) {
      bf16x8_t yf[NF];
      const int yrlo = kc * 32 + klo, yrhi = yrlo + 4;
      {
        // chunk n*2 + (tp>>1) = (n*2) ^ (tp>>1): the fragments of a row differ by an XOR of the slot bits only
        const int ylo = img_off<CQ>(yrlo, tp >> 1) + (tp & 1) * 8, yhi = img_off<CQ>(yrhi, tp >> 1) + (tp & 1) * 8;
#pragma unroll
        for (int n = 0; n < NF; ++n) yf[n] = tr_frag(yimg, ylo ^ (n << 5), yhi ^ (n << 5));
      }
      if (wid == 0 && p.nsteps) {                  // p.nsteps != 0: the caller wants the bias gradient
#pragma unroll
        for (int n = 0; n < NF; ++n) bacc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, yf[n], bacc[n], 0, 0, 0);
      }
#pragma unroll
      for (int a = 0; a < MAXT; ++a) {
        const int unit = wid + a * WAVES;
        if constexpr (CIN >= 16) {
          if (unit >= taps) break;                    // wave-uniform
          const int rlo = prow + S * 2 * kc * TPW + toff[a], rhi = rlo + 4 * S;
          const int xlo = img_off<CPR>(rlo, tp >> 1) + (tp & 1) * 8, xhi = img_off<CPR>(rhi, tp >> 1) + (tp & 1) * 8;
#pragma unroll
          for (int c = 0; c < CF; ++c) {
            const bf16x8_t xf = tr_frag(ximg, xlo ^ (c << 5), xhi ^ (c << 5));
#pragma unroll
            for (int n = 0; n < NF; ++n)
              acc[a][c][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, yf[n], acc[a][c][n], 0, 0, 0);
          }
        } else {
          if (2 * unit >= taps) break;                // wave-uniform
          // lanes tp = 0,1 address the first tap of the pair, tp = 2,3 the second (the last pair of an odd
          // tap count repeats the first tap; its half of the result is never written)
          const int rlo = prow + S * 2 * kc * TPW + toff[a], rhi = rlo + 4 * S;
          const bf16x8_t xf = tr_frag(ximg, rlo * 16 + (tp & 1) * 8, rhi * 16 + (tp & 1) * 8);
#pragma unroll
          for (int n = 0; n < NF; ++n)
            acc[a][0][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, yf[n], acc[a][0][n], 0, 0, 0);
        }
      }
    }
  }
  // ---- one slab per workgroup: [Cout][NK], NK index = tap * Cin + ci ------------------------------
  if (wid == 0 && p.nsteps && g == 0) {            // row 0 of the ones product: sum over pixels per output channel
    float* part = p.slab + (size_t)gridDim.x * p.Cout * p.NK + (size_t)blockIdx.x * p.Cout;
#pragma unroll
    for (int n = 0; n < NF; ++n)
      if (qoff + n * 16 + r16 < p.Cout) part[qoff + n * 16 + r16] = bacc[n][0];
  }
#pragma unroll
  for (int a = 0; a < MAXT; ++a) {
    const int unit = wid + a * WAVES;
    if ((CIN >= 16 ? unit : 2 * unit) >= taps) break;
#pragma unroll
    for (int n = 0; n < NF; ++n) {
      const int co = qoff + n * 16 + r16;
      if (co >= p.Cout) continue;
      if constexpr (CIN >= 16) {
#pragma unroll
        for (int c = 0; c < CF; ++c) {
          const int cc = unit * CIN + c * 16 + g * 4;
          *(f32x4_t*)(p.slab + ((size_t)blockIdx.x * p.Cout + co) * p.NK + cc) = acc[a][c][n];
        }
      } else {
        const int tap = 2 * unit + (g >> 1);            // fragment rows 0-7: first tap, 8-15: second
        if (tap < taps)
          *(f32x4_t*)(p.slab + ((size_t)blockIdx.x * p.Cout + co) * p.NK + tap * 8 + (g & 1) * 4) = acc[a][0][n];
      }
    }
  }
}

static int wpatch_cout(const csmri_wgrad_desc* d) { return d->Cout <= 16 ? 16 : d->Cout; }
static bool wpatch_eligible(const csmri_wgrad_desc* d) {
  // stride 2: the discriminator's first layer (one real input channel padded to 8; reference models/discriminators.py:137-150),
  // which the generic row kernel ran at 8.6 TFLOP/s (62 us for 50 MB of operands)
  if (d->dtype != CSMRI_BF16 || !(d->stride == 1 || (d->stride == 2 && d->Cin == 8 && !d->upsample && !d->in1))) return false;
  if (!(d->Cin == 8 || d->Cin == 32 || d->Cin == 64 || (d->Cin == 128 && d->Cout == 64))) return false;
  if (!(d->Cout == 8 || d->Cout == 16 || d->Cout == 32 || d->Cout == 64)) return false;
  if (d->KH * d->KW > 16 || d->KH < 1 || d->KW < 1) return false;
  if (d->in1 && (d->c0 % 8)) return false;
  const long long tiles = (long long)d->B * ((d->Ho + 15) / 16) * ((d->Wo + 15) / 16);
  return tiles >= 512;                                           // large maps only
}
#ifndef WPATCH_GROUPS
#define WPATCH_GROUPS 512
#endif
static int wpatch_groups(const csmri_wgrad_desc* d) {
  const long long tiles = (long long)d->B * ((d->Ho + 15) / 16) * ((d->Wo + 15) / 16);
  long long g = WPATCH_GROUPS;                                   // persistent workgroups
  if (g > tiles / 4) g = tiles / 4;                              // at least 4 tiles per workgroup
  // every workgroup leaves a slab of Cout x NK floats: keep the slab traffic below the operands' own bytes
  const long long slab = (long long)d->Cout * d->KH * d->KW * d->Cin * 4;
  const long long operands = ((long long)d->B * d->Hin * d->Win * d->Cin + (long long)d->B * d->Ho * d->Wo * d->Cout) * 2;
  if (g * slab > operands) g = operands / slab;
  if (g < 128) g = 128;
  if (g > tiles) g = tiles;
  return (int)(g < 1 ? 1 : g);
}
template <int CIN, int COUT, int WAVES>
static int launch_wpatch(const WParams& p0, hipStream_t st, int qtiles = 1) {
  WParams p = p0;
  const int TPW = 15 * p.S + p.KW, TPH = 15 * p.S + p.KH;
  const int xrows = 1024 / (CIN / 8 * 16);
  const int one = ((TPH * TPW + xrows - 1) / xrows) * 1024 + 256 * COUT * 2;
  // two stages where two workgroups per CU still fit (<= 80 KiB each); the big-patch variants stay single-staged
  p.stages = 2 * one <= 80 * 1024 ? 2 : 1;
  const int lds = p.stages >= 2 ? 2 * one : one;
  CSMRI_SET_MAX_LDS((wpatch_kernel<CIN, COUT, WAVES>), lds);
  hipLaunchKernelGGL((wpatch_kernel<CIN, COUT, WAVES>), dim3(p.splitk, qtiles), dim3(WAVES * 64), lds, st, p);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}
static int wpatch_launch(const WParams& p, const csmri_wgrad_desc* d, hipStream_t st) {
  const int co = wpatch_cout(d);
  if (d->Cin == 8) {
    if (co == 16) return launch_wpatch<8, 16, 4>(p, st);
    if (co == 32) return launch_wpatch<8, 32, 4>(p, st);
    return launch_wpatch<8, 64, 4>(p, st);
  }
  if (d->Cin == 32) {
    if (co == 16) return launch_wpatch<32, 16, 4>(p, st);
    if (co == 32) return launch_wpatch<32, 32, 4>(p, st);
    return launch_wpatch<32, 64, 4>(p, st);
  }
  if (d->Cin == 128) return launch_wpatch<128, 32, 8>(p, st, 2);     // two workgroups per tile, 32 output channels each
  if (co == 16) return launch_wpatch<64, 16, 4>(p, st);
  if (co == 32) return launch_wpatch<64, 32, 4>(p, st);
  return launch_wpatch<64, 64, 8>(p, st);
}

// ---------------------------------------------------------------------------------------------------------------
// Thin layer: the weight gradient of the discriminator's final conv (1024 -> 1 channel, 4 x 4 on an 8 x 8 map:
// reference models/discriminators.py:160-172) as a streaming reduction on the vector unit.  The MFMA kernels above
// spend a 16-wide tile side on that one channel: 63 us for 3 MB of operands; this one takes 21 us.  A block owns
// (pixel range z, filter tap, block of <= 256 input channels); a thread (pixel lane, 8 channels) reads 16 bytes of x
// at the tap-shifted position + the dY scalars per pixel; pixel lanes combine by wave shuffles, the four waves
// through LDS (fixed order).  It leaves slabs [z][Cout][NK] and one bias-gradient partial row per z behind, i.e. it
// plugs into the slab reduction (wgrad_scatter / csmri_wgrad_finish_multi) like wpatch.  (The same form on the
// large-map thin layers -- U-Net head 76 vs 49 us, discriminator first layer 208 vs 60 us -- lost and is not kept.)
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wthin_decomp(const WParams& p, int m, int& b, int& oy, int& ox) {
  const int HoWo = p.Ho * p.Wo;
  b = m / HoWo; const int r = m - b * HoWo; oy = r / p.Wo; ox = r - oy * p.Wo;
}
__device__ __forceinline__ void wthin_ld8(const char* q, int dt, float (&g)[8]) {
  if (dt == CSMRI_BF16) {
    const u32x4_t u = *(const u32x4_t*)q;
#pragma unroll
    for (int i = 0; i < 4; ++i) { g[2 * i] = __uint_as_float(u[i] << 16); g[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u); }
  } else {
    const f32x4_t a = *(const f32x4_t*)q, b = *(const f32x4_t*)(q + 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) { g[i] = a[i]; g[4 + i] = b[i]; }
  }
}

__global__ __launch_bounds__(256) void wthin_out_kernel(const WParams p, int dt, int want_db, int cout_real, int KL,
                                                        int cblocks) {
  __shared__ float red[132][17];
  const int es = dt == CSMRI_BF16 ? 2 : 4;
  const int kl = threadIdx.x % KL, pl = threadIdx.x / KL, lanes = 256 / KL;
  const int z = blockIdx.x, Z = gridDim.x;
  const int tap = blockIdx.y / cblocks, cb = blockIdx.y - tap * cblocks;
  const int ky = tap / p.KW, kx = tap - ky * p.KW;
  const int c = (cb * KL + kl) * 8;                           // this thread's 8 input channels
  const bool cv = c < p.Cin;
  const int per = (p.M + Z - 1) / Z, m_begin = z * per, m_end = min(p.M, m_begin + per);
  float acc[2][8], accb[2] = {0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 8; ++j) { acc[0][j] = 0.f; acc[1][j] = 0.f; }
  for (int m = m_begin + pl; m < m_end; m += lanes) {
    int b, oy, ox;
    wthin_decomp(p, m, b, oy, ox);
    const char* gq = p.dy + (size_t)m * p.dyps * es;
    float g0, g1 = 0.f;
    if (dt == CSMRI_BF16) {
      const unsigned gg = *(const unsigned*)gq;                // channels 0, 1 (padded to 8: always readable)
      g0 = __uint_as_float(gg << 16); g1 = __uint_as_float(gg & 0xffff0000u);
    } else { g0 = ((const float*)gq)[0]; g1 = ((const float*)gq)[1]; }
    accb[0] += g0; accb[1] += g1;
    int u = oy * p.S - p.pt + ky, v = ox * p.S - p.pl + kx;
    bool ok = cv;
    if (p.border == CSMRI_BORDER_REFLECT) { u = reflect_idx(u, p.Hin); v = reflect_idx(v, p.Win); }
    else ok = ok && (unsigned)u < (unsigned)p.Hin && (unsigned)v < (unsigned)p.Win;
    if (ok) {
      float x[8];
      wthin_ld8(p.in0 + (((size_t)b * p.Hin + u) * p.Win + v) * p.ps0 * es + (size_t)c * es, dt, x);
#pragma unroll
      for (int j = 0; j < 8; ++j) { acc[0][j] += x[j] * g0; acc[1][j] += x[j] * g1; }
    }
  }
  // pixel lanes of a wave by shuffles (lane bits above the channel-lane bits), the four waves through LDS, fixed order
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int o = KL; o < 64; o <<= 1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { acc[0][j] += __shfl_xor(acc[0][j], o); acc[1][j] += __shfl_xor(acc[1][j], o); }
    accb[0] += __shfl_xor(accb[0], o); accb[1] += __shfl_xor(accb[1], o);
  }
  if (lane < KL) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[wave * 32 + lane][j] = acc[0][j]; red[wave * 32 + lane][8 + j] = acc[1][j]; }
    red[wave * 32 + lane][16] = 0.f;
  }
  if (lane == 0) { red[128 + wave][0] = accb[0]; red[128 + wave][1] = accb[1]; }
  __syncthreads();
  if (threadIdx.x < KL && (cb * KL + (int)threadIdx.x) * 8 < p.Cin) {
    const int k = threadIdx.x, cc = (cb * KL + k) * 8;
    for (int co = 0; co < cout_real; ++co) {
      float s[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        s[j] = red[k][co * 8 + j] + red[32 + k][co * 8 + j] + red[64 + k][co * 8 + j] + red[96 + k][co * 8 + j];
      float* q = p.slab + ((size_t)z * p.Cout + co) * p.NK + (size_t)tap * p.Cin + cc;
      *(f32x4_t*)q = (f32x4_t){s[0], s[1], s[2], s[3]};
      *(f32x4_t*)(q + 4) = (f32x4_t){s[4], s[5], s[6], s[7]};
    }
  }
  if (want_db && blockIdx.y == 0 && (int)threadIdx.x < cout_real)      // bias-gradient partial row of this pixel range
    p.slab[(size_t)Z * p.Cout * p.NK + (size_t)z * p.Cout + threadIdx.x] =
        red[128][threadIdx.x] + red[129][threadIdx.x] + red[130][threadIdx.x] + red[131][threadIdx.x];
}

static bool wthin_out_eligible(const csmri_wgrad_desc* d) {
  if (d->Cout != 8 || d->Cout_real > 2 || d->in1 || d->upsample || d->Cin % 8) return false;
  return (long long)d->B * d->Ho * d->Wo <= 4096 && d->Cin >= 256;        // deep K on few positions
}
static int wthin_out_kl(const csmri_wgrad_desc* d) { int kl = d->Cin / 8; while (kl > 32) kl = (kl + 1) / 2; int p2 = 1; while (p2 < kl) p2 <<= 1; return p2 > 32 ? 32 : p2; }
static int wthin_splits(const csmri_wgrad_desc* d) {
  const long long M = (long long)d->B * d->Ho * d->Wo;
  const int kl = wthin_out_kl(d), cblocks = (d->Cin / 8 + kl - 1) / kl, kb = d->KH * d->KW * cblocks;
  long long z = 1024 / kb; if (z < 1) z = 1;
  const long long maxz = M / (4 * (256 / kl)); if (z > maxz) z = maxz; if (z < 1) z = 1;
  if (z > 2048) z = 2048;                                                    // DB_ROWS partial rows
  return (int)z;
}
static int wthin_launch(const WParams& p, const csmri_wgrad_desc* d, hipStream_t st) {
  const int kl = wthin_out_kl(d), cblocks = (d->Cin / 8 + kl - 1) / kl;
  hipLaunchKernelGGL(wthin_out_kernel, dim3(p.splitk, d->KH * d->KW * cblocks), dim3(256), 0, st, p, d->dtype,
                     d->db ? 1 : 0, d->Cout_real, kl, cblocks);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Thin layer on a LARGE map: the weight gradient of a 1 x 1 convolution with one or two real output channels (U-Net head
// 32 -> 1, reference models/unet.py:241 backward): dW[co][ci] = sum_m dY[m][co] x[m][ci], 42 MB of operands and 34 MFLOP.
// wpatch ran it on MFMA tiles whose N side is that one channel: 52 us.  Here a thread owns one 16-byte chunk (8 channels)
// of a pixel lane and streams its pixels in BATCHES of 8 loads (x chunk + the dY word) ahead of the arithmetic -- the
// one-pixel-per-iteration form of wthin_out_kernel measured 76 us on this layer (a dependent L2 round trip per pixel).
// Pixel lanes combine by wave shuffles, the four waves through LDS in fixed order; slabs and bias partial rows as wpatch.
// ---------------------------------------------------------------------------------------------------------------
template <int CPR>
__global__ __launch_bounds__(256) void whead_kernel(const WParams p, int want_db, int cout_real) {
  constexpr int LANES = 256 / CPR, U = 8;
  __shared__ float red[4][CPR][17];
  __shared__ float redb[4][2];
  const int q = threadIdx.x % CPR, pl = threadIdx.x / CPR;
  const int z = blockIdx.x, Z = gridDim.x;
  const int per = (p.M + Z - 1) / Z, m_begin = z * per, m_end = min(p.M, m_begin + per);
  float acc[2][8], accb[2] = {0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 8; ++j) { acc[0][j] = 0.f; acc[1][j] = 0.f; }
  const char* xb = p.in0 + q * 16;
  for (int m0 = m_begin; m0 < m_end; m0 += LANES * U) {
    u32x4_t xr[U]; unsigned gr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int m = m0 + u * LANES + pl, mm = m < m_end ? m : m_begin;      // clamped: counted with a zero factor below
      xr[u] = *(const u32x4_t*)(xb + (size_t)mm * p.ps0 * 2);
      gr[u] = *(const unsigned*)(p.dy + (size_t)mm * p.dyps * 2);            // channels 0, 1 of the padded dY pixel
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float f = (m0 + u * LANES + pl) < m_end ? 1.f : 0.f;
      const float g0 = __uint_as_float(gr[u] << 16) * f, g1 = __uint_as_float(gr[u] & 0xffff0000u) * f;
      accb[0] += g0; accb[1] += g1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float x0 = __uint_as_float(xr[u][i] << 16), x1 = __uint_as_float(xr[u][i] & 0xffff0000u);
        acc[0][2 * i] += x0 * g0; acc[0][2 * i + 1] += x1 * g0;
        acc[1][2 * i] += x0 * g1; acc[1][2 * i + 1] += x1 * g1;
      }
    }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int o = CPR; o < 64; o <<= 1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { acc[0][j] += __shfl_xor(acc[0][j], o); acc[1][j] += __shfl_xor(acc[1][j], o); }
    accb[0] += __shfl_xor(accb[0], o); accb[1] += __shfl_xor(accb[1], o);
  }
  if (lane < CPR) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[wave][lane][j] = acc[0][j]; red[wave][lane][8 + j] = acc[1][j]; }
  }
  if (lane == 0) { redb[wave][0] = accb[0]; redb[wave][1] = accb[1]; }
  __syncthreads();
  if ((int)threadIdx.x < CPR * 8 * cout_real) {
    const int co = threadIdx.x / (CPR * 8), r = threadIdx.x % (CPR * 8), k = r / 8, j = r % 8;
    p.slab[((size_t)z * p.Cout + co) * p.NK + k * 8 + j] =
        red[0][k][co * 8 + j] + red[1][k][co * 8 + j] + red[2][k][co * 8 + j] + red[3][k][co * 8 + j];
  }
  if (want_db && (int)threadIdx.x < cout_real)
    p.slab[(size_t)Z * p.Cout * p.NK + (size_t)z * p.Cout + threadIdx.x] =
        redb[0][threadIdx.x] + redb[1][threadIdx.x] + redb[2][threadIdx.x] + redb[3][threadIdx.x];
}
static bool whead_eligible(const csmri_wgrad_desc* d) {
  if (d->dtype != CSMRI_BF16 || d->KH != 1 || d->KW != 1 || d->stride != 1 || d->pad_t || d->pad_l) return false;
  if (d->Cout != 8 || d->Cout_real > 2 || d->in1 || d->upsample || !(d->Cin == 32 || d->Cin == 64)) return false;
  if (d->Hin != d->Ho || d->Win != d->Wo) return false;
  return (long long)d->B * d->Ho * d->Wo >= 65536;
}
static int whead_splits(const csmri_wgrad_desc* d) {
  const long long M = (long long)d->B * d->Ho * d->Wo;
  long long z = M / 2048; if (z > 256) z = 256; if (z < 1) z = 1;        // (the slab reduction sums z rows serially per element)
  return (int)z;
}
static int whead_launch(const WParams& p, const csmri_wgrad_desc* d, hipStream_t st) {
  if (d->Cin == 32) hipLaunchKernelGGL(whead_kernel<4>, dim3(p.splitk), dim3(256), 0, st, p, d->db ? 1 : 0, d->Cout_real);
  else hipLaunchKernelGGL(whead_kernel<8>, dim3(p.splitk), dim3(256), 0, st, p, d->db ? 1 : 0, d->Cout_real);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

struct WConfig { int BP, BQ; };
// geometries the row-aligned LDS-DMA kernel takes (64-pixel K steps aligned with output rows)
static bool wgrad_row_aligned(const csmri_wgrad_desc* d) {
  return d->Wo % 64 == 0 || (64 % d->Wo == 0 && ((long long)d->Ho * d->Wo) % 64 == 0);
}
static WConfig pick_wconfig(const csmri_wgrad_desc* d) {
  WConfig c;
  if (d->Cout > 64) { c.BP = 128; c.BQ = 128; }
  else if (d->Cout > 32) {
    // 256 x 64 with one 64 x 64 tile per wave: 1.0 transposed LDS reads per MFMA instead of 1.5
    const bool wide = d->dtype == CSMRI_BF16 && (long long)d->KH * d->KW * d->Cin >= 256 && wgrad_row_aligned(d);
    c.BP = wide ? 256 : 128; c.BQ = 64;
  }
  else if (d->Cout > 16) { c.BP = 256; c.BQ = 32; }      // (512 x 32, one 128 x 32 tile per wave: measured slower)
  else { c.BP = 256; c.BQ = 16; }
  return c;
}
static int wgrad_ps(int dtype) { return dtype == CSMRI_BF16 ? 64 : 16; }

#ifndef WSK_TARGET
#define WSK_TARGET 512
#endif
extern "C" int csmri_wgrad_suggest_splitk(const csmri_wgrad_desc* d) {
  if (wthin_out_eligible(d)) return wthin_splits(d);
  if (whead_eligible(d)) return whead_splits(d);
  if (wrow_eligible(d)) return wrow_groups(d);
  if (wpatch_eligible(d)) return wpatch_groups(d);
  WConfig c = pick_wconfig(d);
  const long long NK = (long long)d->KH * d->KW * d->Cin;
  const long long tiles = (long long)cdiv(NK, c.BP) * cdiv(d->Cout, c.BQ);
  const int nsteps = cdiv((long long)d->B * d->Ho * d->Wo, wgrad_ps(d->dtype));
  const int target = WSK_TARGET;
  int sk = (int)((target + tiles - 1) / tiles);
  int maxsk = nsteps / 4; if (maxsk < 1) maxsk = 1;
  if (sk > maxsk) sk = maxsk;
  if (sk > 512) sk = 512;
  return sk < 1 ? 1 : sk;
}
extern "C" size_t csmri_wgrad_slab_bytes(const csmri_wgrad_desc* d) {
  const int sk = d->splitk > 0 ? d->splitk : 1;
  return ((size_t)sk * d->Cout * d->KH * d->KW * d->Cin + (size_t)DB_ROWS * d->Cout) * sizeof(float);
}

template <int DT, int BP, int BQ, int WP, int WQ>
static int launch_wgrad(const WParams& p, hipStream_t st) {
  constexpr int KC = DT == CSMRI_BF16 ? 2 : 1;
  constexpr int lds = 2 * KC * (BP + BQ) * 64;
  auto kern = wgrad_kernel<DT, BP, BQ, WP, WQ>;
  CSMRI_SET_MAX_LDS(kern, lds);
  hipLaunchKernelGGL(kern, dim3(p.ptiles * p.qtiles, 1, p.splitk), dim3(256), lds, st, p);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

template <int BP, int BQ, int WP, int WQ>
static int launch_wgrad_glds(const WParams& p, hipStream_t st) {
  constexpr int lds = 64 * (BP + BQ) * 2;
  const bool row_aligned = (p.Wo % 64 == 0 || (64 % p.Wo == 0 && (p.Ho * p.Wo) % 64 == 0)) &&
                           (long long)p.B * p.Hin * p.Win < (1 << 24) && p.ps0 < (1 << 22) && p.ps1 < (1 << 22) &&
                           (long long)p.B * p.Hin * p.Win * (p.ps0 > p.ps1 ? p.ps0 : p.ps1) * 2 < (1ll << 32);
  if (row_aligned) {
    const dim3 grid(p.ptiles * p.qtiles, 1, p.splitk);
    const bool refl = p.border == CSMRI_BORDER_REFLECT;
    if constexpr (BP == 128 && BQ == 128 && WP == 2 && WQ == 2) {
      if (p.dwd) {                                     // (wgrad_direct_ok: no split, 16 taps, no upsampling, real == padded channels)
        CSMRI_SET_MAX_LDS((wgrad_glds_row_kernel<128, 128, 2, 2, true, false, true>), lds);
        CSMRI_SET_MAX_LDS((wgrad_glds_row_kernel<128, 128, 2, 2, false, false, true>), lds);
        if (refl) hipLaunchKernelGGL((wgrad_glds_row_kernel<128, 128, 2, 2, true, false, true>), grid, dim3(256), lds, st, p);
        else hipLaunchKernelGGL((wgrad_glds_row_kernel<128, 128, 2, 2, false, false, true>), grid, dim3(256), lds, st, p);
        CSMRI_LAUNCH_CHECK();
        return CSMRI_OK;
      }
    }
    CSMRI_SET_MAX_LDS((wgrad_glds_row_kernel<BP, BQ, WP, WQ, true, true>), lds);
    CSMRI_SET_MAX_LDS((wgrad_glds_row_kernel<BP, BQ, WP, WQ, true, false>), lds);
    CSMRI_SET_MAX_LDS((wgrad_glds_row_kernel<BP, BQ, WP, WQ, false, true>), lds);
    CSMRI_SET_MAX_LDS((wgrad_glds_row_kernel<BP, BQ, WP, WQ, false, false>), lds);
    const dim3 blk(64 * WP * WQ);
    if (refl && p.ups) hipLaunchKernelGGL((wgrad_glds_row_kernel<BP, BQ, WP, WQ, true, true>), grid, blk, lds, st, p);
    else if (refl) hipLaunchKernelGGL((wgrad_glds_row_kernel<BP, BQ, WP, WQ, true, false>), grid, blk, lds, st, p);
    else if (p.ups) hipLaunchKernelGGL((wgrad_glds_row_kernel<BP, BQ, WP, WQ, false, true>), grid, blk, lds, st, p);
    else hipLaunchKernelGGL((wgrad_glds_row_kernel<BP, BQ, WP, WQ, false, false>), grid, blk, lds, st, p);
  } else {
    if constexpr (WP * WQ != 4) return CSMRI_E_UNSUPPORTED;     // (callers pick 8 waves only for row-aligned geometries)
    else hipLaunchKernelGGL((wgrad_glds_kernel<BP, BQ, WP, WQ>), dim3(p.ptiles * p.qtiles, 1, p.splitk), dim3(256), lds,
                            st, p);
  }
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

// the geometry conditions of the row-aligned kernel (launch_wgrad_glds)
static bool wgrad_row_launchable(const csmri_wgrad_desc* d) {
  return wgrad_row_aligned(d) && (long long)d->B * d->Hin * d->Win < (1 << 24) &&
         d->in0_pix_stride < (1 << 22) && d->in1_pix_stride < (1 << 22) &&
         (long long)d->B * d->Hin * d->Win * (d->in0_pix_stride > d->in1_pix_stride ? d->in0_pix_stride : d->in1_pix_stride) * 2 < (1ll << 32);
}
#ifndef WGRAD_DIRECT
#define WGRAD_DIRECT 1
#endif
// wgrad_glds_row_kernel<128,128,2,2,..,DIRECT>: writes the reference-layout gradient itself
static bool wgrad_direct_ok(const csmri_wgrad_desc* d, int BP, int BQ, bool row) {
  return WGRAD_DIRECT && row && d->dtype == CSMRI_BF16 && BP == 128 && BQ == 128 && (d->splitk <= 1) && !d->defer_finish &&
         d->KH * d->KW == 16 && !d->upsample && d->Cin_real == d->Cin && d->Cout_real == d->Cout && d->Cout % 128 == 0 &&
         !((uintptr_t)d->dw & 15);
}

// template instance csmri_wgrad dispatches to for this problem, spelled as rocprofv3 prints it
extern "C" int csmri_wgrad_kernel_name(const csmri_wgrad_desc* d, char* buf, int n) {
  CSMRI_CHECK_ARG(d && buf && n > 0);
  if (wthin_out_eligible(d)) { snprintf(buf, n, "wthin_out_kernel"); return CSMRI_OK; }
  if (whead_eligible(d)) { snprintf(buf, n, "whead_kernel<%d>", d->Cin / 8); return CSMRI_OK; }
  if (wrow_eligible(d)) { wrow_kernel_name(d, buf, n); return CSMRI_OK; }
  if (wpatch_eligible(d)) {
    snprintf(buf, n, "wpatch_kernel<%d, %d, %d>", d->Cin, d->Cin == 128 ? 32 : wpatch_cout(d),
             d->Cin == 128 ? 8 : (d->Cin == 64 && wpatch_cout(d) == 64 ? 8 : 4));
    return CSMRI_OK;
  }
  WConfig c = pick_wconfig(d);
  const int wp = c.BQ >= 128 || (c.BQ == 64 && c.BP == 128) ? 2 : 4, wq = wp == 2 ? 2 : 1;
  if (d->dtype != CSMRI_BF16) { snprintf(buf, n, "wgrad_kernel<%d, %d, %d, %d, %d>", d->dtype, c.BP, c.BQ, wp, wq); return CSMRI_OK; }
  const bool row = wgrad_row_aligned(d) && (long long)d->B * d->Hin * d->Win < (1 << 24) &&
                   d->in0_pix_stride < (1 << 22) && d->in1_pix_stride < (1 << 22) &&
                   (long long)d->B * d->Hin * d->Win * (d->in0_pix_stride > d->in1_pix_stride ? d->in0_pix_stride : d->in1_pix_stride) * 2 < (1ll << 32);
  if (row) snprintf(buf, n, "wgrad_glds_row_kernel<%d, %d, %d, %d, %s, %s, %s>", c.BP, c.BQ, wp, wq,
                    d->border == CSMRI_BORDER_REFLECT ? "true" : "false", d->upsample ? "true" : "false",
                    wgrad_direct_ok(d, c.BP, c.BQ, row) ? "true" : "false");
  else snprintf(buf, n, "wgrad_glds_kernel<%d, %d, %d, %d>", c.BP, c.BQ, wp, wq);
  return CSMRI_OK;
}

extern "C" int csmri_wgrad(const csmri_wgrad_desc* d, void* stream) {
  CSMRI_CHECK_ARG(d && d->in0 && d->dy && d->dw && d->slab);
  CSMRI_CHECK_ARG(d->dtype == CSMRI_F32 || d->dtype == CSMRI_BF16);
  CSMRI_CHECK_ARG(d->Cin > 0 && d->Cin % 8 == 0 && d->Cout > 0 && d->Cout % 8 == 0);
  CSMRI_CHECK_ARG(d->Cin_real > 0 && d->Cin_real <= d->Cin && d->Cout_real > 0 && d->Cout_real <= d->Cout);
  CSMRI_CHECK_ARG(d->in0_pix_stride % 8 == 0 && d->dy_pix_stride % 8 == 0);
  if (d->in1) CSMRI_CHECK_ARG(d->c0 > 0 && d->c0 % 8 == 0 && d->c0 < d->Cin && d->in1_pix_stride % 8 == 0);
  if (((uintptr_t)d->in0 | (uintptr_t)d->in1 | (uintptr_t)d->dy | (uintptr_t)d->slab) & 15) return CSMRI_E_ALIGN;
  WConfig c = pick_wconfig(d);
  WParams p;
  p.in0 = (const char*)d->in0; p.in1 = (const char*)d->in1; p.ps0 = d->in0_pix_stride; p.ps1 = d->in1_pix_stride;
  p.c0 = d->in1 ? d->c0 : d->Cin;
  p.B = d->B; p.Hin = d->Hin; p.Win = d->Win; p.Cin = d->Cin; p.ups = d->upsample; p.border = d->border;
  p.KH = d->KH; p.KW = d->KW; p.S = d->stride; p.pt = d->pad_t; p.pl = d->pad_l;
  p.dy = (const char*)d->dy; p.dyps = d->dy_pix_stride; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
  p.slab = d->slab; p.splitk = d->splitk > 0 ? d->splitk : 1;
  p.M = d->B * d->Ho * d->Wo; p.NK = d->KH * d->KW * d->Cin;
  p.nsteps = cdiv(p.M, wgrad_ps(d->dtype));
  p.steps_per_split = cdiv(p.nsteps, p.splitk);
  p.ptiles = cdiv(p.NK, c.BP); p.qtiles = cdiv(d->Cout, c.BQ);
  hipStream_t st = (hipStream_t)stream;
  // a split whose step range is empty still has to define its slab: zero everything first
  // when the split count does not divide evenly (cheap; slabs are small next to activations)
  const bool head = whead_eligible(d);
  const bool thin = wthin_out_eligible(d) || head;   // (write every slab entry they own themselves)
  if (!thin && (long long)p.steps_per_split * (p.splitk - 1) >= p.nsteps) {
    hipError_t e = hipMemsetAsync(d->slab, 0, (size_t)p.splitk * d->Cout * p.NK * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
  }
  int rc;
  bool direct = false;
  p.dwd = nullptr; p.dw_acc = 0;
#define WG(DT_, BP_, BQ_, WP_, WQ_) rc = launch_wgrad<DT_, BP_, BQ_, WP_, WQ_>(p, st)
  const bool patch = thin || wpatch_eligible(d) || wrow_eligible(d);   // (all leave their bias-gradient partial rows behind the slabs)
  if (thin) {
    rc = head ? whead_launch(p, d, st) : wthin_launch(p, d, st);
  } else if (patch) {
    WParams q = p;
    q.nsteps = d->db ? 1 : 0;                      // wpatch / wrow reuse the field: also produce the bias-gradient partials
    rc = wrow_eligible(d) ? wrow_launch(q, d, st) : wpatch_launch(q, d, st);
  } else if (d->dtype == CSMRI_BF16) {
    direct = wgrad_direct_ok(d, c.BP, c.BQ, wgrad_row_launchable(d));
    if (direct) { p.dwd = d->dw; p.dw_acc = d->accumulate; }
    if (c.BQ == 128) rc = launch_wgrad_glds<128, 128, 2, 2>(p, st);
    else if (c.BQ == 64 && c.BP == 256) rc = launch_wgrad_glds<256, 64, 4, 1>(p, st);
    else if (c.BQ == 64) rc = launch_wgrad_glds<128, 64, 2, 2>(p, st);
    else if (c.BQ == 32) rc = launch_wgrad_glds<256, 32, 4, 1>(p, st);
    else rc = launch_wgrad_glds<256, 16, 4, 1>(p, st);
  } else {
    if (c.BQ == 128) WG(CSMRI_F32, 128, 128, 2, 2);
    else if (c.BQ == 64) WG(CSMRI_F32, 128, 64, 2, 2);
    else if (c.BQ == 32) WG(CSMRI_F32, 256, 32, 4, 1);
    else WG(CSMRI_F32, 256, 16, 4, 1);
  }
#undef WG
  if (rc != CSMRI_OK) return rc;
  if (d->defer_finish) {
    // the slab reduction (and the patch kernels' bias partials) are left to csmri_wgrad_finish_multi; the bias
    // gradient of the other kernels is a column sum over dY, which may not outlive this call: done now
    if (d->db && !patch) {
      float* part = d->slab + (size_t)p.splitk * d->Cout * p.NK;
      int rows = cdiv(p.M, 512); if (rows > 256) rows = 256; if (rows < 1) rows = 1;   // (final pass: rows / 8 serial steps)
      hipLaunchKernelGGL(colsum_partial_kernel, dim3(rows), dim3(256), 0, st, d->dtype, p.dy, p.dyps,
                         (long long)p.M, d->Cout, part);
      CSMRI_LAUNCH_CHECK();
      hipLaunchKernelGGL(colsum_final_kernel, dim3((d->Cout_real + 31) / 32), dim3(256), 0, st, part, rows,
                         d->Cout, d->Cout_real, d->db, d->accumulate);
      CSMRI_LAUNCH_CHECK();
    }
    return CSMRI_OK;
  }
  if (direct) {
    // (the kernel wrote the reference-layout gradient itself)
  } else if (p.splitk <= 8) {        // big layers: bandwidth-bound transposing copy
    if (d->KH * d->KW <= 16 && (long long)d->Cout_real * ((d->Cin_real + 63) / 64) >= 4096)
      hipLaunchKernelGGL((wgrad_scatter_t_rows_kernel<4>), dim3((d->Cout_real + 3) / 4, (d->Cin_real + 63) / 64), dim3(256), 0, st,
                         d->slab, p.splitk, d->Cout, p.NK, d->Cin, d->KH * d->KW, d->Cin_real, d->Cout_real, d->dw,
                         d->accumulate);
    else
      hipLaunchKernelGGL(wgrad_scatter_t_kernel, dim3(d->Cout_real, (d->Cin_real + 63) / 64), dim3(256), 0, st,
                         d->slab, p.splitk, d->Cout, p.NK, d->Cin, d->KH * d->KW, d->Cin_real, d->dw,
                         d->accumulate);
  } else {                    // few outputs, many splits: one thread per output element
    const long long total = (long long)d->Cout_real * p.NK;
    int blocks = (int)((total + 31) / 32); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(wgrad_scatter_kernel, dim3(blocks), dim3(256), 0, st, d->slab, p.splitk, d->Cout, p.NK,
                       d->Cin, d->KH, d->KW, d->Cout_real, d->Cin_real, d->dw, d->accumulate);
  }
  CSMRI_LAUNCH_CHECK();
  if (d->db && patch) {
    // the patch kernel left one partial row per workgroup behind the slabs
    float* part = d->slab + (size_t)p.splitk * d->Cout * p.NK;
    hipLaunchKernelGGL(colsum_final_kernel, dim3((d->Cout_real + 31) / 32), dim3(256), 0, st, part, p.splitk,
                       d->Cout, d->Cout_real, d->db, d->accumulate);
    CSMRI_LAUNCH_CHECK();
  } else if (d->db) {
    float* part = d->slab + (size_t)p.splitk * d->Cout * p.NK;
    int rows = cdiv(p.M, 512); if (rows > 256) rows = 256; if (rows < 1) rows = 1;   // (final pass: rows / 8 serial steps)
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(rows), dim3(256), 0, st, d->dtype, p.dy, p.dyps,
                       (long long)p.M, d->Cout, part);
    CSMRI_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_final_kernel, dim3((d->Cout_real + 31) / 32), dim3(256), 0, st, part, rows,
                       d->Cout, d->Cout_real, d->db, d->accumulate);
    CSMRI_LAUNCH_CHECK();
  }
  return CSMRI_OK;
}

// The deferred part of up to n csmri_wgrad calls made with defer_finish = 1 (same descriptors, slabs still alive), as
// ONE launch per WFINISH_MAX layers: the per-layer slab reductions are 5-20 us launches that otherwise sit one behind
// each main kernel in the serial tail of a backward pass.  Same arithmetic per element as the single-layer kernels.
// Two descriptors of one call must not share dw (their read-modify-writes would race): the caller finishes the
// earlier one first.
extern "C" int csmri_wgrad_finish_multi(const csmri_wgrad_desc* descs, int n, void* stream) {
  CSMRI_CHECK_ARG(descs && n > 0);
  hipStream_t st = (hipStream_t)stream;
  for (int i0 = 0; i0 < n; i0 += WFINISH_MAX) {
    const int m = n - i0 < WFINISH_MAX ? n - i0 : WFINISH_MAX;
    WFinishTable T;
    int total_blocks = 0;
    for (int i = 0; i < m; ++i) {
      const csmri_wgrad_desc* d = descs + i0 + i;
      CSMRI_CHECK_ARG(d->dw && d->slab && d->Cin > 0 && d->Cout > 0 && d->KH > 0 && d->KW > 0);
      for (int j = 0; j < i; ++j) CSMRI_CHECK_ARG(descs[i0 + j].dw != d->dw);
      WFinishItem& t = T.it[i];
      t.splitk = d->splitk > 0 ? d->splitk : 1;
      t.Cout = d->Cout; t.NK = d->KH * d->KW * d->Cin; t.Cin = d->Cin; t.KH = d->KH; t.KW = d->KW;
      t.Cout_real = d->Cout_real; t.Cin_real = d->Cin_real; t.accumulate = d->accumulate;
      t.slab = d->slab; t.dw = d->dw;
      const bool patch = wpatch_eligible(d) || wrow_eligible(d) || wthin_out_eligible(d) || whead_eligible(d) || d->defer_finish == 2;
      t.db = patch ? d->db : nullptr;                  // (other kernels: bias gradient already written by csmri_wgrad)
      t.part = d->slab + (size_t)t.splitk * d->Cout * t.NK; t.part_rows = t.splitk;
      long long nb = t.splitk <= 8 ? (long long)t.Cout_real * ((t.Cin_real + 63) / 64)
                                   : ((long long)t.Cout_real * t.NK + 31) / 32;
      if (nb > 1024) nb = 1024;                        // the bodies stride over the rest
      if (nb < 1) nb = 1;
      T.first[i] = total_blocks;
      total_blocks += (int)nb;
    }
    T.first[m] = total_blocks;
    hipLaunchKernelGGL(wgrad_finish_multi_kernel, dim3(total_blocks), dim3(256), 0, st, T, m);
    CSMRI_LAUNCH_CHECK();
  }
  return CSMRI_OK;
}
