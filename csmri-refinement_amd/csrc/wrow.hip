// wrow: weight gradient of the U-Net's 4 x 4 stride-1 layers on the 256^2 / 128^2 maps (reference models/unet.py:48,100,241
// backward; Cin, Cout in {32, 64}, 128 -> 64 split over workgroups) as a ROW-STREAMING kernel -- uconv.hip's sliding window
// applied to the reduction over pixels.
//
//   dW[n][ty][tx][ci] = sum_{b, y, x} dY[b][y][x][n] * Xpad[b][y + ty][x + tx][ci]
//
// wpatch (wgrad.hip) owns 16 x 16 output pixels at a time: per 32-pixel K chunk (two tile rows) every tap reads its own
// shifted X fragments, 0.4-0.6 fragment reads per MFMA, the tile's images are staged between two barriers, 320-500 TFLOP/s.
// Here a workgroup streams a COLUMN STRIP of 32 output columns row by row: a K chunk is one row of 32 pixels, the X fragment
// of padded row R shifted by the filter column tx pairs with dY rows R, R-1, R-2, R-3 (filter rows 0..3).  Compute wave tx
// keeps the accumulators of its filter column -- acc[ty][ci fragment][n fragment] -- for the whole launch and, per row, reads
// the CF fragments of X(R) and the NFY fragments of one dY row ONCE for TH x CF x NFY = 16-32 MFMAs (0.19-0.25 reads per
// MFMA); the other operand's last four rows stay in registers (the side with fewer fragments).
//   * 4 compute waves (one per filter column) + 4 loader waves; rows arrive by LDS-DMA into a ring of 12 row slots (3 groups of
//     4 rows; a group is loaded two groups ahead); one workgroup barrier per group = per 64-128 MFMAs of a wave.
//   * Work item = (image, column strip, segment of SEG rows): SEG + 3 padded rows, rounded up to whole groups.  Rows of dY
//     outside the item and rows / columns of X outside the image (zero border) are loaded as ZEROS, so the compute waves run
//     the same unconditional MFMA schedule on every row; the register window ends every item full of zero rows.
//   * Same slab contract as wpatch: one fp32 slab [Cout][taps * Cin] per workgroup (grid.x = split count), bias-gradient
//     partial rows behind the slabs (one more MFMA per dY fragment against a fragment of ones).
// LDS images: [pixel][channel] rows read with ds_read_b64_tr_b16 (wgrad_params.h), row slots of 48 / 32 pixels (multiples of
// 16: the chunk swizzle does not depend on the slot, slot offsets are immediates).
#include <utility>
#include "mma_core.h"
#include "wgrad_params.h"

__device__ __attribute__((aligned(16))) char wr_zero_page[16];

#define WR_RPY 32          // pixels per dY row slot
#define WR_GR 4            // rows per group (one workgroup barrier per group)
#define WR_RING (3 * WR_GR)   // row slots: a group is loaded two groups ahead

template <int N> __device__ __forceinline__ void wr_vmwait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int... I, class F>
__device__ __forceinline__ void wr_unroll(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }

// CF: 16-channel fragments of the input channels a workgroup owns (grid.z splits more), NFY: of the output channels
// (grid.y splits more); WINX: the register window holds X rows (CF <= NFY), else dY rows
template <int CF, int NFY, bool WINX>
__global__ __launch_bounds__(512, 2) void wrow_kernel(const WParams p) {
  constexpr int TH = 4, CPR = CF * 2, CQ = NFY * 2;           // 16-byte chunks per X / dY pixel
  constexpr int PXP = 64 / CPR, PYP = 64 / CQ;                 // pixels per LDS-DMA piece
  constexpr int XP = (35 + PXP - 1) / PXP, YP = 32 / PYP;      // pieces per row
  // X row slot: whole pieces (40 or 48 pixels, 35 used); every slot starts on a 256-byte bank row and the chunk swizzle is a
  // function of the column inside the slot, so a slot is an immediate offset (all of them below 64 KiB)
  constexpr int SLOTX = XP * 1024, SLOTY = WR_RPY * CQ * 16, YOFF = WR_RING * SLOTX;
  constexpr int PG = WR_GR * (XP + YP) / 4;                    // pieces per loader wave and group of rows
  constexpr int LAG = WINX ? 3 : 0;                            // the dY row read with padded row R is R - LAG
  static_assert((WR_GR * (XP + YP)) % 4 == 0 && (WR_GR % 4) == 0 && WR_RING * (SLOTX + SLOTY) <= 160 * 1024, "geometry");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int items = p.wr_items, SEG = p.wr_seg, NXS = p.wr_xs, NSEG = p.wr_segs;
  const int GPI = (SEG + 3 + WR_GR - 1) / WR_GR;               // groups of padded rows per item
  const int worker = blockIdx.x, workers = gridDim.x;
  const int my_items = worker < items ? (items - worker + workers - 1) / workers : 0;
  const int NG = (my_items * GPI + 2) / 3 * 3;                 // groups, padded to whole rounds of the slot ring (zero rows)
  const int qoff = blockIdx.y * NFY * 16, coff = blockIdx.z * CF * 16;   // this workgroup's output / input channels

  if (wv >= 4) {
    // =================================================== loader waves ===================================================
    if (NG == 0) return;
    const int L = wv - 4;
    const int Hv = p.ups ? 2 * p.Hin : p.Hin, Wv = p.ups ? 2 * p.Win : p.Win;
    const int xpix = lane / CPR, xslot = lane % CPR, ypix = lane / CQ, yslot = lane % CQ;
    // per item: this lane's source column of every X piece (-1: outside a zero border), the dY column offsets
    int it_b = 0, it_x0 = 0, it_y0 = 0;
    bool it_null = false;                              // padding groups behind the last item: zero rows
    int xcol[XP]; unsigned xchunk[XP];
    unsigned ychunk[YP];
#pragma unroll
    for (int i = 0; i < YP; ++i) ychunk[i] = (unsigned)(((img_off<CQ>(i * PYP + ypix, yslot) >> 4) % CQ) * 16);
    auto set_item = [&](int k) {                       // k-th item of this workgroup
      const int id = worker + k * workers;
      const int seg = id % NSEG, t_ = id / NSEG, xs = t_ % NXS;
      it_b = t_ / NXS; it_x0 = xs * 32; it_y0 = seg * SEG;
#pragma unroll
      for (int i = 0; i < XP; ++i) {
        const int col = i * PXP + xpix;
        int w = it_x0 - p.pl + col;
        if (p.border == CSMRI_BORDER_REFLECT) { w = w < 0 ? -w : w; w = min(w, 2 * (Wv - 1) - w); w = max(w, 0); }
        const bool ok = (unsigned)w < (unsigned)Wv;
        if (p.ups) w >>= 1;
        xcol[i] = ok ? w : -1;
        xchunk[i] = (unsigned)((img_off<CPR>(col, xslot) >> 4) % CPR);
      }
    };
    const char* zero_page = wr_zero_page;
    // piece q (0 .. 4 (XP + YP) - 1) of group (item k, group gi): row j = q / (XP + YP), then X pieces, then dY pieces
    auto issue_group = [&](int gi, int slot0) {
#pragma unroll
      for (int qq = 0; qq < PG; ++qq) {
        const int q = L + 4 * qq, j = q / (XP + YP), i = q % (XP + YP);
        const int R = it_y0 + gi * WR_GR + j;          // padded row of the image
        char* dst;
        const char* src;
        if (i < XP) {
          int u = R - p.pt;
          if (p.border == CSMRI_BORDER_REFLECT) { u = u < 0 ? -u : u; u = min(u, 2 * (Hv - 1) - u); u = max(u, 0); }
          const bool rok = (unsigned)u < (unsigned)Hv && gi * WR_GR + j < SEG + 3 && !it_null;
          if (p.ups) u >>= 1;
          int xc = 0; unsigned ch = 0;
#pragma unroll
          for (int ii = 0; ii < XP; ++ii) if (ii == i) { xc = xcol[ii]; ch = xchunk[ii]; }
          const unsigned c = (unsigned)coff + ch * 8u;           // channel of this lane's chunk
          const size_t pix = ((size_t)it_b * p.Hin + u) * p.Win + xc;
          src = c < (unsigned)p.c0 ? p.in0 + (pix * p.ps0 + c) * 2 : p.in1 + (pix * p.ps1 + (c - p.c0)) * 2;
          src = (rok && xc >= 0) ? src : zero_page;
          dst = smem + (slot0 + j) * SLOTX + i * 1024;
        } else {
          const int iy = i - XP;
          const int y = R - LAG;
          const bool rok = y >= it_y0 && y < it_y0 + SEG && y < p.Ho && !it_null;
          unsigned ch = 0;
#pragma unroll
          for (int ii = 0; ii < YP; ++ii) if (ii == iy) ch = ychunk[ii];
          const size_t pix = ((size_t)it_b * p.Ho + y) * p.Wo + it_x0 + iy * PYP + ypix;
          src = rok ? p.dy + (pix * p.dyps + qoff) * 2 + ch : zero_page;
          dst = smem + YOFF + (slot0 + j) * SLOTY + iy * 1024;
        }
        __builtin_amdgcn_global_load_lds((wgptr_t)src, (wlptr_t)dst, 16, 0, 0);
      }
    };
    // flat sequence of groups over this workgroup's items; group G lives in slots 4 (G % 3) ..
    int lk = 0, lgi = 0, lslot = 0;                    // next group to issue: item, group in item, first slot
    set_item(0);
    auto issue_next = [&]() {
      issue_group(lgi, lslot);
      lslot = lslot == 2 * WR_GR ? 0 : lslot + WR_GR;
      if (++lgi == GPI) { lgi = 0; ++lk; if (lk < my_items) set_item(lk); else it_null = true; }
    };
    issue_next();
    if (NG > 1) issue_next();
    for (int G = 0; G < NG; ++G) {
      if (G + 1 < NG) wr_vmwait<PG>(); else wr_vmwait<0>();    // group G has landed (group G+1 may be in flight)
      __builtin_amdgcn_s_barrier();
      if (G + 2 < NG) issue_next();                    // into the slots group G-1 has left
    }
    return;
  }

  // ===================================================== compute waves =====================================================
  const int tx = wv;                                   // this wave's filter column
  const int g = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3, r16 = lane & 15;
  const int klo = 8 * g + tq;                          // this lane's pixel of the 32-pixel K chunk (+4 for the high half)
  f32x4_t acc[TH][CF][NFY], bacc[NFY];
#pragma unroll
  for (int a = 0; a < TH; ++a)
#pragma unroll
    for (int c = 0; c < CF; ++c)
#pragma unroll
      for (int n = 0; n < NFY; ++n) acc[a][c][n] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int n = 0; n < NFY; ++n) bacc[n] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, (u32x4_t){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});
  const bf16x8_t zfrag = __builtin_bit_cast(bf16x8_t, (u32x4_t){0u, 0u, 0u, 0u});
  constexpr int WN = WINX ? CF : NFY;
  bf16x8_t win[4][WN];                                 // the last four rows of the operand that stays in registers
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int c = 0; c < WN; ++c) win[j][c] = zfrag;
  // fragment offsets inside a row slot (wgrad_params.h): the fragments of a row differ by an XOR of the chunk bits
  int xlo[CF], xhi[CF], ylo[NFY], yhi[NFY];
  {
    const int x0 = img_off<CPR>(tx + klo, tp >> 1) + (tp & 1) * 8, x1 = img_off<CPR>(tx + klo + 4, tp >> 1) + (tp & 1) * 8;
    const int y0 = img_off<CQ>(klo, tp >> 1) + (tp & 1) * 8, y1 = img_off<CQ>(klo + 4, tp >> 1) + (tp & 1) * 8;
#pragma unroll
    for (int c = 0; c < CF; ++c) { xlo[c] = x0 ^ (c << 5); xhi[c] = x1 ^ (c << 5); }
#pragma unroll
    for (int n = 0; n < NFY; ++n) { ylo[n] = (y0 ^ (n << 5)) + YOFF; yhi[n] = (y1 ^ (n << 5)) + YOFF; }
  }
  const bool want_db = p.nsteps != 0 && wv == 0 && blockIdx.z == 0;

  auto group = [&](auto slotc) {                       // four padded rows in slots slot0 .. slot0 + 3
    constexpr int slot0 = decltype(slotc)::value;
    wr_unroll(std::make_integer_sequence<int, WR_GR>{}, [&](auto jc) {
      constexpr int jj = decltype(jc)::value, j = jj & 3, s = slot0 + jj;
      bf16x8_t xf[CF], yf[NFY];
#pragma unroll
      for (int c = 0; c < CF; ++c) xf[c] = tr_frag(smem + s * SLOTX, xlo[c], xhi[c]);
#pragma unroll
      for (int n = 0; n < NFY; ++n) yf[n] = tr_frag(smem + s * SLOTY, ylo[n], yhi[n]);
      if (want_db) {
#pragma unroll
        for (int n = 0; n < NFY; ++n) bacc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, yf[n], bacc[n], 0, 0, 0);
      }
      if constexpr (WINX) {
        // X(R) enters the window; the dY row read here is y = R - 3 and pairs with X(y + ty) = window[(j + 1 + ty) & 3]
#pragma unroll
        for (int c = 0; c < CF; ++c) win[j][c] = xf[c];
#pragma unroll
        for (int ty = 0; ty < TH; ++ty)
#pragma unroll
          for (int c = 0; c < CF; ++c)
#pragma unroll
            for (int n = 0; n < NFY; ++n)
              acc[ty][c][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(win[(j + 1 + ty) & 3][c], yf[n], acc[ty][c][n], 0, 0, 0);
      } else {
        // dY(R) enters the window; X(R) pairs with dY(R - ty) = window[(j - ty) & 3]
#pragma unroll
        for (int n = 0; n < NFY; ++n) win[j][n] = yf[n];
#pragma unroll
        for (int ty = 0; ty < TH; ++ty)
#pragma unroll
          for (int c = 0; c < CF; ++c)
#pragma unroll
            for (int n = 0; n < NFY; ++n)
              acc[ty][c][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[c], win[(j - ty) & 3][n], acc[ty][c][n], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);               // (rows stay apart: hoisting four rows of fragment reads spills)
    });
  };
  for (int G = 0; G < NG; G += 3) {                    // (NG is a multiple of 3: no branch between the slot variants)
    __syncthreads();                                   // group G is visible; every wave is done with group G-1's slots
    group(std::integral_constant<int, 0>{});
    __syncthreads();
    group(std::integral_constant<int, WR_GR>{});
    __syncthreads();
    group(std::integral_constant<int, 2 * WR_GR>{});
  }

  // ---- one slab per workgroup: [Cout][NK], NK index = tap * Cin + ci (wpatch's contract) ---------------------------
  if (want_db && g == 0) {                             // row 0 of the ones product: sum over pixels per output channel
    float* part = p.slab + (size_t)gridDim.x * p.Cout * p.NK + (size_t)blockIdx.x * p.Cout;
#pragma unroll
    for (int n = 0; n < NFY; ++n) part[qoff + n * 16 + r16] = bacc[n][0];
  }
#pragma unroll
  for (int ty = 0; ty < TH; ++ty)
#pragma unroll
    for (int n = 0; n < NFY; ++n) {
      const int co = qoff + n * 16 + r16;
#pragma unroll
      for (int c = 0; c < CF; ++c) {
        const int cc = (ty * 4 + tx) * p.Cin + coff + c * 16 + g * 4;
        *(f32x4_t*)(p.slab + ((size_t)blockIdx.x * p.Cout + co) * p.NK + cc) = acc[ty][c][n];
      }
    }
}

// ---------------------------------------------------------------------------------------------
static void wrow_shape(const csmri_wgrad_desc* d, int* cf, int* nfy, int* ny, int* nz) {
  // fragments per workgroup.  Built and compiled: 4 x 2 and 2 x 4 (32 accumulator fragments per wave): 59-82 spilled
  // registers inside the row loop at the 256-register cap of two waves per SIMD; 2 x 2 (16 fragments, 151 registers)
  // with the rest of the channels on grid.y / grid.z is what runs: X is then read Cout/32 times, dY Cin/32 times
  *cf = 2; *nfy = 2;
  *ny = d->Cout / (*nfy * 16); *nz = d->Cin / (*cf * 16);
}
#ifndef WROW_WGS
#define WROW_WGS 256       // workgroups in all (grid.x * grid.y * grid.z): one per CU; grid.x of them = slabs
#endif
// rows per work item: the longest segment that still gives every workgroup an item
static int wrow_seg(const csmri_wgrad_desc* d) {
  int cf, nfy, ny, nz;
  wrow_shape(d, &cf, &nfy, &ny, &nz);
  long long want = WROW_WGS / (ny * nz); if (want < 16) want = 16;
  for (int seg = d->Ho; seg >= 16; seg >>= 1)
    if (d->Ho % seg == 0 && (long long)d->B * (d->Wo / 32) * (d->Ho / seg) >= want) return seg;
  return d->Ho % 16 == 0 ? 16 : 0;
}
bool wrow_eligible(const csmri_wgrad_desc* d) {
  if (d->dtype != CSMRI_BF16 || d->stride != 1 || d->KH != 4 || d->KW != 4) return false;
  if (!(d->Cin == 32 || d->Cin == 64 || d->Cin == 128) || !(d->Cout == 32 || d->Cout == 64 || d->Cout == 128)) return false;
  if (d->in1 && (d->c0 % 8)) return false;
  if (d->Wo % 32 || d->Ho < 64 || d->Wo < 64 || wrow_seg(d) == 0) return false;
  if (d->Ho != (d->upsample ? 2 * d->Hin : d->Hin) || d->Wo != (d->upsample ? 2 * d->Win : d->Win)) return false;   // SAME padding
  return true;
}
int wrow_groups(const csmri_wgrad_desc* d) {
  const int seg = wrow_seg(d);
  int cf, nfy, ny, nz;
  wrow_shape(d, &cf, &nfy, &ny, &nz);
  const long long items = (long long)d->B * (d->Wo / 32) * (d->Ho / seg);
  long long g = WROW_WGS / (ny * nz);
  if (g < 16) g = 16;
  return (int)(items < g ? items : g);
}
template <int CF, int NFY, bool WINX>
static int launch_wrow(const WParams& p, dim3 grid, hipStream_t st) {
  constexpr int lds = WR_RING * (((35 + 32 / CF - 1) / (32 / CF)) * 1024 + WR_RPY * NFY * 32);
  static_assert(lds <= 160 * 1024, "LDS");
  CSMRI_SET_MAX_LDS((wrow_kernel<CF, NFY, WINX>), lds);
  hipLaunchKernelGGL((wrow_kernel<CF, NFY, WINX>), grid, dim3(512), lds, st, p);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}
int wrow_launch(const WParams& p0, const csmri_wgrad_desc* d, hipStream_t st) {
  WParams p = p0;
  int cf, nfy, ny, nz;
  wrow_shape(d, &cf, &nfy, &ny, &nz);
  p.wr_seg = wrow_seg(d); p.wr_xs = d->Wo / 32; p.wr_segs = d->Ho / p.wr_seg;
  p.wr_items = d->B * p.wr_xs * p.wr_segs;
  if (p.splitk > p.wr_items) return CSMRI_E_ARG;
  const dim3 grid(p.splitk, ny, nz);
  return launch_wrow<2, 2, false>(p, grid, st);
}
void wrow_kernel_name(const csmri_wgrad_desc* d, char* buf, int n) {
  int cf, nfy, ny, nz;
  wrow_shape(d, &cf, &nfy, &ny, &nz);
  snprintf(buf, n, "wrow_kernel<%d, %d, %s>", cf, nfy, (cf == 2 && nfy == 4) ? "true" : "false");
}
