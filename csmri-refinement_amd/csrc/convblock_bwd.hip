// csmri_convblock_fused_bwd: the backward pass of one RecNet conv block -- [ZeroPad(1) -> Conv3x3 -> LeakyReLU] x 2 ->
// ZeroPad(1) -> Conv3x3, channels 2 -> 32 -> 32 -> 2 (reference models/recnet.py:29-62; its backward is what
// `loss.backward()` of training/runner.py:154-178 runs through every block) -- as ONE kernel: bf16 operands, fp32
// accumulation.
//
// Per layer the unfused backward is a data-gradient conv, a weight-gradient reduction and an activation-derivative
// pass: six kernels per block that move the 32-channel tensors (64-byte pixels) through HBM ~2.7 GB per block at batch
// 64 (a1, a2 read twice each, dA2 / dA1 written and read twice).  Here a workgroup owns a 16 x 16 tile of the block's
// INPUT pixels and carries the whole chain through LDS, the mirror image of convblock_fwd_kernel:
//
//   G3  = dY on 22 x 22 (halo 3)                                            [global: 2 channels]
//   dA2 = conv3x3(G3, W3 flipped) * lrelu'(a2)   on 20 x 20                 [a2 from global, 20 x 20]
//   dA1 = conv3x3(dA2, W2 flipped) * lrelu'(a1)  on 18 x 18                 [a1 from global, 18 x 18]
//   dX  = conv3x3(dA1, W1 flipped)               on 16 x 16  -> global      [x  from global, 18 x 18: for dW1 only]
//   dW3 += a2^T G3, dW2 += a1^T dA2, dW1 += x^T dA1, db_l += sum dY_l       over the tile's central 16 x 16 pixels
//
// so x, a1, a2 and dY are read once (halo: 1.27 / 1.27 / 1.56 / 1.9 x) and dX is the only tensor written; the weight
// and bias gradients stay in registers over all tiles of the persistent workgroup and leave as ONE slab per layer and
// workgroup in csmri_wgrad's slab format (the existing slab reduction finishes them).
//
// All LDS images are plane-major in "pitch space" (convblock.hip): [8-channel plane][pixel q = row * 22 + col][16 B],
// plane stride = 64 mod 256 bytes.  The data-gradient stages read them exactly like the forward layers
// (ds_read_b128, 16 consecutive pixels of a plane per lane group: conflict-free at any tap shift); the weight-gradient
// products need pixel-contiguous fragments and read the SAME images with ds_read_b64_tr_b16 (a 4-pixel x 16-channel
// block per 16 lanes, the two planes of a 16-channel group 64 mod 256 bytes apart: conflict-free).
// Data-gradient weights: csmri_pack_weight mode 3 (flipped taps, roles swapped) of each layer, so every stage is a
// plain "same" correlation.  Activation pixels outside the image carry zero gradient (the forward zero-pads them).
#include <type_traits>
#include "common.h"

struct BBParams {
  const char* x; int xps;                    // bf16 [B,H,W,>=8]   block input (weight gradient of layer 1 only)
  const char* a1; int a1ps; const char* a2; int a2ps;   // bf16 [B,H,W,>=32] saved activations
  const char* gy; int gyps, gy_dt;           // dY of the block: fp32 [B,H,W,2] (gyps == 2) or [B,H,W,>=8] bf16 / fp32
  int B, H, W, tiles_x, tiles_y;
  const char* wd1; const char* wd2; const char* wd3; int kp1, kp2, kp3;   // mode-3 packs of layers 1, 2, 3
  float slope;
  char* dx; int dxps;                        // bf16 [B,H,W,>=8] or NULL (first block of the cascade)
  float* slab1; float* slab2; float* slab3;  // [Z][Cout_p][NK] + [Z][Cout_p] bias partial rows behind them
  int want_db;
  int x_split, dx_split;                     // CSMRI_BF16_SPLIT block input / input gradient
  unsigned long long* dbg;                   // CSMRI_DBG_STAMPS builds: per-wave phase stamps (want_db == 2: dx is the buffer)
};

#define BB_T 16
#define BB_PW 22
#define BB_G3PX 496                          // 22 x 22 + the reach of stage 1's last fragment (zero filled)
#define BB_A2PX 448                          // 20 x 22 = 440 positions (28 fragments)
#define BB_A1PX 448                          // 18 x 22 = 396 positions (25 fragments); 7 LDS-DMA pieces of 64 pixels
#define BB_XPX 448
#define BB_A2F 28
#define BB_A1F 25
#define BB_PS2 (BB_A2PX * 16 + 64)           // plane strides: 64 mod 256 (transposed reads of a plane pair: no conflict)
#define BB_PS1 (BB_A1PX * 16 + 64)
#define BB_OFF_G3 0
#define BB_OFF_X (BB_G3PX * 16)
#define BB_OFF_A2 (BB_OFF_X + BB_XPX * 16)
#define BB_OFF_A1 (BB_OFF_A2 + 4 * BB_PS2)
#define BB_OFF_D2 (BB_OFF_A1 + 4 * BB_PS1)
#define BB_OFF_D1 (BB_OFF_D2 + 4 * BB_PS2)
#define BB_OFF_W3 (BB_OFF_D1 + 4 * BB_PS1)   // stage-3 weights as per-lane fragments: 9 x 1 KiB
#define BB_OFF_W1 (BB_OFF_W3 + 9 * 1024)  // stage-1 weights likewise: 6 x 1 KiB
#define BB_LDS (BB_OFF_W1 + 6 * 1024)
#define BB_THREADS 512
#ifndef BB_KC_UNROLL
#define BB_KC_UNROLL 8
#endif
#define BB_STR2(x) #x
#define BB_STR(x) BB_STR2(x)
#define BB_UNROLL_KC _Pragma(BB_STR(unroll BB_KC_UNROLL))

typedef __attribute__((ext_vector_type(4))) short bb_s16x4_t;
struct bb_s16x8_pair { bb_s16x4_t lo, hi; };
__device__ __forceinline__ bf16x8_t bb_tr(const char* lo, const char* hi) {
  bb_s16x8_pair r;
  r.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bb_s16x4_t*)lo);
  r.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bb_s16x4_t*)hi);
  return __builtin_bit_cast(bf16x8_t, r);
}
__device__ __forceinline__ f32x4_t bb_unpack4(u32x2_t u) {
  return (f32x4_t){__uint_as_float(u[0] << 16), __uint_as_float(u[0] & 0xffff0000u), __uint_as_float(u[1] << 16),
                   __uint_as_float(u[1] & 0xffff0000u)};
}

__device__ __attribute__((aligned(16))) char bb_zero_page[16];
typedef __attribute__((address_space(1))) const void* bb_gptr_t;
typedef __attribute__((address_space(3))) void* bb_lptr_t;

__global__ __launch_bounds__(BB_THREADS, 1) void convblock_bwd_kernel(const BBParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* G3 = smem + BB_OFF_G3;
  char* XI = smem + BB_OFF_X;
  char* A2 = smem + BB_OFF_A2;
  char* A1 = smem + BB_OFF_A1;
  char* D2 = smem + BB_OFF_D2;
  char* D1 = smem + BB_OFF_D1;
  char* W3L = smem + BB_OFF_W3;
  char* W1L = smem + BB_OFF_W1;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const f32x4_t zero4 = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  // (static priority for the second-dispatched half of the waves -- MI355X_MICROARCH.md, two waves per SIMD, item 4 --
  // measured 1-3 % slower here: profiles/r04_negative_results.log)

  // ---- data-gradient weights: stage 2 (the heavy one) in registers for the whole launch, stages 1 and 3 as per-lane
  // fragments in LDS (all 33 fragments in registers spill at two waves per SIMD)
  u32x4_t w2f[2][9];
  if (wv >= 2) {
    for (int f = wv - 2; f < 6; f += 6)               // fragment f = i * 3 + s of stage 1
      *(u32x4_t*)(W1L + (f * 64 + lane) * 16) =
          *(const u32x4_t*)(p.wd3 + ((size_t)((f / 3) * 16 + r16) * p.kp3 + (f % 3) * 32 + g * 8) * 2);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int s = 0; s < 9; ++s)
      w2f[i][s] = *(const u32x4_t*)(p.wd2 + ((size_t)(i * 16 + r16) * p.kp2 + s * 32 + g * 8) * 2);
  if (wv < 2) {
    for (int s = wv; s < 9; s += 2)
      *(u32x4_t*)(W3L + (s * 64 + lane) * 16) = *(const u32x4_t*)(p.wd1 + ((size_t)r16 * p.kp1 + s * 32 + g * 8) * 2);
  }
  // zero the image tails that junk columns read (never written again): G3 pixels 484.., D2 440.., D1 396..
  for (int i = tid; i < (BB_G3PX - 484); i += BB_THREADS) *(u32x4_t*)(G3 + (484 + i) * 16) = (u32x4_t){0u, 0u, 0u, 0u};
  for (int i = tid; i < 4 * (BB_A2PX - 440); i += BB_THREADS) {
    const int k = i / (BB_A2PX - 440), q = 440 + i % (BB_A2PX - 440);
    *(u32x4_t*)(D2 + k * BB_PS2 + q * 16) = (u32x4_t){0u, 0u, 0u, 0u};
  }
  for (int i = tid; i < 4 * (BB_A1PX - 396); i += BB_THREADS) {
    const int k = i / (BB_A1PX - 396), q = 396 + i % (BB_A1PX - 396);
    *(u32x4_t*)(D1 + k * BB_PS1 + q * 16) = (u32x4_t){0u, 0u, 0u, 0u};
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int s = 0; s < 9; ++s) asm volatile("" : "+v"(w2f[i][s]));
  __syncthreads();

  // ---- weight-gradient accumulators (persist over the tiles of this workgroup) ----------------------------------
  // units: layer 3 and 2: (tap, 16-channel half of ci) = 18 units, wave wv owns units 7 - wv, 15 - wv, 23 - wv (< 18):
  // the waves that get a fourth data-gradient fragment in a phase (the low ones) get two units, the high ones three;
  //        layer 1: tap PAIRS (8 channels each) = 5 units, waves 0..4 own one each
  f32x4_t aw3[3], aw2[3][2], aw1[2];
#pragma unroll
  for (int a = 0; a < 3; ++a) { aw3[a] = zero4; aw2[a][0] = zero4; aw2[a][1] = zero4; }
  aw1[0] = zero4; aw1[1] = zero4;
  f32x4_t bs2[2] = {zero4, zero4};                               // bias gradient of layer 2: ones x dY products (wave 3)
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, (u32x4_t){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});
  float bs3[2] = {0.f, 0.f};

  // transposed-read lane constants: K chunk kc = tile rows 2kc, 2kc + 1; this lane addresses pixel klo = 8 * (lane>>4) +
  // ((lane>>2)&3) (and klo + 4) of the chunk and the 4-channel quad (lane & 3) of a 16-channel group
  const int tq = (lane >> 2) & 3, tp = lane & 3;
  const int klo = 8 * g + tq;
  const int trow = (klo >> 4) * BB_PW + (klo & 15);              // pixel offset of klo inside the chunk (tile coords)
  const int tquad = (tp & 1) * 8;                                // byte offset inside the 16-byte plane slot
  const int tplane = tp >> 1;                                    // which plane of the group's pair

  const int ntiles = p.B * p.tiles_x * p.tiles_y;
  const int tpi = p.tiles_x * p.tiles_y;
  // ---- input pipeline (LDS-DMA): a2 of tile t + 1 (and its dY, through registers: it needs the fp32 -> bf16
  // conversion) stream in as soon as phase 1 of tile t has consumed them; a1 and x of tile t under its own phase 1.
  // A DMA piece = 64 consecutive pixels of one plane (1 KiB); out-of-image pixels and the images' tails come from a
  // zero page.
  // piece i of an image = plane i / 7, pixels (i % 7) * 64 + lane; wave wv issues pieces wv, wv + 8, wv + 16, wv + 24.
  // The piece's pixel (py, px) inside the region is a per-lane constant: decoded once, not per tile.
  // Wave c (0..6) issues chunk c (pixels 64 c + lane) of every plane of an image -- the four planes of a 32-channel
  // image in iterations 0..3, the one plane of the block input with iteration 0 --, wave 7 none: the piece's pixel is
  // ONE per-lane constant for all of a wave's pieces (two registers instead of nine).
  const int dpp = (wv < 7 ? wv : 6) * 64 + lane;
  const int dpy = (dpp * 2979) >> 16, dpx = dpp - dpy * BB_PW;
  const int dpix = dpy * p.W + dpx;                   // pixel offset inside a region (regions are 22 wide)
  // A DMA piece is issued from INSIDE the data-gradient loops, one per fragment iteration: the stamped kernel
  // (tools/stamp_convblock_bwd.py) spent 4,400 of its 18,500 cycles per tile in two phases in which all eight waves did
  // nothing but compute addresses and issue their pieces.  Per tile and image the plan (base pointer, border flag) is
  // computed once; a piece's offset inside the region is a per-lane constant computed once per launch.
  struct DmaPlan { const char* base; int yb, xb; bool inner; };
  auto dma_plan = [&](int tl, const char* src, int sps, int oy, int ox, int rows) {
    DmaPlan pl;
    const int b_ = tl / tpi, r_ = tl - b_ * tpi, ty_ = r_ / p.tiles_x, tx_ = r_ - ty_ * p.tiles_x;
    pl.yb = ty_ * BB_T - oy; pl.xb = tx_ * BB_T - ox;
    pl.inner = pl.yb >= 0 && pl.xb >= 0 && pl.yb + rows <= p.H && pl.xb + BB_PW <= p.W;     // workgroup-uniform
    pl.base = src + (((long long)b_ * p.H + pl.yb) * p.W + pl.xb) * (long long)sps * 2;
    return pl;
  };
  auto dma_piece = [&](const DmaPlan& pl, int plane, int sps, char* img, int pstride, int rows) {
    if (wv >= 7) return;                               // wave-uniform
    bool ok = dpy < rows;
    if (!pl.inner) ok = ok && (unsigned)(pl.yb + dpy) < (unsigned)p.H && (unsigned)(pl.xb + dpx) < (unsigned)p.W;
    const char* g_ = ok ? pl.base + (unsigned)((dpix * sps + plane * 8) * 2) : bb_zero_page;
    __builtin_amdgcn_global_load_lds((bb_gptr_t)g_, (bb_lptr_t)(img + plane * pstride + wv * 1024), 16, 0, 0);
  };
  // this thread's dY pixel of the NEXT tile, loaded at the START of the current tile and kept as loaded (the conversion to
  // bf16 -- i.e. the wait for the load -- happens where the value is stored into its image, a whole tile later)
  u32x2_t vg3r = (u32x2_t){0u, 0u};                  // fp32 pair, or the bf16 pair in [0]
  auto load_g3 = [&](int tl) {
    vg3r = (u32x2_t){0u, 0u};
    if (tid >= 484 || tl >= ntiles) return;
    const int b_ = tl / tpi, r_ = tl - b_ * tpi, ty_ = r_ / p.tiles_x, tx_ = r_ - ty_ * p.tiles_x;
    const int py = tid / BB_PW, px = tid - py * BB_PW;
    const int y = ty_ * BB_T - 3 + py, x = tx_ * BB_T - 3 + px;
    if ((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) {
      const size_t pix = ((size_t)b_ * p.H + y) * p.W + x;
      if (p.gy_dt == CSMRI_F32) vg3r = *(const u32x2_t*)(p.gy + pix * (size_t)p.gyps * 4);
      else vg3r[0] = *(const unsigned*)(p.gy + pix * (size_t)p.gyps * 2);
    }
  };
  const int gpy = tid / BB_PW, gpx = tid - gpy * BB_PW;
  const bool g3_central = tid < 484 && gpy >= 3 && gpy < 3 + BB_T && gpx >= 3 && gpx < 3 + BB_T;
  if ((int)blockIdx.x < ntiles) {
    load_g3(blockIdx.x);
    const DmaPlan pl0 = dma_plan(blockIdx.x, p.a2, p.a2ps, 2, 2, 20);
#pragma unroll
    for (int k = 0; k < 4; ++k) dma_piece(pl0, k, p.a2ps, A2, BB_PS2, 20);
  }
#ifdef CSMRI_DBG_STAMPS
  unsigned long long ph[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, last_t;
#define BB_STAMP(i) do { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
    ph[i] += t_ - last_t; last_t = t_; } while (0)
  { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last_t) :: "memory"); }
#else
#define BB_STAMP(i) do {} while (0)
#endif
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int t = tile;
    const int b = t / tpi;
    t -= b * tpi;
    const int tyi = t / p.tiles_x, txi = t - tyi * p.tiles_x;
    const int y0 = tyi * BB_T, x0 = txi * BB_T;
    const bool border = y0 < 2 || x0 < 2 || y0 + BB_T + 2 > p.H || x0 + BB_T + 2 > p.W;
    const int next = tile + gridDim.x;
    const bool has_next = next < ntiles;

    // ---- this tile's dY pixel into its image (bias gradient of layer 3 on the bf16 values the products see); all of
    // this tile's DMA pieces were issued during the previous tile: wait for mine, the barrier publishes everybody's
    if (tid < 484) {
      const unsigned vg3 = p.gy_dt == CSMRI_F32
          ? __builtin_bit_cast(unsigned, __builtin_convertvector(__builtin_bit_cast(f32x2_t, vg3r), bf16x2_t)) : vg3r[0];
      *(u32x4_t*)(G3 + tid * 16) = (u32x4_t){vg3, 0u, 0u, 0u};
      if (g3_central) { bs3[0] += __uint_as_float(vg3 << 16); bs3[1] += __uint_as_float(vg3 & 0xffff0000u); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BB_STAMP(0);
    __syncthreads();
    BB_STAMP(1);
    load_g3(next);                               // (a register prefetch: the dY image itself is not touched)
    // a1 and x of this tile stream in under phase 1 (their images are free since the previous tile's phases 2 / 3), one
    // piece per fragment iteration of stage 1
    const DmaPlan plA1 = dma_plan(tile, p.a1, p.a1ps, 1, 1, 18);
    const DmaPlan plX = dma_plan(tile, p.x, p.xps, 1, 1, 18);
    BB_STAMP(2);

    // ---- phase 1: dA2 = conv(G3, W3 flipped) * lrelu'(a2) on 20 rows (origin y0 - 2, x0 - 2) ---------------------
    // K step s = filter row, lane group g = tap column (column 3 meets zero weights)
    auto stage1 = [&](auto border_tag) {
      constexpr bool BORDER = decltype(border_tag)::value;
      int it = 0;                                      // (a ROLLED loop: unrolled, its four bodies spill)
      dma_piece(plX, 0, p.xps, XI, 0, 18);
#pragma nounroll
      for (int j = wv; j < BB_A2F; j += 8, ++it) {
        dma_piece(plA1, it, p.a1ps, A1, BB_PS1, 18);
        bf16x8_t xf[3];
        const char* src = G3 + (j * 16 + r16 + g) * 16;
#pragma unroll
        for (int s = 0; s < 3; ++s) xf[s] = *(const bf16x8_t*)(src + s * BB_PW * 16);
        f32x4_t acc0 = zero4, acc1 = zero4;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8_t*)(W1L + (s * 64 + lane) * 16), xf[s], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8_t*)(W1L + ((3 + s) * 64 + lane) * 16), xf[s], acc1, 0, 0, 0);
        }
        const int q = j * 16 + r16;
        float one = 1.f, sl = p.slope;                               // derivative factors of an in-image pixel
        if (BORDER) {
          const int qy = (q * 2979) >> 16, qx = q - qy * BB_PW;      // q / 22 for q < 448
          if (!((unsigned)(y0 - 2 + qy) < (unsigned)p.H && (unsigned)(x0 - 2 + qx) < (unsigned)p.W)) { one = 0.f; sl = 0.f; }
        }
        const int so0 = (g >> 1) * BB_PS2 + q * 16 + (g & 1) * 8, so1 = so0 + 2 * BB_PS2;
        const f32x4_t m0 = bb_unpack4(*(const u32x2_t*)(A2 + so0)), m1 = bb_unpack4(*(const u32x2_t*)(A2 + so1));
        f32x4_t v0, v1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v0[r] = acc0[r] * (m0[r] > 0.f ? one : sl);
          v1[r] = acc1[r] * (m1[r] > 0.f ? one : sl);
        }
        *(u32x2_t*)(D2 + so0) = pack4_bf16(v0);
        *(u32x2_t*)(D2 + so1) = pack4_bf16(v1);
      }
#pragma nounroll
      for (; it < 4; ++it) dma_piece(plA1, it, p.a1ps, A1, BB_PS1, 18);     // waves with three fragments
    };
    if (border) stage1(std::true_type{}); else stage1(std::false_type{});
    BB_STAMP(3);
    // ---- ... and the weight gradient of layer 3: x = a2 (patch origin = image origin + (1, 1)), dY = G3 centre ------
    {
      const char* yb = G3 + (3 * BB_PW + 3 + trow) * 16 + tquad;            // one plane: both halves of the pair read it
      const char* xb[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const int unit = min(7 - wv + 8 * a, 17), tap = unit % 9, cf = unit / 9, ty = tap / 3, tx = tap - ty * 3;
        xb[a] = A2 + (cf * 2 + tplane) * BB_PS2 + ((1 + ty) * BB_PW + 1 + tx + trow) * 16 + tquad;
      }
BB_UNROLL_KC
      for (int kc = 0; kc < 8; ++kc) {
        const int ko = kc * 2 * BB_PW * 16;
        const bf16x8_t yf = bb_tr(yb + ko, yb + ko + 64);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          if (7 - wv + 8 * a >= 18) break;                                      // wave-uniform
          aw3[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb_tr(xb[a] + ko, xb[a] + ko + 64), yf, aw3[a], 0, 0, 0);
        }
      }
    }
    BB_STAMP(4);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();               // dA2 complete, a1 / x landed; dY and a2 images are free
    BB_STAMP(5);
    // a2 of the next tile streams in under phase 2, one piece per fragment iteration (its image is free from here on)
    const DmaPlan plA2 = dma_plan(has_next ? next : tile, p.a2, p.a2ps, 2, 2, 20);
    BB_STAMP(6);

    // ---- phase 2: dA1 = conv(dA2, W2 flipped) * lrelu'(a1) on 18 rows (origin y0 - 1, x0 - 1) --------------------
    auto stage2 = [&](auto border_tag) {
      constexpr bool BORDER = decltype(border_tag)::value;
      int it = 0;
#pragma nounroll
      for (int j = wv; j < BB_A1F; j += 8, ++it) {
        if (has_next) dma_piece(plA2, it, p.a2ps, A2, BB_PS2, 20);
        bf16x8_t xf[9];
        const char* src = D2 + g * BB_PS2 + (j * 16 + r16) * 16;
#pragma unroll
        for (int s = 0; s < 9; ++s) xf[s] = *(const bf16x8_t*)(src + ((s / 3) * BB_PW + (s % 3)) * 16);
        f32x4_t acc0 = zero4, acc1 = zero4;
#pragma unroll
        for (int s = 0; s < 9; ++s) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w2f[0][s]), xf[s], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w2f[1][s]), xf[s], acc1, 0, 0, 0);
        }
        const int q = j * 16 + r16;
        float one = 1.f, sl = p.slope;
        if (BORDER) {
          const int qy = (q * 2979) >> 16, qx = q - qy * BB_PW;
          if (!((unsigned)(y0 - 1 + qy) < (unsigned)p.H && (unsigned)(x0 - 1 + qx) < (unsigned)p.W)) { one = 0.f; sl = 0.f; }
        }
        const int so0 = (g >> 1) * BB_PS1 + q * 16 + (g & 1) * 8, so1 = so0 + 2 * BB_PS1;
        const f32x4_t m0 = bb_unpack4(*(const u32x2_t*)(A1 + so0)), m1 = bb_unpack4(*(const u32x2_t*)(A1 + so1));
        f32x4_t v0, v1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v0[r] = acc0[r] * (m0[r] > 0.f ? one : sl);
          v1[r] = acc1[r] * (m1[r] > 0.f ? one : sl);
        }
        *(u32x2_t*)(D1 + so0) = pack4_bf16(v0);
        *(u32x2_t*)(D1 + so1) = pack4_bf16(v1);
      }
      if (has_next) {
#pragma nounroll
        for (; it < 4; ++it) dma_piece(plA2, it, p.a2ps, A2, BB_PS2, 20);
      }
    };
    if (border) stage2(std::true_type{}); else stage2(std::false_type{});
    BB_STAMP(7);
    // ---- ... and the weight gradient of layer 2: x = a1 (patch origin = image origin), dY = dA2 centre (2, 2).  The dY
    // fragments of a K chunk are read once and shared by this wave's units (3 transposed reads per MFMA otherwise:
    // past what the LDS delivers next to the matrix pipe)
    {
      const char* yb = D2 + tplane * BB_PS2 + (2 * BB_PW + 2 + trow) * 16 + tquad;
      const char* xb[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const int unit = min(7 - wv + 8 * a, 17), tap = unit % 9, cf = unit / 9, ty = tap / 3, tx = tap - ty * 3;
        xb[a] = A1 + (cf * 2 + tplane) * BB_PS1 + (ty * BB_PW + tx + trow) * 16 + tquad;
      }
BB_UNROLL_KC
      for (int kc = 0; kc < 8; ++kc) {
        const int ko = kc * 2 * BB_PW * 16;
        const bf16x8_t yf0 = bb_tr(yb + ko, yb + ko + 64), yf1 = bb_tr(yb + 2 * BB_PS2 + ko, yb + 2 * BB_PS2 + ko + 64);
        if (wv == 3) {                                                        // bias gradient of layer 2 (row 0 of ones x dY)
          bs2[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, yf0, bs2[0], 0, 0, 0);
          bs2[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, yf1, bs2[1], 0, 0, 0);
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          if (7 - wv + 8 * a >= 18) break;                                      // wave-uniform
          const bf16x8_t xf = bb_tr(xb[a] + ko, xb[a] + ko + 64);
          aw2[a][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, yf0, aw2[a][0], 0, 0, 0);
          aw2[a][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, yf1, aw2[a][1], 0, 0, 0);
        }
      }
    }
    BB_STAMP(8);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();               // dA1 complete; the a1 image is free
    BB_STAMP(9);

    // ---- phase 3: dX = conv(dA1, W1 flipped) on the 16 x 16 tile: fragment j = output row j --------------------------
    if (p.dx) {
      for (int j = wv; j < BB_T; j += 8) {
        bf16x8_t xf[9];
        const char* src = D1 + g * BB_PS1 + (j * BB_PW + r16) * 16;
#pragma unroll
        for (int s = 0; s < 9; ++s) xf[s] = *(const bf16x8_t*)(src + ((s / 3) * BB_PW + (s % 3)) * 16);
        f32x4_t acc = zero4;
#pragma unroll
        for (int s = 0; s < 9; ++s)
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8_t*)(W3L + (s * 64 + lane) * 16), xf[s], acc, 0, 0, 0);
        const int y = y0 + j, x = x0 + r16;
        if (p.dx_split && g == 0) {
          // channels 2,3 (zero weight rows: exact zeros so far) carry what the bf16 rounding of channels 0,1 drops
          acc[2] = acc[0] - bf16_bits_to_f32(f32_to_bf16_bits(acc[0]));
          acc[3] = acc[1] - bf16_bits_to_f32(f32_to_bf16_bits(acc[1]));
        }
        if (g < 2 && y < p.H && x < p.W)
          *(u32x2_t*)(p.dx + ((((size_t)b * p.H + y) * p.W + x) * (size_t)p.dxps + 4 * g) * 2) = pack4_bf16(acc);
      }
    }
    BB_STAMP(10);
    // ---- ... and the weight gradient of layer 1: x = block input, 8 channels: a fragment is a PAIR of taps x 8
    // channels (lanes tp = 0, 1 address the first tap, tp = 2, 3 the second; the last pair repeats tap 8, its second
    // half is never written); dY = dA1 centre (1, 1)
    if (wv < 6) {                                 // (wave 5: the bias gradient of layer 1 instead of a tap pair)
      const int tap = min(2 * wv + tplane, 8), ty = tap / 3, tx = tap - ty * 3;
      const char* xb = XI + (ty * BB_PW + tx + trow) * 16 + tquad;
      const char* yb = D1 + tplane * BB_PS1 + (1 * BB_PW + 1 + trow) * 16 + tquad;
BB_UNROLL_KC
      for (int kc = 0; kc < 8; ++kc) {
        const int ko = kc * 2 * BB_PW * 16;
        const bf16x8_t xf = wv == 5 ? ones : bb_tr(xb + ko, xb + ko + 64);
        const bf16x8_t yf0 = bb_tr(yb + ko, yb + ko + 64), yf1 = bb_tr(yb + 2 * BB_PS1 + ko, yb + 2 * BB_PS1 + ko + 64);
        aw1[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, yf0, aw1[0], 0, 0, 0);
        aw1[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, yf1, aw1[1], 0, 0, 0);
      }
    }
    BB_STAMP(11);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();               // the x image is free (and dA2 / dA1 may be overwritten)
    BB_STAMP(12);
  }
#ifdef CSMRI_DBG_STAMPS
  if (p.dbg && lane == 0) {
#pragma unroll
    for (int i = 0; i < 13; ++i) p.dbg[((size_t)blockIdx.x * 8 + wv) * 16 + i] = ph[i];
  }
#endif

  // ---- one slab per layer and workgroup: [Cout_p][NK], NK index = tap * Cin_p + ci; D row = 4g + reg, column = r16 ---
  const int z = blockIdx.x, Z = gridDim.x;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int unit = 7 - wv + 8 * a;
    if (unit >= 18) break;
    const int tap = unit % 9, cf = unit / 9;
    if (r16 < 8) *(f32x4_t*)(p.slab3 + ((size_t)z * 8 + r16) * 288 + tap * 32 + cf * 16 + g * 4) = aw3[a];
#pragma unroll
    for (int n = 0; n < 2; ++n)
      *(f32x4_t*)(p.slab2 + ((size_t)z * 32 + n * 16 + r16) * 288 + tap * 32 + cf * 16 + g * 4) = aw2[a][n];
  }
  if (wv < 5) {
    const int tap = 2 * wv + (g >> 1);
    if (tap < 9) {
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        f32x4_t v = aw1[n];
        if (p.x_split && (g & 1) == 0) {          // input channels 0..3 = (hi, lo) of the two real ones: one gradient
          v[0] += v[2]; v[1] += v[3]; v[2] = 0.f; v[3] = 0.f;
        }
        *(f32x4_t*)(p.slab1 + ((size_t)z * 32 + n * 16 + r16) * 72 + tap * 8 + (g & 1) * 4) = v;
      }
    }
  }
  if (p.want_db) {
    // bias partial rows behind the slabs.  Layers 2 and 1: row 0 of the ones products (lanes g == 0, register 0, column
    // r16 = channel within the 16-channel fragment); layer 3: per-thread sums of the staging pass, waves through LDS
    if (wv == 3 && g == 0) {
      p.slab2[(size_t)Z * 32 * 288 + (size_t)z * 32 + r16] = bs2[0][0];
      p.slab2[(size_t)Z * 32 * 288 + (size_t)z * 32 + 16 + r16] = bs2[1][0];
    }
    if (wv == 5 && g == 0) {
      p.slab1[(size_t)Z * 32 * 72 + (size_t)z * 32 + r16] = aw1[0][0];
      p.slab1[(size_t)Z * 32 * 72 + (size_t)z * 32 + 16 + r16] = aw1[1][0];
    }
    __syncthreads();
    float* red = (float*)smem;
    float v30 = bs3[0], v31 = bs3[1];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { v30 += __shfl_xor(v30, o); v31 += __shfl_xor(v31, o); }
    if (lane == 0) { red[wv * 2] = v30; red[wv * 2 + 1] = v31; }
    __syncthreads();
    if (tid < 2) {
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) sum += red[w * 2 + tid];
      p.slab3[(size_t)Z * 8 * 288 + (size_t)z * 8 + tid] = sum;
    }
  }
}

extern "C" int csmri_convblock_fused_bwd(const csmri_convblock_bwd_desc* d, void* stream) {
  CSMRI_CHECK_ARG(d && d->x && d->act[0] && d->act[1] && d->gy && d->wd[0] && d->wd[1] && d->wd[2] && d->slab[0] &&
                  d->slab[1] && d->slab[2]);
  if (d->dtype != CSMRI_BF16 || d->num_convs != 3 || d->num_filters != 32 || d->kernel_size != 3 || d->num_inputs != 2 ||
      d->num_outputs != 2 || d->border != CSMRI_BORDER_ZERO) return CSMRI_E_UNSUPPORTED;
  CSMRI_CHECK_ARG(d->B > 0 && d->H > 0 && d->W > 0 && d->splits > 0 && d->splits <= 2048);
  CSMRI_CHECK_ARG(d->x_pix_stride >= 8 && d->x_pix_stride % 8 == 0 && d->act_pix_stride[0] >= 32 &&
                  d->act_pix_stride[0] % 8 == 0 && d->act_pix_stride[1] >= 32 && d->act_pix_stride[1] % 8 == 0);
  CSMRI_CHECK_ARG((d->gy_pix_stride == 2 && d->gy_dtype == CSMRI_F32) || (d->gy_pix_stride >= 8 && d->gy_pix_stride % 2 == 0));
  CSMRI_CHECK_ARG(d->gy_dtype == CSMRI_F32 || d->gy_dtype == CSMRI_BF16);
  CSMRI_CHECK_ARG(d->Kp[0] >= 288 && d->Kp[1] >= 288 && d->Kp[2] >= 96);
  if ((long long)d->H * d->W * (d->act_pix_stride[0] > d->act_pix_stride[1] ? d->act_pix_stride[0] : d->act_pix_stride[1]) * 2 >= (1ll << 31)) return CSMRI_E_UNSUPPORTED;   // 32-bit offsets inside an image
  if (d->dx) CSMRI_CHECK_ARG(d->dx_pix_stride >= 8 && d->dx_pix_stride % 4 == 0);
  if (((uintptr_t)d->x | (uintptr_t)d->act[0] | (uintptr_t)d->act[1] | (uintptr_t)d->wd[0] | (uintptr_t)d->wd[1] |
       (uintptr_t)d->wd[2] | (uintptr_t)d->dx | (uintptr_t)d->slab[0] | (uintptr_t)d->slab[1] | (uintptr_t)d->slab[2]) & 15)
    return CSMRI_E_ALIGN;
  if ((uintptr_t)d->gy & 7) return CSMRI_E_ALIGN;
  BBParams p;
  p.x = (const char*)d->x; p.xps = d->x_pix_stride;
  p.a1 = (const char*)d->act[0]; p.a1ps = d->act_pix_stride[0]; p.a2 = (const char*)d->act[1]; p.a2ps = d->act_pix_stride[1];
  p.gy = (const char*)d->gy; p.gyps = d->gy_pix_stride; p.gy_dt = d->gy_dtype;
  p.B = d->B; p.H = d->H; p.W = d->W;
  p.tiles_x = (d->W + BB_T - 1) / BB_T; p.tiles_y = (d->H + BB_T - 1) / BB_T;
  p.wd1 = (const char*)d->wd[0]; p.wd2 = (const char*)d->wd[1]; p.wd3 = (const char*)d->wd[2];
  p.kp1 = d->Kp[0]; p.kp2 = d->Kp[1]; p.kp3 = d->Kp[2];
  p.slope = d->slope;
  p.dx = (char*)d->dx; p.dxps = d->dx_pix_stride;
  p.dbg = nullptr;
#ifdef CSMRI_DBG_STAMPS
  if (d->want_db == 2) { p.dbg = (unsigned long long*)d->dx; p.dx = nullptr; }
#endif
  p.slab1 = d->slab[0]; p.slab2 = d->slab[1]; p.slab3 = d->slab[2];
  p.want_db = d->want_db;
  p.x_split = d->x_split; p.dx_split = d->dx_split;
  const long long tiles = (long long)d->B * p.tiles_x * p.tiles_y;
  if (tiles >= (1ll << 31)) return CSMRI_E_UNSUPPORTED;
  CSMRI_CHECK_ARG(d->splits <= tiles);
  CSMRI_SET_MAX_LDS(convblock_bwd_kernel, BB_LDS);
  hipLaunchKernelGGL(convblock_bwd_kernel, dim3(d->splits), dim3(BB_THREADS), BB_LDS, (hipStream_t)stream, p);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

// workgroups (= slabs per layer) csmri_convblock_fused_bwd should be launched with for this problem
extern "C" int csmri_convblock_fused_bwd_splits(int B, int H, int W) {
  const long long tiles = (long long)B * ((H + BB_T - 1) / BB_T) * ((W + BB_T - 1) / BB_T);
  long long z = 256;                                  // one persistent workgroup per CU (124 KiB of LDS each)
  if (z > tiles) z = tiles;
  return (int)(z < 1 ? 1 : z);
}
