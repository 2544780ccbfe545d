// csmri_convblock_fused_fwd: one RecNet conv block -- [ZeroPad(1) -> Conv3x3 -> LeakyReLU] x 2 -> ZeroPad(1) ->
// Conv3x3, channels 2 -> 32 -> 32 -> 2 (reference models/recnet.py:29-62) -- as ONE kernel, bf16 operands, fp32
// accumulate.
//
// The three layers as separate launches are HBM passes over 64-byte pixels (the 32-channel intermediates: 2 x 268 MB
// written and read back per block at batch 64).  Here a workgroup owns a 16 x 16 output tile and carries the whole
// block through LDS: the 22 x 22 input patch (halo 3), the first activation on 20 x 20, the second on 18 x 18, the
// output on 16 x 16; the halo rings are recomputed (1.56x / 1.27x of the layer's FLOPs -- MFMA time this path has to
// spare) so that no intermediate is ever re-read from HBM.  In training the centre 16 x 16 of both activations is
// also written out (the existing data-gradient / weight-gradient kernels consume them); a frozen or evaluating
// block writes nothing but its output.
//
// Per layer the GEMM is D[cout][position] = sum_k W[cout][k] * X[position][k] on v_mfma_f32_16x16x32_bf16, positions
// in fragments of 16 consecutive pixels of the image in "pitch space" (see CB_PW below), K = (tap, channel):
//   layer 1: 8 padded input channels, filter rows padded to 4 taps -> one 32-wide K step per filter row (lane group
//            g reads tap column g of the patch; column 3 meets zero weights);
//   layer 2/3: one K step per tap = the 32 channels of the tap-shifted pixel.
// LDS images are plane-major -- [8-channel plane][pixel][16 B] -- so the 16 lanes of a group read 16 consecutive
// 16-byte slots (conflict-free ds_read_b128) at any tap shift.  The weights (6 + 18 + 9 fragments of 16 B per lane)
// are loaded once per workgroup straight into registers from the packed forward weights (csmri_pack_weight mode 0).
// Activation pixels that lie outside the image are stored as ZERO, which is what the next layer's zero padding
// reads in the reference.  Results are bit-identical to the three-launch path's (same K order per output element).
#include <type_traits>
#include "common.h"

struct CBParams {
  const char* x; int xps;                 // bf16 [B,H,W,>=8]
  int B, H, W, tiles_x, tiles_y;
  const char* w1; const char* w2; const char* w3; int kp1, kp2, kp3;
  const float* b1; const float* b2; const float* b3;
  float slope;
  char* a1; int a1ps; char* a2; int a2ps;   // bf16 [B,H,W,32] (training) or NULL
  char* out; int out_dt, ops;               // [B,H,W,8]
  int x_split;                              // x is CSMRI_BF16_SPLIT (hi in channels 0,1, lo in 2,3)
};

#define CB_T 16
// One row pitch for all three LDS images ("pitch space"): an output position q of a layer is the pixel index q of
// its image AND of its source image shifted by the tap, so the 16 positions of a fragment are always 16
// consecutive 16-byte slots of a plane -- no division per position, tap offsets are immediates, and no bank
// conflicts (fragments that wrap around a narrower region's row end read two runs that collide: 45 % of the LDS
// cycles in the first version).  The price: the columns between a region's width and the pitch are computed too
// (junk that only ever feeds junk): 28 + 25 fragments instead of 25 + 21 in layers 1 and 2.
#define CB_PW 22                             // pitch = width of the input patch (16 + 2 * 3)
#define CB_P0PX 496                          // 22 x 22 patch + the reach of layer 1's last fragment (447 + 2*22 + 3)
#define CB_A1F 28                            // layer-1 fragments: 20 rows x 22 = 440 positions
#define CB_A1PX (CB_A1F * 16)                // 448 pixels per plane (7168 B = 28 x 256: planes bank-aligned)
#define CB_A2F 25                            // layer-2 fragments: 18 rows x 22 = 396 positions
#define CB_A2PX (CB_A2F * 16)                // 400 pixels per plane (6400 B = 25 x 256)
#define CB_P0_BYTES (CB_P0PX * 16)
#define CB_A1_BYTES (4 * CB_A1PX * 16)
#define CB_A2_BYTES (4 * CB_A2PX * 16)
#define CB_LDS (CB_P0_BYTES + CB_A1_BYTES + CB_A2_BYTES)

__device__ __forceinline__ u32x2_t cb_pack4(f32x4_t v) { return pack4_bf16(v); }

// LeakyReLU as max(v, slope*v) (0 <= slope <= 1: exactly v or slope*v, the reference's v*slope rounding); the
// in-image flag is only applied on tiles that touch the image border (wave-uniform branch)
template <bool BORDER>
__device__ __forceinline__ f32x4_t cb_act(f32x4_t v, float slope, float keep) {
  f32x4_t o;
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = fmaxf(v[r], slope * v[r]);
  if (BORDER) {
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] *= keep;
  }
  return o;
}
// channels 4g..4g+3 of fragment 0 -> plane g>>1; of fragment 1 (16 + 4g ..) -> plane 2 + (g>>1)
__device__ __forceinline__ void cb_store_act(char* A, int plane_px, int q, int g, f32x4_t v0, f32x4_t v1) {
  *(u32x2_t*)(A + ((g >> 1) * plane_px + q) * 16 + (g & 1) * 8) = cb_pack4(v0);
  *(u32x2_t*)(A + ((2 + (g >> 1)) * plane_px + q) * 16 + (g & 1) * 8) = cb_pack4(v1);
}

template <bool SAVE>
__global__ __launch_bounds__(256, 2) void convblock_fwd_kernel(const CBParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* P0 = smem;
  char* A1 = smem + CB_P0_BYTES;
  char* A2 = A1 + CB_A1_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  // ---- weights -> registers, ONCE per (persistent) workgroup: 33 fragments of 16 B per lane = 132 KB per workgroup
  // through L2; reloading them per tile made the kernel L2-bound (every CU pulling the same 33 KB), 63 us per block
  // at batch 8 where the loop below takes a third of that ---------------------------------------------------------
  u32x4_t w1f[2][3], w2f[2][9], w3f[9];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int s = 0; s < 3; ++s)
      w1f[i][s] = *(const u32x4_t*)(p.w1 + ((size_t)(i * 16 + r16) * p.kp1 + s * 32 + g * 8) * 2);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int s = 0; s < 9; ++s)
      w2f[i][s] = *(const u32x4_t*)(p.w2 + ((size_t)(i * 16 + r16) * p.kp2 + s * 32 + g * 8) * 2);
#pragma unroll
  for (int s = 0; s < 9; ++s) w3f[s] = *(const u32x4_t*)(p.w3 + ((size_t)r16 * p.kp3 + s * 32 + g * 8) * 2);

  if (p.x_split) {
    // split input (channels 0,1 = hi, 2,3 = lo of the same two real channels): repeat the weights of channels 0,1 on
    // channels 2,3 -- a lane's 16 bytes are the 8 channels of one tap -- so that layer 1 multiplies hi + lo
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int s = 0; s < 3; ++s) w1f[i][s][1] = w1f[i][s][0];
  }
  const f32x4_t zero4 = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  f32x4_t bias10 = *(const f32x4_t*)(p.b1 + 4 * g), bias11 = *(const f32x4_t*)(p.b1 + 16 + 4 * g);
  f32x4_t bias20 = *(const f32x4_t*)(p.b2 + 4 * g), bias21 = *(const f32x4_t*)(p.b2 + 16 + 4 * g);
  f32x4_t bias3 = g < 2 ? *(const f32x4_t*)(p.b3 + 4 * g) : zero4;
  // The loads above are complete from here on, and the compiler is told so (each register passes through an empty
  // asm): otherwise its in-loop `s_waitcnt vmcnt` for "maybe still loading" weights would also wait for the next
  // tile's patch prefetch and for the previous tile's stores.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int s = 0; s < 3; ++s) asm volatile("" : "+v"(w1f[i][s]));
#pragma unroll
    for (int s = 0; s < 9; ++s) asm volatile("" : "+v"(w2f[i][s]));
  }
#pragma unroll
  for (int s = 0; s < 9; ++s) asm volatile("" : "+v"(w3f[s]));
  asm volatile("" : "+v"(bias10), "+v"(bias11), "+v"(bias20), "+v"(bias21), "+v"(bias3));
#ifdef CSMRI_DBG_STAMPS
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_t;
#define CB_STAMP(i) do { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
    ph[i] += t_ - last_t; last_t = t_; } while (0)
  { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last_t) :: "memory"); }
#else
#define CB_STAMP(i) do {} while (0)
#endif
  const int ntiles = p.B * p.tiles_x * p.tiles_y;
  // this thread's two pixels of a tile's input patch (halo 3; zero outside the image and in the tail pixels that
  // layer 1's junk columns reach), loaded one tile AHEAD: the global latency runs under the previous tile's layers
  const int pp0 = tid, pp1 = tid + 256;
  const int py0 = pp0 / CB_PW, px0 = pp0 - py0 * CB_PW, py1 = pp1 / CB_PW, px1 = pp1 - py1 * CB_PW;
  auto load_patch = [&](int tl, u32x4_t& v0, u32x4_t& v1) {
    v0 = (u32x4_t){0u, 0u, 0u, 0u}; v1 = v0;
    if (tl >= ntiles) return;
    int t_ = tl;
    const int b_ = t_ / (p.tiles_x * p.tiles_y);
    t_ -= b_ * p.tiles_x * p.tiles_y;
    const int ty_ = t_ / p.tiles_x, tx_ = t_ - ty_ * p.tiles_x;
    const int ya = ty_ * CB_T - 3 + py0, xa = tx_ * CB_T - 3 + px0, yb = ty_ * CB_T - 3 + py1, xb = tx_ * CB_T - 3 + px1;
    if ((unsigned)ya < (unsigned)p.H && (unsigned)xa < (unsigned)p.W)
      v0 = *(const u32x4_t*)(p.x + ((size_t)(b_ * p.H + ya) * p.W + xa) * (size_t)p.xps * 2);
    if (pp1 < CB_PW * CB_PW && (unsigned)yb < (unsigned)p.H && (unsigned)xb < (unsigned)p.W)
      v1 = *(const u32x4_t*)(p.x + ((size_t)(b_ * p.H + yb) * p.W + xb) * (size_t)p.xps * 2);
  };
  u32x4_t pre0, pre1;
  load_patch(blockIdx.x, pre0, pre1);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
  int t = tile;
  const int b = t / (p.tiles_x * p.tiles_y);
  t -= b * p.tiles_x * p.tiles_y;
  const int tyi = t / p.tiles_x, txi = t - tyi * p.tiles_x;
  const int y0 = tyi * CB_T, x0 = txi * CB_T;
  // does any pixel of the 20 x 20 region lie outside the image?  (tile index is workgroup-uniform)
  const bool border = y0 < 2 || x0 < 2 || y0 + CB_T + 2 > p.H || x0 + CB_T + 2 > p.W;

  // ---- input patch: the prefetched pixels into LDS, then the NEXT tile's loads go out ------------------------
  *(u32x4_t*)(P0 + pp0 * 16) = pre0;
  if (pp1 < CB_P0PX) *(u32x4_t*)(P0 + pp1 * 16) = pre1;
  CB_STAMP(0);
  __syncthreads();
  load_patch(tile + gridDim.x, pre0, pre1);
  CB_STAMP(1);

  // ---- layer 1: 8 -> 32 on 20 rows (image origin y0-2, x0-2); K step s = filter row, lane group g = tap column --
  auto layer1 = [&](auto border_tag) {
    constexpr bool BORDER = decltype(border_tag)::value;
    // two fragments (j, j + 4) per iteration: four independent MFMA chains
    for (int j = wv; j < CB_A1F; j += 8) {
      const bool two = j + 4 < CB_A1F;                  // wave-uniform
      f32x4_t acc[2][2];
      bf16x8_t xf[2][3];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const char* src = P0 + (((u && two) ? j + 4 : j) * 16 + r16 + g) * 16;
#pragma unroll
        for (int s = 0; s < 3; ++s) xf[u][s] = *(const bf16x8_t*)(src + s * CB_PW * 16);
        acc[u][0] = zero4; acc[u][1] = zero4;
      }
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          acc[u][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w1f[0][s]), xf[u][s], acc[u][0], 0, 0, 0);
          acc[u][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w1f[1][s]), xf[u][s], acc[u][1], 0, 0, 0);
        }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (u && !two) break;
        const int q = (j + 4 * u) * 16 + r16;
        float keep = 1.f;
        if (BORDER) {
          const int qy = q / CB_PW, qx = q - qy * CB_PW;
          keep = ((unsigned)(y0 - 2 + qy) < (unsigned)p.H && (unsigned)(x0 - 2 + qx) < (unsigned)p.W) ? 1.f : 0.f;
        }
        cb_store_act(A1, CB_A1PX, q, g, cb_act<BORDER>(acc[u][0] + bias10, p.slope, keep),
                     cb_act<BORDER>(acc[u][1] + bias11, p.slope, keep));
      }
    }
  };
  if (border) layer1(std::true_type{}); else layer1(std::false_type{});
  CB_STAMP(2);
  __syncthreads();
  CB_STAMP(3);
  if (SAVE) {   // centre 16 x 16 of the first activation, 64 B per pixel, coalesced
    for (int idx = tid; idx < CB_T * CB_T * 4; idx += 256) {
      const int pl = idx & 3, pix = idx >> 2, cy = pix >> 4, cx = pix & 15;
      const int y = y0 + cy, x = x0 + cx;
      if (y < p.H && x < p.W)
        *(u32x4_t*)(p.a1 + ((size_t)(b * p.H + y) * p.W + x) * (size_t)p.a1ps * 2 + pl * 16) =
            *(const u32x4_t*)(A1 + (pl * CB_A1PX + (cy + 2) * CB_PW + cx + 2) * 16);
    }
  }
  // ---- layer 2: 32 -> 32 on 18 rows (image origin y0-1, x0-1); K step = tap, lane group g = channel plane ------
  auto layer2 = [&](auto border_tag) {
    constexpr bool BORDER = decltype(border_tag)::value;
    for (int j = wv; j < CB_A2F; j += 8) {
      const bool two = j + 4 < CB_A2F;
      f32x4_t acc[2][2];
      bf16x8_t xf[2][9];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const char* src = A1 + (g * CB_A1PX + ((u && two) ? j + 4 : j) * 16 + r16) * 16;
#pragma unroll
        for (int s = 0; s < 9; ++s) xf[u][s] = *(const bf16x8_t*)(src + ((s / 3) * CB_PW + (s % 3)) * 16);
        acc[u][0] = zero4; acc[u][1] = zero4;
      }
#pragma unroll
      for (int s = 0; s < 9; ++s)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          acc[u][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w2f[0][s]), xf[u][s], acc[u][0], 0, 0, 0);
          acc[u][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w2f[1][s]), xf[u][s], acc[u][1], 0, 0, 0);
        }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (u && !two) break;
        const int q = (j + 4 * u) * 16 + r16;
        float keep = 1.f;
        if (BORDER) {
          const int qy = q / CB_PW, qx = q - qy * CB_PW;
          keep = ((unsigned)(y0 - 1 + qy) < (unsigned)p.H && (unsigned)(x0 - 1 + qx) < (unsigned)p.W) ? 1.f : 0.f;
        }
        cb_store_act(A2, CB_A2PX, q, g, cb_act<BORDER>(acc[u][0] + bias20, p.slope, keep),
                     cb_act<BORDER>(acc[u][1] + bias21, p.slope, keep));
      }
    }
  };
  if (border) layer2(std::true_type{}); else layer2(std::false_type{});
  CB_STAMP(4);
  __syncthreads();
  CB_STAMP(5);
  if (SAVE) {
    for (int idx = tid; idx < CB_T * CB_T * 4; idx += 256) {
      const int pl = idx & 3, pix = idx >> 2, cy = pix >> 4, cx = pix & 15;
      const int y = y0 + cy, x = x0 + cx;
      if (y < p.H && x < p.W)
        *(u32x4_t*)(p.a2 + ((size_t)(b * p.H + y) * p.W + x) * (size_t)p.a2ps * 2 + pl * 16) =
            *(const u32x4_t*)(A2 + (pl * CB_A2PX + (cy + 1) * CB_PW + cx + 1) * 16);
    }
  }
  // ---- layer 3: 32 -> 2 (8 stored) on the 16 x 16 tile: fragment j = output row j (row-aligned: no junk) --------
  for (int j = wv; j < CB_T; j += 8) {
    bf16x8_t xf[2][9];
    f32x4_t acc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const char* src = A2 + (g * CB_A2PX + (j + 4 * u) * CB_PW + r16) * 16;
#pragma unroll
      for (int s = 0; s < 9; ++s) xf[u][s] = *(const bf16x8_t*)(src + ((s / 3) * CB_PW + (s % 3)) * 16);
      acc[u] = zero4;
    }
#pragma unroll
    for (int s = 0; s < 9; ++s)
#pragma unroll
      for (int u = 0; u < 2; ++u)
        acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w3f[s]), xf[u][s], acc[u], 0, 0, 0);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int y = y0 + j + 4 * u, x = x0 + r16;
      if (g < 2 && y < p.H && x < p.W) {
        const size_t o = ((size_t)(b * p.H + y) * p.W + x) * (size_t)p.ops + 4 * g;
        const f32x4_t v = acc[u] + bias3;
        if (p.ops == 2) {                 // dense interleaved complex output: the two real channels only
          if (g == 0) *(f32x2_t*)((float*)p.out + o) = (f32x2_t){v[0], v[1]};
        } else if (p.out_dt == CSMRI_F32) *(f32x4_t*)((float*)p.out + o) = v;
        else *(u32x2_t*)((unsigned short*)p.out + o) = cb_pack4(v);
      }
    }
  }
  CB_STAMP(6);
  }   // persistent tile loop (the next tile's patch store touches P0 only, which nobody reads after layer 1)
#ifdef CSMRI_DBG_STAMPS
  if (!SAVE && p.a1 && lane == 0) {
    unsigned long long* dbg = (unsigned long long*)p.a1 + ((size_t)blockIdx.x * 4 + wv) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) dbg[i] = ph[i];
  }
#endif
}

extern "C" int csmri_convblock_fused_supported(const csmri_convblock_desc* d) {
  if (!d || d->dtype != CSMRI_BF16) return 0;
  if (d->num_convs != 3 || d->num_filters != 32 || d->kernel_size != 3 || d->num_inputs != 2 || d->num_outputs != 2) return 0;
  if (d->border != CSMRI_BORDER_ZERO) return 0;
  if (!(d->slope >= 0.f && d->slope <= 1.f)) return 0;     // LeakyReLU is evaluated as max(v, slope * v)
  return 1;
}

extern "C" int csmri_convblock_fused_fwd(const csmri_convblock_desc* d, void* stream) {
  CSMRI_CHECK_ARG(d && d->x && d->out && d->w[0] && d->w[1] && d->w[2] && d->bias[0] && d->bias[1] && d->bias[2]);
  if (!csmri_convblock_fused_supported(d)) return CSMRI_E_UNSUPPORTED;
  CSMRI_CHECK_ARG(d->B > 0 && d->H > 0 && d->W > 0 && d->x_pix_stride >= 8 && d->x_pix_stride % 8 == 0);
  // out: [B,H,W,8] (channels 2..7 zero), or the dense interleaved complex fp32 image [B,H,W,2] the DC layer consumes
  CSMRI_CHECK_ARG((d->out_pix_stride >= 8 && d->out_pix_stride % 4 == 0) ||
                  (d->out_pix_stride == 2 && d->out_dtype == CSMRI_F32));
  CSMRI_CHECK_ARG(d->out_dtype == CSMRI_F32 || d->out_dtype == CSMRI_BF16);
  CSMRI_CHECK_ARG(d->Kp[0] >= 96 && d->Kp[1] >= 288 && d->Kp[2] >= 288);
#ifndef CSMRI_DBG_STAMPS
  CSMRI_CHECK_ARG((d->act[0] == nullptr) == (d->act[1] == nullptr));
#endif
  if (d->act[0] && d->act[1]) CSMRI_CHECK_ARG(d->act_pix_stride[0] >= 32 && d->act_pix_stride[0] % 8 == 0 &&
                                 d->act_pix_stride[1] >= 32 && d->act_pix_stride[1] % 8 == 0);
  if (((uintptr_t)d->x | (uintptr_t)d->out | (uintptr_t)d->w[0] | (uintptr_t)d->w[1] | (uintptr_t)d->w[2] |
       (uintptr_t)d->act[0] | (uintptr_t)d->act[1] | (uintptr_t)d->bias[0] | (uintptr_t)d->bias[1] |
       (uintptr_t)d->bias[2]) & 15) return CSMRI_E_ALIGN;
  CBParams p;
  p.x = (const char*)d->x; p.xps = d->x_pix_stride;
  p.B = d->B; p.H = d->H; p.W = d->W;
  p.tiles_x = (d->W + CB_T - 1) / CB_T; p.tiles_y = (d->H + CB_T - 1) / CB_T;
  p.w1 = (const char*)d->w[0]; p.w2 = (const char*)d->w[1]; p.w3 = (const char*)d->w[2];
  p.kp1 = d->Kp[0]; p.kp2 = d->Kp[1]; p.kp3 = d->Kp[2];
  p.b1 = d->bias[0]; p.b2 = d->bias[1]; p.b3 = d->bias[2];
  p.slope = d->slope;
  p.a1 = (char*)d->act[0]; p.a1ps = d->act_pix_stride[0]; p.a2 = (char*)d->act[1]; p.a2ps = d->act_pix_stride[1];
  p.out = (char*)d->out; p.out_dt = d->out_dtype; p.ops = d->out_pix_stride;
  p.x_split = d->x_split;
  const long long blocks = (long long)d->B * p.tiles_x * p.tiles_y;
  if (blocks >= (1ll << 31)) return CSMRI_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int wgs = 512;                                  // persistent workgroups: 2 per CU
  const int grid = (int)(blocks < wgs ? blocks : wgs);
  if (d->act[0] && d->act[1]) {
    CSMRI_SET_MAX_LDS(convblock_fwd_kernel<true>, CB_LDS);
    hipLaunchKernelGGL(convblock_fwd_kernel<true>, dim3(grid), dim3(256), CB_LDS, st, p);
  } else {
    CSMRI_SET_MAX_LDS(convblock_fwd_kernel<false>, CB_LDS);
    hipLaunchKernelGGL(convblock_fwd_kernel<false>, dim3(grid), dim3(256), CB_LDS, st, p);
  }
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}
