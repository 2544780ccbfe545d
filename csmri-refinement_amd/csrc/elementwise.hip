// HBM-bound companions of the conv kernels: layout converters, weight packing,
// BatchNorm (train) + LeakyReLU + Dropout2d, max-pool, reflect-pad gradient fold.
// All tensors NHWC with an explicit pixel stride; channel vectors of 4 elements
// (16 B fp32 / 8 B bf16) per lane, grid-stride loops capped at 2048 blocks.
#include "common.h"

static inline int grid_for(long long work, int threads = 256) {
  long long b = (work + threads - 1) / threads;
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;
  return (int)b;
}
#define GRID_STRIDE(i, n) \
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)
// 32-bit element index (host checks n < 2^31): the div/mod chains below cost a fraction of the 64-bit ones
#define GRID_STRIDE32(i, n) \
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)(n); i += gridDim.x * blockDim.x)
#define CSMRI_CHECK_I32(n) do { if ((long long)(n) >= (1ll << 31)) return CSMRI_E_UNSUPPORTED; } while (0)

// ---------------------------------------------------------------- layout ----
__global__ void nchw_to_nhwc_kernel(const float* src, int B, int C, long long HW, void* dst,
                                    int dt, int ps, int Cpad) {
  const long long total = (long long)B * HW * Cpad;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % Cpad);
    const long long pix = i / Cpad;
    const long long b = pix / HW, r = pix - b * HW;
    const float v = c < C ? src[(b * C + c) * HW + r] : 0.f;
    store_elem(dst, pix * ps + c, dt, v);
  }
}
// dst = pad(src) + add: the gradient of a tensor handed out both as NCHW fp32 (API layout) and as NHWC (device layout)
__global__ void nchw_to_nhwc_add_kernel(const float* src, int B, int C, long long HW, void* dst, int dt, int ps,
                                        int Cpad, const void* add, int adt, int aps) {
  const long long total = (long long)B * HW * Cpad;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % Cpad);
    const long long pix = i / Cpad;
    const long long b = pix / HW, r = pix - b * HW;
    float v = c < C ? src[(b * C + c) * HW + r] : 0.f;
    if (add) v += load_elem(add, pix * aps + c, adt);
    store_elem(dst, pix * ps + c, dt, v);
  }
}
__global__ void nhwc_to_nchw_kernel(const void* src, int dt, int ps, int B, int C, long long HW,
                                    float* dst) {
  const long long total = (long long)B * C * HW;
  GRID_STRIDE(i, total) {
    const long long r = i % HW;
    const long long bc = i / HW;
    const int c = (int)(bc % C);
    const long long b = bc / C;
    dst[i] = load_elem(src, (b * HW + r) * ps + c, dt);
  }
}
__global__ void mask_to_u8_kernel(const float* m, int B, long long HW, uint8_t* dst) {
  GRID_STRIDE(i, (long long)B * HW) {
    const long long b = i / HW, r = i - b * HW;
    dst[i] = m[(b * 2) * HW + r] != 0.f ? 1 : 0;
  }
}

// Fast paths for the 1- and 2-channel images of the path (complex images as [B,2,H,W] fp32 planes of the reference
// API: inp, kspace, target; magnitudes as [B,1,H,W]): a thread converts FOUR consecutive pixels -- one 16-byte load
// per channel plane, 16-byte stores -- instead of one element per thread (the generic kernel below ran these
// 67 MB conversions at 1.9 TB/s; 6 per RecNet training step).
template <int C, int CP, int DT>
__global__ __launch_bounds__(256) void nchw_to_nhwc_c12_kernel(const float* __restrict__ src, int B, long long HW,
                                                               char* __restrict__ dst, int ps) {
  constexpr int ES = DT == CSMRI_F32 ? 4 : 2;
  const long long quads = (HW >> 2) * B;
  GRID_STRIDE(i, quads) {
    const long long b = i / (HW >> 2), r = (i - b * (HW >> 2)) << 2;
    f32x4_t v[C];
#pragma unroll
    for (int c = 0; c < C; ++c) v[c] = *(const f32x4_t*)(src + (b * C + c) * HW + r);
    char* o = dst + (b * HW + r) * (long long)ps * ES;
    if constexpr (CP == 2 && DT == CSMRI_F32 && C == 2) {            // dense interleaved complex: 4 pixels = 32 bytes
      *(f32x4_t*)o = (f32x4_t){v[0][0], v[1][0], v[0][1], v[1][1]};
      *(f32x4_t*)(o + 16) = (f32x4_t){v[0][2], v[1][2], v[0][3], v[1][3]};
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float e[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < C; ++c) e[c] = v[c][q];
        char* oq = o + (long long)q * ps * ES;
        if constexpr (DT == CSMRI_BF16_SPLIT) {          // C == 2: channels 2,3 = what the rounding of 0,1 dropped
          const float h0 = bf16_bits_to_f32(f32_to_bf16_bits(e[0])), h1 = bf16_bits_to_f32(f32_to_bf16_bits(e[1]));
          const u32x2_t lo = pack4_bf16((f32x4_t){e[0], e[1], e[0] - h0, e[1] - h1});
          *(u32x4_t*)oq = (u32x4_t){lo[0], lo[1], 0u, 0u};
        } else if constexpr (DT == CSMRI_BF16) {
          const u32x2_t lo = pack4_bf16((f32x4_t){e[0], e[1], e[2], e[3]});
          *(u32x4_t*)oq = (u32x4_t){lo[0], lo[1], 0u, 0u};
        } else {
          *(f32x4_t*)oq = (f32x4_t){e[0], e[1], e[2], e[3]};
          *(f32x4_t*)(oq + 16) = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        }
      }
    }
  }
}

extern "C" int csmri_nchw_to_nhwc(const float* src, int B, int C, int H, int W, void* dst,
                                  int dst_dtype, int dst_pix_stride, int Cpad, void* stream) {
  CSMRI_CHECK_ARG(src && dst && Cpad >= C && dst_pix_stride >= Cpad);
  if (dst_dtype == CSMRI_BF16_SPLIT) {
    // a 2-channel image as bf16 hi + lo in a padded pixel of 8: the fast path below only
    if (C != 2 || Cpad != 8 || dst_pix_stride != 8 || (((uintptr_t)src | (uintptr_t)dst) & 15) || ((long long)H * W) % 4)
      return CSMRI_E_UNSUPPORTED;
    const long long HW_ = (long long)H * W;
    hipLaunchKernelGGL((nchw_to_nhwc_c12_kernel<2, 8, CSMRI_BF16_SPLIT>), dim3(grid_for(HW_ / 4 * B)), dim3(256), 0,
                       (hipStream_t)stream, src, B, HW_, (char*)dst, dst_pix_stride);
    CSMRI_LAUNCH_CHECK();
    return CSMRI_OK;
  }
  long long total = (long long)B * H * W * Cpad;
  const long long HW = (long long)H * W;
  const bool al = !(((uintptr_t)src | (uintptr_t)dst) & 15) && HW % 4 == 0;
  if (al && (C == 1 || C == 2) && ((Cpad == 8 && dst_pix_stride == 8) || (Cpad == 2 && C == 2 && dst_pix_stride == 2 &&
                                                                        dst_dtype == CSMRI_F32))) {
    const int grid = grid_for(HW / 4 * B);
    hipStream_t st = (hipStream_t)stream;
#define C12(C_, CP_, DT_) hipLaunchKernelGGL((nchw_to_nhwc_c12_kernel<C_, CP_, DT_>), dim3(grid), dim3(256), 0, st, src, B, HW, (char*)dst, dst_pix_stride)
    if (Cpad == 2) C12(2, 2, CSMRI_F32);
    else if (dst_dtype == CSMRI_BF16) { if (C == 2) C12(2, 8, CSMRI_BF16); else C12(1, 8, CSMRI_BF16); }
    else { if (C == 2) C12(2, 8, CSMRI_F32); else C12(1, 8, CSMRI_F32); }
#undef C12
    CSMRI_LAUNCH_CHECK();
    return CSMRI_OK;
  }
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                     src, B, C, (long long)H * W, dst, dst_dtype, dst_pix_stride, Cpad);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}
extern "C" int csmri_nchw_to_nhwc_add(const float* src, int B, int C, int H, int W, void* dst, int dst_dtype,
                                      int dst_pix_stride, int Cpad, const void* add, int add_dtype,
                                      int add_pix_stride, void* stream) {
  CSMRI_CHECK_ARG(src && dst && Cpad >= C && dst_pix_stride >= Cpad && (!add || add_pix_stride >= Cpad));
  const long long total = (long long)B * H * W * Cpad;
  hipLaunchKernelGGL(nchw_to_nhwc_add_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, B, C,
                     (long long)H * W, dst, dst_dtype, dst_pix_stride, Cpad, add, add_dtype, add_pix_stride);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}
extern "C" int csmri_nhwc_to_nchw(const void* src, int src_dtype, int src_pix_stride, int B,
                                  int C, int H, int W, float* dst, void* stream) {
  CSMRI_CHECK_ARG(src && dst);
  long long total = (long long)B * H * W * C;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                     src, src_dtype, src_pix_stride, B, C, (long long)H * W, dst);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}
extern "C" int csmri_mask_to_u8(const float* mask_nchw, int B, int H, int W, uint8_t* dst, void* stream) {
  CSMRI_CHECK_ARG(mask_nchw && dst);
  hipLaunchKernelGGL(mask_to_u8_kernel, dim3(grid_for((long long)B * H * W)), dim3(256), 0,
                     (hipStream_t)stream, mask_nchw, B, (long long)H * W, dst);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

// -------------------------------------------------------- weight packing ----
__host__ __device__ static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

// tw = packed filter width: for few-channel layers (chan_pad 8/16) the width is padded with
// zero taps to a multiple of 32/chan_pad so that a 32-wide K chunk is a whole number of
// horizontally adjacent taps (tconv.hip).
struct PackGeom { int rows, rows_pad, chan_pad, th, tw, taps, Kp, nclass; };
__host__ __device__ static PackGeom pack_geom(int mode, int Cout, int Cin, int KH, int KW) {
  PackGeom g;
  const bool swapped = mode != 0;
  g.rows = swapped ? Cin : Cout;
  g.chan_pad = round_up(swapped ? Cout : Cin, 8);
  g.nclass = mode == 2 ? 4 : 1;
  g.th = mode == 2 ? KH / 2 : KH;
  g.tw = mode == 2 ? KW / 2 : KW;
  if (mode != 2 && (g.chan_pad == 8 || g.chan_pad == 16)) g.tw = round_up(g.tw, 32 / g.chan_pad);
  g.taps = g.th * g.tw;
  g.rows_pad = round_up(g.rows, 128);
  g.Kp = round_up(g.taps * g.chan_pad, 64);
  return g;
}

__global__ void pack_weight_kernel(int mode, int dt, const float* w, int Cout, int Cin, int KH, int KW,
                                   void* out, int rows, int rows_pad, int chan_pad, int taps, int Kp,
                                   int nclass, int tw) {
  const long long per_class = (long long)rows_pad * Kp;
  GRID_STRIDE(i, per_class * nclass) {
    const int cls = (int)(i / per_class);
    const long long rem = i - cls * per_class;
    const int row = (int)(rem / Kp), k = (int)(rem - (long long)row * Kp);
    const int tap = k / chan_pad, ch = k - tap * chan_pad;
    const int ty = tap / tw, tx = tap - ty * tw;
    float v = 0.f;
    if (row < rows && tap < taps) {
      if (mode == 0) {
        if (ch < Cin && tx < KW) v = w[(((long long)row * Cin + ch) * KH + ty) * KW + tx];
      } else if (mode == 1) {
        if (ch < Cout && tx < KW) v = w[(((long long)ch * Cin + row) * KH + ty) * KW + tx];
      } else if (mode == 3) {      // flipped taps: dgrad as a plain correlation
        if (ch < Cout && tx < KW) v = w[(((long long)ch * Cin + row) * KH + (KH - 1 - ty)) * KW + (KW - 1 - tx)];
      } else {
        const int ky = (cls >> 1) + 2 * ty, kx = (cls & 1) + 2 * tx;
        if (ch < Cout) v = w[(((long long)ch * Cin + row) * KH + ky) * KW + kx];
      }
    }
    store_elem(out, i, dt, v);
  }
}

// All layers of a network in ONE launch: blockIdx.y selects the item (table in device memory),
// blockIdx.x grid-strides over that item's packed elements.
__device__ __forceinline__ float pack_elem(const csmri_pack_item& it, const PackGeom& g, long long i) {
  const long long per_class = (long long)g.rows_pad * g.Kp;
  const int cls = (int)(i / per_class);
  const long long rem = i - cls * per_class;
  const int row = (int)(rem / g.Kp), k = (int)(rem - (long long)row * g.Kp);
  const int tap = k / g.chan_pad, ch = k - tap * g.chan_pad;
  const int ty = tap / g.tw, tx = tap - ty * g.tw;
  if (!(row < g.rows && tap < g.taps)) return 0.f;
  const float* w = it.w;
  const int Cin = it.Cin, Cout = it.Cout, KH = it.KH, KW = it.KW;
  if (it.mode == 0) return (ch < Cin && tx < KW) ? w[(((long long)row * Cin + ch) * KH + ty) * KW + tx] : 0.f;
  if (it.mode == 1) return (ch < Cout && tx < KW) ? w[(((long long)ch * Cin + row) * KH + ty) * KW + tx] : 0.f;
  if (it.mode == 3) return (ch < Cout && tx < KW) ? w[(((long long)ch * Cin + row) * KH + (KH - 1 - ty)) * KW + (KW - 1 - tx)] : 0.f;
  const int ky = (cls >> 1) + 2 * ty, kx = (cls & 1) + 2 * tx;
  return ch < Cout ? w[(((long long)ch * Cin + row) * KH + ky) * KW + kx] : 0.f;
}
// Re-pack of buffers that were fully written once by csmri_pack_weight (so their zero padding
// of rows / K tail is in place): a tiled 3-D transpose through LDS.  One tile = one destination
// row (co, or ci for the swapped modes) x 64 destination channels x all taps; the fp32 source is
// read in runs of taps (mode 0: 64*taps contiguous floats) and the packed destination written as
// 64 consecutive elements per tap.  Requires KH*KW <= 16.
__global__ __launch_bounds__(256) void pack_weight_multi_kernel(const csmri_pack_item* items) {
  __shared__ float tile[16][65];
  const csmri_pack_item it = items[blockIdx.y];
  const float* __restrict__ it_w = it.w;
  if (blockIdx.x == 0 && it.bias)        // the layer's bias into its zero-padded fp32 buffer, same launch
    for (int c = threadIdx.x; c < it.Cout; c += 256) it.bias_out[c] = it.bias[c];
  if (it.mode < 0) return;               // bias-only item
  const PackGeom g = pack_geom(it.mode, it.Cout, it.Cin, it.KH, it.KW);
  const int T = it.KH * it.KW, KW = it.KW, KH = it.KH;
  const bool swapped = it.mode != 0;
  const int chan = swapped ? it.Cout : it.Cin;
  const int nchunk = (g.chan_pad + 63) / 64;
  const long long per_class = (long long)g.rows_pad * g.Kp;
  const int ntap = g.th * g.tw;
  for (int t = blockIdx.x; t < g.rows * nchunk; t += gridDim.x) {
    const int r = t / nchunk, c0 = (t - r * nchunk) * 64;
    const int nc = max(0, min(64, chan - c0));
    if (T == 16 && KW == 4 && nc == 64 && it.mode != 2 && it.dtype == CSMRI_BF16 && g.chan_pad - c0 >= 64 &&
        (((uintptr_t)it_w | (uintptr_t)it.out) & 15) == 0 && (g.chan_pad & 3) == 0 && (g.Kp & 3) == 0) {
      // full 64-channel x 4x4-tap tile (every discriminator / U-Net 4x4 layer): one 16-byte load of four taps and one
      // 8-byte store of four channels per thread instead of four scalar loads and four 2-byte stores
      const int ch = threadIdx.x >> 2, q4 = threadIdx.x & 3;
      const f32x4_t v = *(const f32x4_t*)(swapped ? it_w + ((long long)(c0 + ch) * it.Cin + r) * 16 + q4 * 4
                                                  : it_w + ((long long)r * it.Cin + c0 + ch) * 16 + q4 * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) tile[q4 * 4 + j][ch] = v[j];
      __syncthreads();
      const int tp = threadIdx.x >> 4, c4 = (threadIdx.x & 15) * 4;
      const int src = it.mode == 3 ? 15 - tp : tp;            // mode 3: taps flipped in both axes
      const f32x4_t o = (f32x4_t){tile[src][c4], tile[src][c4 + 1], tile[src][c4 + 2], tile[src][c4 + 3]};
      store4(it.out, (long long)r * g.Kp + (long long)tp * g.chan_pad + c0 + c4, CSMRI_BF16, o);
      __syncthreads();
      continue;
    }
    // at most 64*16 elements per tile = 4 per thread: fixed-trip loops keep all four global loads
    // (and later all four stores) of a thread in flight together
    float vals[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = threadIdx.x + u * 256;
      vals[u] = 0.f;
      if (e < nc * T) {
        const int ch = e / T, tap = e - ch * T;
        vals[u] = swapped ? it_w[((long long)(c0 + ch) * it.Cin + r) * T + tap]
                           : it_w[((long long)r * it.Cin + c0 + ch) * T + tap];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = threadIdx.x + u * 256;
      if (e < nc * T) { const int ch = e / T, tap = e - ch * T; tile[tap][ch] = vals[u]; }
    }
    __syncthreads();
    const int ncp = min(64, g.chan_pad - c0);
    const int nout = g.nclass * ntap * ncp;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = threadIdx.x + u * 256;
      if (e < nout) {
        const int ch = e % ncp, q = e / ncp;
        const int tp = q % ntap, cls = q / ntap;
        const int ty = tp / g.tw, tx = tp - ty * g.tw;
        int src = -1;
        if (it.mode == 2) src = ((cls >> 1) + 2 * ty) * KW + (cls & 1) + 2 * tx;
        else if (tx < KW) src = it.mode == 3 ? (KH - 1 - ty) * KW + (KW - 1 - tx) : ty * KW + tx;
        const float v = (src >= 0 && ch < nc) ? tile[src][ch] : 0.f;
        store_elem(it.out, cls * per_class + (long long)r * g.Kp + (long long)tp * g.chan_pad + c0 + ch, it.dtype, v);
      }
    }
    __syncthreads();
  }
}
extern "C" int csmri_pack_weight_multi(const csmri_pack_item* items_dev, int n, void* stream) {
  CSMRI_CHECK_ARG(items_dev && n > 0);
  hipLaunchKernelGGL(pack_weight_multi_kernel, dim3(2048, n), dim3(256), 0, (hipStream_t)stream, items_dev);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

extern "C" size_t csmri_pack_weight_bytes(int mode, int dtype, int Cout, int Cin, int KH, int KW) {
  PackGeom g = pack_geom(mode, Cout, Cin, KH, KW);
  return (size_t)g.nclass * g.rows_pad * g.Kp * dtype_size(dtype);
}
extern "C" int csmri_pack_weight(int mode, int dtype, const float* w_ref, int Cout, int Cin, int KH,
                                 int KW, void* out, int* Kp_out, long long* class_stride_out,
                                 int* TW_out, void* stream) {
  CSMRI_CHECK_ARG(w_ref && out && mode >= 0 && mode <= 3);
  if (mode == 2) CSMRI_CHECK_ARG(KH % 2 == 0 && KW % 2 == 0);
  PackGeom g = pack_geom(mode, Cout, Cin, KH, KW);
  if (Kp_out) *Kp_out = g.Kp;
  if (class_stride_out) *class_stride_out = (long long)g.rows_pad * g.Kp;
  if (TW_out) *TW_out = g.tw;
  long long total = (long long)g.nclass * g.rows_pad * g.Kp;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                     mode, dtype, w_ref, Cout, Cin, KH, KW, out, g.rows, g.rows_pad, g.chan_pad,
                     g.taps, g.Kp, g.nclass, g.tw);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

__global__ void act_bwd_kernel(int dt, const void* dz, int dzps, const void* z, int zps, void* dy,
                               int dyps, long long npix, int C, float slope, const void* dz2, int dz2ps) {
  const int nv = C >> 2;
  GRID_STRIDE32(i, npix * nv) {
    const int c = (int)(i % nv) * 4;
    const unsigned p = i / nv;
    f32x4_t g = load4(dz, (long long)p * dzps + c, dt), zz = load4(z, (long long)p * zps + c, dt);
    if (dz2) g += load4(dz2, (long long)p * dz2ps + c, dt);    // gradient fan-in of z: summed here in fp32
    for (int q = 0; q < 4; ++q) g[q] = zz[q] > 0.f ? g[q] : g[q] * slope;
    store4(dy, (long long)p * dyps + c, dt, g);
  }
}
extern "C" int csmri_act_bwd(int dtype, const void* dz, int dz_pix_stride, const void* z,
                             int z_pix_stride, void* dy, int dy_pix_stride, long long npix, int C,
                             float slope, const void* dz2, int dz2_pix_stride, void* stream) {
  CSMRI_CHECK_ARG(dz && z && dy && C % 4 == 0);
  CSMRI_CHECK_I32(npix * C);
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(npix * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                     dtype, dz, dz_pix_stride, z, z_pix_stride, dy, dy_pix_stride, npix, C, slope, dz2,
                     dz2_pix_stride);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

// --------------------------------------------------------------- max pool ----
// yq / qs / amax (csmri_maxpool2_q): an fp8 (e4m3fn) copy of the pooled bf16 output, value * *qs rounded by
// v_cvt_pk_fp8_f32 (bit for bit csmri_quantize_fp8 of y), and atomicMax of the bit patterns |y| (csmri_absmax of y)
__global__ void maxpool2_kernel(int dt, const void* x, int xps, void* y, int yps, uint8_t* arg, int B,
                                int H, int W, int C, uint8_t* yq = nullptr, int yqps = 0, const float* qs = nullptr,
                                unsigned* amax = nullptr) {
  const int nv = C >> 2, Ho = H >> 1, Wo = W >> 1;
  const float qscale = qs ? *qs : 1.f;
  unsigned am = 0;
  GRID_STRIDE32(i, (long long)B * Ho * Wo * nv) {
    const int c = (int)(i % nv) * 4;
    const unsigned p = i / nv;
    const int ox = (int)(p % Wo);
    const unsigned t = p / Wo;
    const int oy = (int)(t % Ho), b = (int)(t / Ho);
    const long long base = ((long long)b * H + 2 * oy) * W + 2 * ox;
    f32x4_t best = load4(x, base * xps + c, dt);
    int idx[4] = {0, 0, 0, 0};
    for (int k = 1; k < 4; ++k) {
      f32x4_t v = load4(x, (base + (k >> 1) * W + (k & 1)) * xps + c, dt);
      for (int q = 0; q < 4; ++q)
        if (v[q] > best[q] || v[q] != v[q]) { best[q] = v[q]; idx[q] = k; }
    }
    store4(y, (long long)p * yps + c, dt, best);
    if (arg) *(uint32_t*)(arg + (long long)p * C + c) = idx[0] | (idx[1] << 8) | (idx[2] << 16) | (idx[3] << 24);
    if (yq || amax) {
      // (the pooled values are elements of x: bf16-exact already when dt is bf16)
      if (amax) {
        for (int q = 0; q < 4; ++q) {
          const unsigned ab = __float_as_uint(best[q]) & 0x7fffffffu;
          am = (ab <= 0x7f800000u && ab > am) ? ab : am;
        }
      }
      if (yq) {
        int r = 0;
        r = __builtin_amdgcn_cvt_pk_fp8_f32(sat_e4m3(best[0] * qscale), sat_e4m3(best[1] * qscale), r, false);
        r = __builtin_amdgcn_cvt_pk_fp8_f32(sat_e4m3(best[2] * qscale), sat_e4m3(best[3] * qscale), r, true);
        *(uint32_t*)(yq + (long long)p * yqps + c) = (uint32_t)r;
      }
    }
  }
  if (amax) {
    for (int o = 32; o > 0; o >>= 1) { const unsigned t_ = __shfl_xor(am, o); am = t_ > am ? t_ : am; }
    if ((threadIdx.x & 63) == 0 && am) {       // (read first: the maximum only grows, most waves cannot raise it)
      const unsigned cur = __hip_atomic_load(amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (am > cur) atomicMax(amax, am);
    }
  }
}
// gadd != null: a second gradient of the pooled tensor's source (skip connection) is added in the same pass.
// gsrc != null: dx additionally carries the activation derivative of the layer that produced the pooled
// tensor (gsrc = its output): one pass instead of maxpool2_bwd + act_bwd over the full-resolution gradient
// gpooled: gsrc is the POOLED tensor (the pool's output): at the position the gradient is routed to, the producer's output
// IS the pooled value, everywhere else the result is zero whatever the gate says -- same bits as gating with the
// full-resolution tensor, a quarter of its bytes (csmri_maxpool2_bwd_pooled_gate)
__global__ void maxpool2_bwd_kernel(int dt, const void* dy, int dyps, const uint8_t* arg, void* dx,
                                    int dxps, int B, int H, int W, int C, const void* gsrc, int gps,
                                    float gslope, const void* gadd, int gaps, int gpooled = 0) {
  const int nv = C >> 2, Ho = H >> 1, Wo = W >> 1;
  GRID_STRIDE32(i, (long long)B * Ho * Wo * nv) {
    const int c = (int)(i % nv) * 4;
    const unsigned p = i / nv;
    const int ox = (int)(p % Wo);
    const unsigned t = p / Wo;
    const int oy = (int)(t % Ho), b = (int)(t / Ho);
    const long long base = ((long long)b * H + 2 * oy) * W + 2 * ox;
    f32x4_t g = load4(dy, (long long)p * dyps + c, dt);
    const uint32_t a = *(const uint32_t*)(arg + (long long)p * C + c);
    for (int k = 0; k < 4; ++k) {
      f32x4_t o;
      for (int q = 0; q < 4; ++q) o[q] = ((a >> (8 * q)) & 0xff) == (unsigned)k ? g[q] : 0.f;
      if (gadd) o += load4(gadd, (base + (k >> 1) * W + (k & 1)) * gaps + c, dt);
      if (gsrc) {
        const f32x4_t s = load4(gsrc, (gpooled ? (long long)p : base + (k >> 1) * W + (k & 1)) * gps + c, dt);
        for (int q = 0; q < 4; ++q) o[q] = s[q] > 0.f ? o[q] : o[q] * gslope;
      }
      store4(dx, (base + (k >> 1) * W + (k & 1)) * dxps + c, dt, o);
    }
  }
}
extern "C" int csmri_maxpool2(int dtype, const void* x, int x_pix_stride, void* y, int y_pix_stride,
                              uint8_t* argmax, int B, int H, int W, int C, void* stream) {
  CSMRI_CHECK_ARG(x && y && H % 2 == 0 && W % 2 == 0 && C % 4 == 0);
  long long n = (long long)B * (H / 2) * (W / 2) * (C / 4);
  CSMRI_CHECK_I32((long long)B * H * W * C);
  hipLaunchKernelGGL(maxpool2_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dtype, x,
                     x_pix_stride, y, y_pix_stride, argmax, B, H, W, C);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}
extern "C" int csmri_maxpool2_q(int dtype, const void* x, int x_pix_stride, void* y, int y_pix_stride,
                                uint8_t* argmax, int B, int H, int W, int C, void* y_q, int y_q_pix_stride,
                                const float* q_scale, float* amax, void* stream) {
  CSMRI_CHECK_ARG(x && y && H % 2 == 0 && W % 2 == 0 && C % 4 == 0 && dtype == CSMRI_BF16);
  CSMRI_CHECK_ARG((!y_q || (q_scale && y_q_pix_stride % 4 == 0 && !((uintptr_t)y_q & 3))));
  long long n = (long long)B * (H / 2) * (W / 2) * (C / 4);
  CSMRI_CHECK_I32((long long)B * H * W * C);
  hipLaunchKernelGGL(maxpool2_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dtype, x,
                     x_pix_stride, y, y_pix_stride, argmax, B, H, W, C, (uint8_t*)y_q, y_q_pix_stride, q_scale, (unsigned*)amax);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}
extern "C" int csmri_maxpool2_bwd_act(int dtype, const void* dy, int dy_pix_stride, const uint8_t* argmax,
                                      void* dx, int dx_pix_stride, int B, int H, int W, int C, const void* g_src,
                                      int g_pix_stride, float g_slope, const void* g_add, int g_add_pix_stride,
                                      void* stream) {
  CSMRI_CHECK_ARG(dy && dx && argmax && H % 2 == 0 && W % 2 == 0 && C % 4 == 0);
  long long n = (long long)B * (H / 2) * (W / 2) * (C / 4);
  CSMRI_CHECK_I32((long long)B * H * W * C);
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dtype,
                     dy, dy_pix_stride, argmax, dx, dx_pix_stride, B, H, W, C, g_src, g_pix_stride, g_slope, g_add,
                     g_add_pix_stride);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}
extern "C" int csmri_maxpool2_bwd_pooled_gate(int dtype, const void* dy, int dy_pix_stride, const uint8_t* argmax,
                                              void* dx, int dx_pix_stride, int B, int H, int W, int C,
                                              const void* g_pooled, int g_pix_stride, float g_slope, void* stream) {
  CSMRI_CHECK_ARG(dy && dx && argmax && g_pooled && H % 2 == 0 && W % 2 == 0 && C % 4 == 0);
  long long n = (long long)B * (H / 2) * (W / 2) * (C / 4);
  CSMRI_CHECK_I32((long long)B * H * W * C);
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dtype,
                     dy, dy_pix_stride, argmax, dx, dx_pix_stride, B, H, W, C, g_pooled, g_pix_stride, g_slope,
                     (const void*)nullptr, 0, 1);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}
extern "C" int csmri_maxpool2_bwd(int dtype, const void* dy, int dy_pix_stride, const uint8_t* argmax,
                                  void* dx, int dx_pix_stride, int B, int H, int W, int C, void* stream) {
  return csmri_maxpool2_bwd_act(dtype, dy, dy_pix_stride, argmax, dx, dx_pix_stride, B, H, W, C, nullptr, 0, 1.f,
                                nullptr, 0, stream);
}

// ----------------------------------------------------- reflect-pad fold ----
// positions of the padded axis that map onto un-padded index i (n = extent)
__device__ __forceinline__ int fold_sources(int i, int n, int lo, int hi, int* src) {
  int k = 0;
  src[k++] = i + lo;
  if (i >= 1 && i <= lo) src[k++] = lo - i;
  if (i >= n - 1 - hi && i <= n - 2) src[k++] = lo + n + (n - 2 - i);
  return k;
}
__global__ void fold_pad_grad_kernel(int dt, const void* gp, void* out, int ops, int B, int H, int W,
                                     int C, int pt, int pb, int pl, int pr, int ups, const void* gsrc,
                                     int gps, float gslope) {
  const int nv = C >> 2;
  const int Hu = ups ? 2 * H : H, Wu = ups ? 2 * W : W;
  const int Hp = Hu + pt + pb, Wp = Wu + pl + pr;
  GRID_STRIDE32(i, (long long)B * H * W * nv) {
    const int c = (int)(i % nv) * 4;
    const unsigned p = i / nv;
    const int x = (int)(p % W);
    const unsigned t = p / W;
    const int y = (int)(t % H), b = (int)(t / H);
    f32x4_t acc = (f32x4_t){0, 0, 0, 0};
    const int ny = ups ? 2 : 1;
    for (int sy = 0; sy < ny; ++sy)
      for (int sx = 0; sx < ny; ++sx) {
        int ys[3], xs[3];
        const int ky = fold_sources(ups ? 2 * y + sy : y, Hu, pt, pb, ys);
        const int kx = fold_sources(ups ? 2 * x + sx : x, Wu, pl, pr, xs);
        for (int a = 0; a < ky; ++a)
          for (int e = 0; e < kx; ++e)
            acc += load4(gp, (((long long)b * Hp + ys[a]) * Wp + xs[e]) * C + c, dt);
      }
    if (gsrc) {
      f32x4_t s = load4(gsrc, (long long)p * gps + c, dt);
      for (int q = 0; q < 4; ++q) acc[q] = s[q] > 0.f ? acc[q] : acc[q] * gslope;
    }
    store4(out, (long long)p * ops + c, dt, acc);
  }
}
// border-only fold: dx (dense, un-padded) already holds the centre contribution; add what the
// reflection padding mirrors back from the halo positions of the padded gradient
__global__ void fold_halo_kernel(int dt, const void* gp, void* dx, int dxps, int B, int H, int W, int C,
                                 int pt, int pb, int pl, int pr, const void* gsrc, int gps, float gslope) {
  const int nv = C >> 2;
  const int Hp = H + pt + pb, Wp = W + pl + pr;
  const int nyb = pt + pb, nxb = pl + pr;
  const unsigned per_img = (unsigned)(nyb * W + (H - nyb) * nxb);
  GRID_STRIDE32(i, (unsigned)B * per_img * nv) {
    const int c = (int)(i % nv) * 4;
    const unsigned q = i / nv;
    const int b = (int)(q / per_img);
    const int j = (int)(q - (unsigned)b * per_img);
    int y, x;
    if (j < nyb * W) {                     // border rows, every column
      const int yi = j / W;
      x = j - yi * W;
      y = yi < pt ? 1 + yi : H - 1 - pb + (yi - pt);
    } else {                               // remaining rows, border columns only
      const int k = j - nyb * W, yy = k / nxb, xi = k - yy * nxb;
      y = yy == 0 ? 0 : (yy <= H - 2 - pb - pt ? pt + yy : H - 1);
      x = xi < pl ? 1 + xi : W - 1 - pr + (xi - pl);
    }
    int ys[3], xs[3];
    const int ky = fold_sources(y, H, pt, pb, ys), kx = fold_sources(x, W, pl, pr, xs);
    f32x4_t acc = (f32x4_t){0, 0, 0, 0};
    for (int a = 0; a < ky; ++a)
      for (int e = 0; e < kx; ++e)
        if (a | e) acc += load4(gp, (((long long)b * Hp + ys[a]) * Wp + xs[e]) * C + c, dt);
    const long long p = ((long long)b * H + y) * W + x;
    if (gsrc) {
      f32x4_t s = load4(gsrc, p * gps + c, dt);
      for (int r = 0; r < 4; ++r) acc[r] = s[r] > 0.f ? acc[r] : acc[r] * gslope;
    }
    f32x4_t cur = load4(dx, p * dxps + c, dt);
    store4(dx, p * dxps + c, dt, cur + acc);
  }
}
extern "C" int csmri_fold_halo(int dtype, const void* gpad_halo, void* dx, int dx_pix_stride, int B, int H,
                               int W, int C, int pt, int pb, int pl, int pr, const void* g_src,
                               int g_pix_stride, float g_slope, void* stream) {
  CSMRI_CHECK_ARG(gpad_halo && dx && C % 4 == 0 && H >= pt + pb + 2 && W >= pl + pr + 2);
  CSMRI_CHECK_I32((long long)B * (H + pt + pb) * (W + pl + pr) * C);
  const long long n = (long long)B * ((pt + pb) * W + (H - pt - pb) * (pl + pr)) * (C / 4);
  if (n <= 0) return CSMRI_OK;
  hipLaunchKernelGGL(fold_halo_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dtype, gpad_halo, dx,
                     dx_pix_stride, B, H, W, C, pt, pb, pl, pr, g_src, g_pix_stride, g_slope);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}
extern "C" int csmri_fold_pad_grad(int dtype, const void* gpad, void* out, int out_pix_stride, int B,
                                   int H, int W, int C, int pt, int pb, int pl, int pr, int upsample,
                                   const void* g_src, int g_pix_stride, float g_slope, void* stream) {
  CSMRI_CHECK_ARG(gpad && out && C % 4 == 0);
  long long n = (long long)B * H * W * (C / 4);
  CSMRI_CHECK_I32((long long)B * (2 * H + pt + pb) * (2 * W + pl + pr) * C);
  hipLaunchKernelGGL(fold_pad_grad_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dtype,
                     gpad, out, out_pix_stride, B, H, W, C, pt, pb, pl, pr, upsample, g_src,
                     g_pix_stride, g_slope);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

// ------------------------------------------------------------------- misc ----
__global__ void fill_kernel(float* p, long long n, float v) { GRID_STRIDE(i, n) p[i] = v; }
extern "C" int csmri_fill_f32(float* p, long long n, float v, void* stream) {
  CSMRI_CHECK_ARG(p);
  hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, n, v);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}
__global__ void cast_kernel(const void* s, int sdt, void* d, int ddt, long long n) {
  GRID_STRIDE(i, n) store_elem(d, i, ddt, load_elem(s, i, sdt));
}
extern "C" int csmri_cast(const void* src, int src_dtype, void* dst, int dst_dtype, long long n, void* stream) {
  CSMRI_CHECK_ARG(src && dst);
  hipLaunchKernelGGL(cast_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, src, src_dtype,
                     dst, dst_dtype, n);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

__global__ void copy_channels_kernel(const void* s, int sdt, int sps, int cs, void* d, int ddt, int dps,
                                     int cd, long long npix) {
  GRID_STRIDE32(i, npix * cd) {
    const int c = (int)(i % cd);
    const unsigned p = i / cd;
    store_elem(d, (long long)p * dps + c, ddt, c < cs ? load_elem(s, (long long)p * sps + c, sdt) : 0.f);
  }
}
extern "C" int csmri_copy_channels(const void* src, int src_dtype, int src_pix_stride, int C_src, void* dst,
                                   int dst_dtype, int dst_pix_stride, int C_dst, long long npix,
                                   void* stream) {
  CSMRI_CHECK_ARG(src && dst && C_src > 0 && C_dst > 0 && npix > 0);
  CSMRI_CHECK_I32(npix * (C_dst > C_src ? C_dst : C_src));
  hipLaunchKernelGGL(copy_channels_kernel, dim3(grid_for(npix * C_dst)), dim3(256), 0, (hipStream_t)stream,
                     src, src_dtype, src_pix_stride, C_src, dst, dst_dtype, dst_pix_stride, C_dst, npix);
  CSMRI_LAUNCH_CHECK();
  return CSMRI_OK;
}

extern "C" int csmri_version(void) { return 103; }   // 103: csmri_adam_dev_lr, csmri_gconv_desc.flags dispatch switches; 101: csmri_dc_in_bf16 takes x_dtype; csmri_dropout2d_mask state is uint64[3]
extern "C" const char* csmri_error_string(int code) {
  switch (code) {
    case CSMRI_OK: return "ok";
    case CSMRI_E_ARG: return "csmri: bad argument";
    case CSMRI_E_UNSUPPORTED: return "csmri: unsupported shape or option";
    case CSMRI_E_ALIGN: return "csmri: pointer not 16-byte aligned";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "csmri: unknown error";
  }
}
extern "C" int csmri_shutdown(void) { return CSMRI_OK; }
