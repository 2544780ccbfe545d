// Parameter block and LDS-image helpers shared by the weight-gradient kernels (wgrad.hip, wrow.hip).
#pragma once
#include "common.h"

struct WParams {
  const char* in0; const char* in1; int ps0, ps1, c0;
  int B, Hin, Win, Cin, ups, border;
  int KH, KW, S, pt, pl;
  const char* dy; int dyps; int Ho, Wo, Cout;
  float* slab; int splitk; int M, NK, nsteps, steps_per_split, ptiles, qtiles;
  int wr_items, wr_seg, wr_xs, wr_segs;   // wrow: work items, rows per item, column strips per image, segments per strip
  int stages;   // wpatch: 2 = the next tile's images stream in under this tile's MFMAs (LDS permitting), 1 = in place
  float* dwd; int dw_acc;   // wgrad_glds_row DIRECT: the reference-layout gradient itself (no slab, no scatter launch), accumulate flag
};

typedef __attribute__((address_space(1))) const void* wgptr_t;
typedef __attribute__((address_space(3))) void* wlptr_t;

// [pixel][channel] LDS image read with ds_read_b64_tr_b16 (hardware transposed read: a 16-lane group reads a 4-pixel x
// 16-channel block and each lane receives one channel's 4 pixels).  16-byte chunks of a pixel row are XOR-swizzled per row
// so that the transposed reads of a 32-lane half (8 pixel rows x 2 chunks) hit 16 distinct slots.
template <int CPR> __device__ __forceinline__ int img_off(int row, int chunk) {
  int f;
  if constexpr (CPR >= 16) f = ((row & 3) | (((row >> 3) & 1) << 2)) << 1;
  else if constexpr (CPR == 8) f = (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1;
  else if constexpr (CPR == 4) f = ((row >> 3) & 1) << 1;
  else f = 0;
  return row * CPR * 16 + ((chunk ^ f) << 4);
}
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
struct s16x8_pair { s16x4_t lo, hi; };
__device__ __forceinline__ bf16x8_t tr_frag(const char* img, int off_lo, int off_hi) {
  s16x8_pair r;
  r.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(img + off_lo));
  r.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(img + off_hi));
  return __builtin_bit_cast(bf16x8_t, r);
}

// wrow.hip
struct csmri_wgrad_desc;
bool wrow_eligible(const csmri_wgrad_desc* d);
int wrow_groups(const csmri_wgrad_desc* d);
int wrow_launch(const WParams& p, const csmri_wgrad_desc* d, hipStream_t st);
void wrow_kernel_name(const csmri_wgrad_desc* d, char* buf, int n);
