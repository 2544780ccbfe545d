// LDS tile layout + MFMA inner product shared by the implicit-GEMM kernels.
//
// A tile is ROWS x 64 bytes (one "K chunk": 32 bf16 or 16 fp32 along K), stored
// row-major with the four 16-byte slots of a row XOR-swizzled so that the
// ds_read_b128 fragment reads (16 rows x one slot per 16-lane group) and the
// ds_write_b128 staging writes are bank-conflict free (bank row = 256 B = 4 tile
// rows; lane groups of ds_read_b128 per MI355X_MICROARCH.md section LDS).
//
// D[p][q] += sum_k P[p][k] * Q[q][k].  With v_mfma_f32_16x16x32_bf16 /
// v_mfma_f32_16x16x4_f32 lane l holds D[p = 4*(l>>4)+r][q = l&15], r = 0..3:
// four consecutive P rows per lane, so the P side is chosen as the dimension that
// is contiguous in the output tensor (channels).
#pragma once
#include "common.h"

__device__ __forceinline__ int tile_swz(int row) { return (4 - ((row >> 2) & 3)) & 3; }
__device__ __forceinline__ int tile_off(int row, int chunk) {
  return row * 64 + ((chunk ^ tile_swz(row)) << 4);
}

template <int DT, int FP, int FQ>
struct MmaCore {
  f32x4_t acc[FP][FQ];

  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < FP; ++i)
#pragma unroll
      for (int j = 0; j < FQ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }

  // consume one 64-byte K chunk of both tiles
  __device__ __forceinline__ void step(const char* Pt, const char* Qt, int prow0, int qrow0,
                                       int lane) {
    const int r = lane & 15, g = lane >> 4;
    u32x4_t pf[FP], qf[FQ];
#pragma unroll
    for (int i = 0; i < FP; ++i)
      pf[i] = *(const u32x4_t*)(Pt + tile_off(prow0 + i * 16 + r, g));
#pragma unroll
    for (int j = 0; j < FQ; ++j)
      qf[j] = *(const u32x4_t*)(Qt + tile_off(qrow0 + j * 16 + r, g));
    if constexpr (DT == CSMRI_BF16) {
#pragma unroll
      for (int i = 0; i < FP; ++i)
#pragma unroll
        for (int j = 0; j < FQ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              __builtin_bit_cast(bf16x8_t, pf[i]), __builtin_bit_cast(bf16x8_t, qf[j]),
              acc[i][j], 0, 0, 0);
    } else {
      // lane (r,g) holds K elements 4g..4g+3 of its row; MFMA t pairs element t of
      // every lane, i.e. K slot g of MFMA t carries k = 4g+t on both operands.
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < FP; ++i)
#pragma unroll
          for (int j = 0; j < FQ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                __uint_as_float(pf[i][t]), __uint_as_float(qf[j][t]), acc[i][j], 0, 0, 0);
    }
  }
};
