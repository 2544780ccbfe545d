// Kernel-side parameter block shared by gconv.hip (generic implicit GEMM) and tconv.hip
// (LDS-patch kernel for few-channel stride-1 layers).
#pragma once
#include "common.h"

struct GParams {
  const char* in0; const char* in1; int ps0, ps1, c0;
  int B, Hin, Win, Cin, ups, border;
  int TH, TW, S, dy0, dys, dx0, dxs;
  const char* w; int Kp; int nclass; long long wcs;
  char* out; int ops; int Hout_t, Wout_t, Ho, Wo, osy, osx, ooy, oox, Cout; int out_dt;
  const float* bias; float slope; const char* gsrc; int gps; float gslope; int gdt;
  float* stats; int splitk; float* slab;
  int M, nsteps, steps_per_split, mtiles, ntiles;
  int nt_major;   // gconv_glds: tile order that keeps the larger operand shared inside an XCD
  char* out2; int o2ps, win_y0, win_x0, win_h, win_w;   // output window (csmri_gconv_desc.out_halo)
  // fast paths (host-decided, wave-uniform):
  int wo_shift, howo_shift;   // log2(Wo), log2(Ho*Wo) when both are powers of two, else -1: m -> (b,oy,ox) by shifts
  int dense_out;              // output position index == m (no window / stride / offset / classes): no division at all
  int off32;                  // every input / output byte offset fits 32 bits
  const float* dq0; const float* dq1;   // fp8 operands: device scalars whose product dequantises the accumulators
  int us_n, us_x, us_y;                 // uconv: strips in all, strips per row of strips, rows of strips per image
  unsigned dv_howo_m, dv_howo_s, dv_wo_m, dv_wo_s;   // gpipe: magic numbers of the divisions by Ho*Wo and Wo (non powers of two)
  char* outq; int oqps; const float* oqs; unsigned* oamax;   // pconv2: fp8 copy of the output, its scale, |out| maximum (bits)
};

// where output position (b, ty, tx) of the tensor goes: window -> dense `out`, else halo buffer
struct OutPos { char* base; size_t opix, gpix; bool g_ok; };
__device__ __forceinline__ OutPos gconv_out_pos(const GParams& p, int b, int ty, int tx) {
  OutPos o;
  if (p.out2) {
    const int cy = ty - p.win_y0, cx = tx - p.win_x0;
    if ((unsigned)cy < (unsigned)p.win_h && (unsigned)cx < (unsigned)p.win_w) {
      const size_t pp = ((size_t)b * p.win_h + cy) * p.win_w + cx;
      o.base = p.out; o.opix = pp * p.ops; o.gpix = pp * p.gps; o.g_ok = true;
    } else {
      const size_t pp = ((size_t)b * p.Hout_t + ty) * p.Wout_t + tx;
      o.base = p.out2; o.opix = pp * p.o2ps; o.gpix = 0; o.g_ok = false;
    }
  } else {
    const size_t pp = ((size_t)b * p.Hout_t + ty) * p.Wout_t + tx;
    o.base = p.out; o.opix = pp * p.ops; o.gpix = pp * p.gps; o.g_ok = true;
  }
  return o;
}

// thin.hip: one real channel on the N side
int thin_out1_eligible(const csmri_gconv_desc* d);
int thin_out1_launch(const GParams& p, const csmri_gconv_desc* d, hipStream_t st);
void thin_kernel_name(const csmri_gconv_desc* d, char* buf, int n);

// tconv.hip
int tconv_eligible(const csmri_gconv_desc* d);
int tconv_stats_rows(const csmri_gconv_desc* d);
int tconv_launch(const GParams& p, const csmri_gconv_desc* d, hipStream_t st);
void tconv_kernel_name(const csmri_gconv_desc* d, char* buf, int n);

// uconv.hip
int uconv_eligible(const csmri_gconv_desc* d);
int uconv_stats_rows(const csmri_gconv_desc* d);
int uconv_launch(const GParams& p, const csmri_gconv_desc* d, hipStream_t st);
void uconv_kernel_name(const csmri_gconv_desc* d, char* buf, int n);

// pconv2.hip
int pconv2_eligible(const csmri_gconv_desc* d);
int pconv2_bn(const csmri_gconv_desc* d);
int pconv2_launch(const GParams& p, const csmri_gconv_desc* d, hipStream_t st);

// gpipe.hip
int gpipe_eligible(const csmri_gconv_desc* d);
int gpipe_splitk(const csmri_gconv_desc* d);
int gpipe_stats_rows(const csmri_gconv_desc* d);
int gpipe_launch(const GParams& p, const csmri_gconv_desc* d, hipStream_t st);
void gpipe_kernel_name(const csmri_gconv_desc* d, char* buf, int n);

// gconv_glds.hip
int gconv_glds_eligible(const csmri_gconv_desc* d);
int gconv_glds_bn(const csmri_gconv_desc* d);
int gconv_glds_launch(const GParams& p, const csmri_gconv_desc* d, hipStream_t st);
void gconv_glds_kernel_name(const csmri_gconv_desc* d, char* buf, int n);


// gconv_fp8.hip
int gconv_fp8_eligible(const csmri_gconv_desc* d);
int gconv_fp8_bn(const csmri_gconv_desc* d);
int gconv_fp8_launch(const GParams& p, const csmri_gconv_desc* d, hipStream_t st);
void gconv_fp8_kernel_name(const csmri_gconv_desc* d, char* buf, int n);
