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
};

// tconv.hip
int tconv_eligible(const csmri_gconv_desc* d);
int tconv_stats_rows(const csmri_gconv_desc* d);
int tconv_launch(const GParams& p, const csmri_gconv_desc* d, hipStream_t st);
void tconv_kernel_name(const csmri_gconv_desc* d, char* buf, int n);

// gconv_glds.hip
int gconv_glds_eligible(const csmri_gconv_desc* d);
int gconv_glds_bn(const csmri_gconv_desc* d);
int gconv_glds_launch(const GParams& p, const csmri_gconv_desc* d, hipStream_t st);
void gconv_glds_kernel_name(const csmri_gconv_desc* d, char* buf, int n);
