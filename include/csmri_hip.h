/*
 * csmri_hip.h -- C-ABI of libcsmri_hip.so, the MI355X (gfx950) operator library
 * behind the CS-MRI GAN-refinement training path.
 *
 * This is the drop-in boundary.  The reference (mseitzer/csmri-refinement) is
 * pure Python and reaches native code only through torch.nn / pytorch_fft; each
 * entry point below names the reference call site it replaces (path:line in the
 * reference tree).  A Python binding (ctypes) lives in
 * csmri-refinement_amd/csmri_hip/; INTEGRATION.md shows the stub a reference
 * maintainer would add.
 *
 * Conventions
 *   - All pointers are DEVICE pointers unless named host_*.  The caller owns
 *     every buffer including workspaces; the library allocates nothing except
 *     cached FFT twiddle tables (freed by csmri_shutdown).
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous on it.
 *   - Return value: 0 = ok, negative = CSMRI_E_* (bad argument / unsupported),
 *     positive = hipError_t passed through.  No exceptions cross this boundary.
 *   - Activations are NHWC ("pixel-major"): element (b,y,x,c) of a tensor lives
 *     at base[((b*H + y)*W + x)*pix_stride + c].  pix_stride >= C allows channel
 *     slices of wider buffers (concat without copies).  Channel counts seen by
 *     the GEMM kernels are padded to a multiple of 8; pad channels hold zeros.
 *   - dtype: CSMRI_F32 (exact fp32 MFMA path, v_mfma_f32_16x16x4_f32) or
 *     CSMRI_BF16 (v_mfma_f32_16x16x32_bf16, fp32 accumulate).
 */
#ifndef CSMRI_HIP_H
#define CSMRI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CSMRI_F32 0
#define CSMRI_BF16 1
#define CSMRI_FP8 2   /* OCP e4m3fn, per-tensor power-of-two scale: operand type of the fp8 csmri_gconv variant only */
/* A 2-channel image (re, im) in a channel-padded bf16 pixel [8] as a SPLIT pair: channels 0,1 = hi = bf16(v), channels
 * 2,3 = lo = bf16(v - hi), channels 4..7 zero -- 16 significant bits in the 16 bytes a padded bf16 pixel occupies anyway.
 * A convolution whose weights repeat input channels 0,1 on channels 2,3 (csmri_convblock_desc.x_split) multiplies
 * hi + lo; one whose weights are zero there reads plain bf16.  Accepted as the padded-copy type of csmri_dc* and as
 * dst_dtype of csmri_nchw_to_nhwc (C = 2, Cpad = 8); never a compute dtype. */
#define CSMRI_BF16_SPLIT 3

#define CSMRI_OK 0
#define CSMRI_E_ARG (-1)
#define CSMRI_E_UNSUPPORTED (-2)
#define CSMRI_E_ALIGN (-3)

#define CSMRI_BORDER_ZERO 0
#define CSMRI_BORDER_REFLECT 1

/* library / device ------------------------------------------------------- */
int csmri_version(void);
const char* csmri_error_string(int code);
int csmri_shutdown(void);

/* ------------------------------------------------------------------------
 * Gather convolution = implicit-GEMM convolution with MFMA.
 * Replaces nn.Conv2d forward and its input-gradient, together with the
 * padding layer in front of it (models/utils.py:58-85) and optional fused
 * nearest x2 upsampling (models/unet.py:98), bias, LeakyReLU/ReLU
 * (models/recnet.py:40-48, models/unet.py:48-58, models/discriminators.py:137-150,
 * torchvision VGG19 features as used by models/vgg.py:35).
 *
 *   out[b, oy*out_sy+out_oy, ox*out_sx+out_ox, n] =
 *       epilogue( sum_{ty<TH, tx<TW, c<Cin}
 *                 in[b, Y(oy*in_s + dy0 + ty*dy_step), X(ox*in_s + dx0 + tx*dx_step), c]
 *                 * w[n][(ty*TW+tx)*Cin + c] )
 *   Y()/X() apply the border rule on the virtual extent (2*Hin if upsample else
 *   Hin) -- zero: outside -> 0, reflect: mirror without edge repeat -- and then
 *   halve the coordinate if upsample.
 *   epilogue(v) = actgrad( act( v + bias[n] ) ):
 *     act: v<0 ? v*act_slope : v   (act_slope = 1 -> identity, 0 -> ReLU)
 *     actgrad: multiply by (g_src[b,y,x,n] > 0 ? 1 : g_slope) if g_src != NULL
 *   The same routine expresses forward convs, stride-1 dgrad (flipped taps),
 *   and stride-2 dgrad as 4 output-parity classes (nclass = 4: class z uses
 *   weights w + z*w_class_stride and output offset (z>>1, z&1)).
 * ---------------------------------------------------------------------- */
typedef struct csmri_gconv_desc {
  int dtype;                 /* CSMRI_F32 / CSMRI_BF16 / CSMRI_FP8: type of in, w */
  int out_dtype;             /* type of out */
  /* input: channels [0,c0) from in0, [c0,Cin) from in1 (in1 may be NULL) */
  const void* in0; const void* in1;
  int in0_pix_stride, in1_pix_stride, c0;
  int B, Hin, Win, Cin;      /* Cin: total padded input channels (multiple of 8) */
  int upsample;              /* 1: input is virtually nearest-upsampled x2 */
  int border;                /* CSMRI_BORDER_* */
  /* taps */
  int TH, TW, in_s, dy0, dy_step, dx0, dx_step;
  /* weights: [Npad][Kp] row-major, K index = (ty*TW+tx)*Cin + c, zero padded */
  const void* w; int Kp; int nclass; long long w_class_stride;
  /* output */
  void* out; int out_pix_stride; int Hout_t, Wout_t;   /* tensor extents */
  int Ho, Wo;                /* positions computed per image (GEMM M = B*Ho*Wo) */
  int out_sy, out_sx, out_oy, out_ox;
  int Cout;                  /* padded output channels written (multiple of 8) */
  /* epilogue */
  const float* bias;         /* [Cout] or NULL */
  float act_slope;           /* 1 = none */
  const void* g_src; int g_pix_stride; float g_slope; /* dtype = out_dtype... see .c */
  int g_dtype;
  float* stats_partial;      /* NULL or [2][Cout][stats_rows]: per-wave-row sum / sumsq */
  /* split-K */
  int splitk;                /* >=1; >1 needs slab */
  float* slab;               /* [splitk][M][Cout] fp32 workspace */
  int flags;                 /* CSMRI_GCONV_DEFER_REDUCE: caller runs csmri_gconv_reduce itself; CSMRI_GCONV_USE_GPIPE */
  /* optional output window (reflection-padded dgrad without a full fold pass): positions whose
   * tensor coordinate (ty, tx) lies inside [win_y0, win_y0+win_h) x [win_x0, win_x0+win_w) are
   * written to `out`, now a dense [B, win_h, win_w] tensor indexed by (ty-win_y0, tx-win_x0)
   * (g_src is indexed the same way); all other positions go, without actgrad, to `out_halo`
   * ([B, Hout_t, Wout_t] extents, only those positions are touched).  NULL: no window. */
  void* out_halo; int halo_pix_stride; int win_y0, win_x0, win_h, win_w;
  /* dtype == CSMRI_FP8: device scalars (csmri_quantize_fp8's scales[1] of the input and of the packed
   * weights) whose product multiplies the fp32 accumulators before the epilogue; NULL = 1.  The fp8
   * variant needs Cin % 128 == 0, c0 % 16 == 0, Cout % 64 == 0, pixel strides % 16 == 0. */
  const float* in_dequant; const float* w_dequant;
  /* real (un-padded) channel counts of the K and N sides, 0 = unknown.  1 selects the thin-layer kernels
   * (discriminator first layer and its data gradient, U-Net head, discriminator final conv): same result, the
   * seven pad channels are simply not multiplied. */
  int cin_real, cout_real;
  /* ABI 102.  Optional fp8 (OCP e4m3fn) copy of the output for a following fp8 convolution -- the frozen VGG stack
   * (reference models/vgg.py:35) chains its 3 x 3 layers this way --: out_q[pos * out_q_pix_stride + c] =
   * fp8_rne(out[pos][c] * *out_q_scale), i.e. exactly what csmri_quantize_fp8 makes of the stored bf16 output with that
   * scale, written by the producing kernel's epilogue instead of by a pass of its own; *out_amax (device word, optional)
   * receives atomicMax of the bit patterns |out| (what csmri_absmax returns), from which the caller derives the NEXT
   * step's scale (delayed scaling).  Only the 3 x 3 patch kernel takes these (else CSMRI_E_UNSUPPORTED). */
  void* out_q; int out_q_pix_stride; const float* out_q_scale; float* out_amax;
} csmri_gconv_desc;
#define CSMRI_GCONV_DEFER_REDUCE 1
#define CSMRI_GCONV_USE_GPIPE 2     /* take the persistent pipelined gather kernel (gpipe.hip) wherever it is eligible, not only where it measured faster */
#define CSMRI_GCONV_NO_GPIPE 16     /* never take it (A/B against gconv_glds); 4 / 8: force its 256- / 192-row tile (tools/sk_sweep.py) */

int csmri_gconv(const csmri_gconv_desc* d, void* stream);
/* second stage of a split-K launch (slab sum + epilogue); no-op when splitk <= 1 */
int csmri_gconv_reduce(const csmri_gconv_desc* d, void* stream);
/* name of the kernel instance csmri_gconv launches for d, as profilers print it (for reports) */
int csmri_gconv_kernel_name(const csmri_gconv_desc* d, char* buf, int n);
/* rows of stats_partial written by csmri_gconv for this problem (0 if splitk>1) */
int csmri_gconv_stats_rows(const csmri_gconv_desc* d);
size_t csmri_gconv_slab_bytes(const csmri_gconv_desc* d);
/* heuristic split-K factor for this shape (>=1) */
int csmri_gconv_suggest_splitk(const csmri_gconv_desc* d);

/* Weight packing.  w_ref: fp32 [Cout][Cin][KH][KW] (nn.Conv2d layout, state-dict
 * contract SURVEY A-12).  mode 0: forward taps; mode 1: stride-1 dgrad
 * (roles of Cout/Cin swapped, taps kept -- the descriptor flips them with
 * dy_step=-1); mode 2: stride-2 dgrad, 4 parity classes of (KH/2 x KW/2) taps.
 * Output rows padded to a multiple of 128, K padded to a multiple of 64, zeros.
 * mode 3: stride-1 dgrad with FLIPPED taps (K index (ty,tx) holds w[..][KH-1-ty][KW-1-tx]), so
 * the input-gradient is a plain correlation (dy_step = dx_step = +1, dy0 = pad_top-(KH-1)).
 * For 8- or 16-channel K sides the packed filter width TW is KW rounded up to a multiple of
 * 32/channels (zero taps); *TW_out returns it and is what the descriptor's TW must be. */
size_t csmri_pack_weight_bytes(int mode, int dtype, int Cout, int Cin, int KH, int KW);
int csmri_pack_weight(int mode, int dtype, const float* w_ref, int Cout, int Cin,
                      int KH, int KW, void* out, int* Kp_out, long long* class_stride_out,
                      int* TW_out, void* stream);

/* Multi-tensor form: one launch packs every item of a table that lives in DEVICE memory
 * (a network's weights are re-packed after each optimizer step). */
typedef struct csmri_pack_item {
  const float* w; void* out;
  int mode, dtype, Cout, Cin, KH, KW;
  const float* bias; float* bias_out;   /* optional: bias[0..Cout) copied to bias_out (the layer's zero-padded fp32
                                           bias buffer) by the same launch; mode < 0 = bias only, w/out unused */
} csmri_pack_item;
int csmri_pack_weight_multi(const csmri_pack_item* items_dev, int n, void* stream);

/* ------------------------------------------------------------------------
 * One RecNet conv block as a single launch (SURVEY 8b `csmri_convblock_fused_fwd`):
 *   [ZeroPad(1) -> Conv3x3 + bias -> LeakyReLU(slope)] x (num_convs-1) -> ZeroPad(1) -> Conv3x3 + bias
 * (reference models/recnet.py:29-62, padding rule models/utils.py:75-85), intermediates kept in LDS, halo rings
 * recomputed.  Supported: bf16, num_convs 3, num_filters 32, kernel 3, 2 -> 2 channels, zero padding
 * (csmri_convblock_fused_supported); anything else: the per-layer csmri_gconv path.
 * x: [B,H,W,>=8] bf16, channels 0,1; w[i] / Kp[i]: csmri_pack_weight(mode 0, CSMRI_BF16) of layer i; bias[i]:
 * fp32, zero padded to >= 32 / 32 / 16 entries; out: [B,H,W,>=8] of out_dtype, channels 0..7 written (2..7 zero).
 * act[0], act[1] (both or neither): if set, the activations after layer 1 and 2, [B,H,W,>=32] bf16, are also
 * written (what the data-gradient / weight-gradient kernels of the backward pass read).
 * ---------------------------------------------------------------------- */
typedef struct csmri_convblock_desc {
  int dtype;
  int num_convs, num_filters, kernel_size, num_inputs, num_outputs, border;
  const void* x; int x_pix_stride;
  int B, H, W;
  const void* w[3]; int Kp[3];
  const float* bias[3];
  float slope;
  void* act[2]; int act_pix_stride[2];
  void* out; int out_dtype; int out_pix_stride;       /* [B,H,W,8] (out_pix_stride >= 8), or out_pix_stride == 2 with
                                                         fp32: the dense interleaved complex image [B,H,W,2] that
                                                         DataConsistencyInKspace.perform (myfft.py:145-163) consumes */
  int x_split;                                        /* x is CSMRI_BF16_SPLIT: layer 1 multiplies hi + lo */
} csmri_convblock_desc;
int csmri_convblock_fused_supported(const csmri_convblock_desc* d);
int csmri_convblock_fused_fwd(const csmri_convblock_desc* d, void* stream);

/* ------------------------------------------------------------------------
 * The BACKWARD pass of the same block as a single launch (SURVEY 8b `csmri_convblock_fused_bwd`): what
 * `loss.backward()` (reference training/runner.py:163) runs through models/recnet.py:29-62 -- three data-gradient
 * convolutions, two LeakyReLU derivatives, three weight gradients and three bias gradients -- with the
 * intermediate gradients kept in LDS:
 *   dA2 = conv3x3(dY, W3 flipped) * lrelu'(a2);  dA1 = conv3x3(dA2, W2 flipped) * lrelu'(a1);
 *   dX = conv3x3(dA1, W1 flipped);  dW3 += a2^T dY, dW2 += a1^T dA2, dW1 += x^T dA1;  db_l += sum dY_l
 * x, act[0] (a1), act[1] (a2): the block input and the saved activations of csmri_convblock_fused_fwd (bf16);
 * gy: the gradient of the block output, fp32 dense complex [B,H,W,2] (gy_pix_stride 2) or [B,H,W,>=8] bf16 / fp32;
 * wd[i] / Kp[i]: csmri_pack_weight(mode 3, CSMRI_BF16) of layer i; dx: [B,H,W,>=8] bf16 (channels 0..7 written,
 * 2..7 zero) or NULL when the block input needs no gradient.
 * Weight / bias gradients leave as `splits` slabs per layer in csmri_wgrad's slab format -- slab[i]:
 * fp32 [splits][Cout_p][KH*KW*Cin_p] followed by [splits][Cout_p] bias partial rows (Cout_p, Cin_p = 32, 8 / 32,
 * 32 / 8, 32 for i = 0, 1, 2; csmri_wgrad_slab_bytes of the layer's descriptor with splitk = splits) -- and are
 * reduced into the fp32 reference-layout gradients by csmri_wgrad_finish_multi on the three layers' descriptors
 * (splitk = splits, defer_finish = 2).  splits: workgroups launched (csmri_convblock_fused_bwd_splits).
 * Same support matrix as the forward (bf16, 3 convs, 32 filters, kernel 3, 2 -> 2 channels, zero padding).
 * ---------------------------------------------------------------------- */
typedef struct csmri_convblock_bwd_desc {
  int dtype;
  int num_convs, num_filters, kernel_size, num_inputs, num_outputs, border;
  const void* x; int x_pix_stride;
  int B, H, W;
  const void* act[2]; int act_pix_stride[2];
  const void* gy; int gy_dtype; int gy_pix_stride;
  const void* wd[3]; int Kp[3];
  float slope;
  void* dx; int dx_pix_stride;
  float* slab[3]; int splits; int want_db;
  int x_split;        /* x is CSMRI_BF16_SPLIT: the weight gradient of layer 1 sums the hi and the lo products */
  int dx_split;       /* write dx as CSMRI_BF16_SPLIT (channels 2,3 = what the bf16 rounding of channels 0,1 dropped):
                         csmri_dc_in_bf16, the adjoint that consumes it, adds the two */
} csmri_convblock_bwd_desc;
int csmri_convblock_fused_bwd(const csmri_convblock_bwd_desc* d, void* stream);
int csmri_convblock_fused_bwd_splits(int B, int H, int W);

/* ------------------------------------------------------------------------
 * fp8 operand preparation (BASELINE.json config 5: "fp8 MFMA convs").  The reference has no fp8 path;
 * the variant computes the same nn.Conv2d (models/unet.py:40-52, models/discriminators.py,
 * models/vgg.py) on operands rounded to OCP e4m3fn with one power-of-two scale per tensor:
 *   e = floor(log2(amax)), scale = 2^(7-e), q = e4m3_rne(x * scale)   (|x*scale| < 256, never saturates)
 * csmri_absmax: amax[0] = max |x| over n contiguous elements (device scalar, NaNs skipped).
 * csmri_quantize_fp8: q[i] = e4m3(x[i] * scale); scales[0] = scale, scales[1] = 1/scale (device,
 * may be NULL).  n % 16 == 0.  Weights: pack with csmri_pack_weight(mode, CSMRI_F32, ...) and quantise
 * the packed buffer (same [rows][Kp] indexing).
 * ---------------------------------------------------------------------- */
int csmri_absmax(int dtype, const void* x, long long n, float* amax, void* stream);
int csmri_quantize_fp8(int dtype, const void* x, void* q, long long n, const float* amax, float* scales,
                       void* stream);
/* Delayed scaling for a chain of fp8 tensors (ABI 102; the frozen VGG stack): amax[j] = the |x| maximum of tensor j its
 * producers accumulated this step (csmri_gconv_desc.out_amax, csmri_maxpool2_q); scales[2j], scales[2j+1] become the
 * quantisation / dequantisation pair of the NEXT step, 2^(7 - floor(log2 amax) - margin) and its inverse; amax[j] is cleared
 * (a tensor that saw no data keeps its scales).  0 <= margin <= 4. */
int csmri_fp8_scales_update(float* amax, float* scales, int n, int margin, void* stream);

/* ------------------------------------------------------------------------
 * Weight gradient of a convolution (nn.Conv2d backward w.r.t. weight):
 *   dW[n][c][ky][kx] += sum_{b,oy,ox} dY[b,oy,ox,n] * Xb[b, oy*s+ky-pt, ox*s+kx-pl, c]
 * with the same border / upsample / two-source gather as csmri_gconv.
 * Split over pixels (splitk slabs, deterministic), result accumulated into the
 * fp32 reference-layout gradient.  db (bias grad) optional.
 * ---------------------------------------------------------------------- */
typedef struct csmri_wgrad_desc {
  int dtype;                 /* type of x and dy */
  const void* in0; const void* in1;
  int in0_pix_stride, in1_pix_stride, c0;
  int B, Hin, Win, Cin;      /* padded Cin */
  int upsample, border;
  int KH, KW, stride, pad_t, pad_l;
  const void* dy; int dy_pix_stride; int Ho, Wo; int Cout; /* padded Cout */
  int Cin_real, Cout_real;
  float* dw;                 /* fp32 [Cout_real][Cin_real][KH][KW], accumulated into */
  float* db;                 /* fp32 [Cout_real] accumulated into, or NULL */
  int splitk; float* slab;   /* [splitk][CoutPad][KH*KW*Cin] fp32 */
  int accumulate;            /* 0: overwrite dw/db, 1: add */
  int defer_finish;          /* 1: leave the slab reduction into dw (and the patch kernels' bias partials) to
                                csmri_wgrad_finish_multi; the slab must stay alive until then.  2 (descriptors handed
                                to csmri_wgrad_finish_multi only): the slabs AND one bias partial row per split were
                                produced by csmri_convblock_fused_bwd */
} csmri_wgrad_desc;

int csmri_wgrad(const csmri_wgrad_desc* d, void* stream);
/* the deferred slab reductions of n csmri_wgrad calls (their descriptors, by value, in call order) in one launch per
 * 24 layers: a backward pass (reference adversarial_runner.py:314-320 `loss.backward()`) ends with one reduction
 * launch instead of one or two small launches behind every layer's weight-gradient kernel.  No two descriptors may
 * share dw. */
int csmri_wgrad_finish_multi(const csmri_wgrad_desc* descs, int n, void* stream);
size_t csmri_wgrad_slab_bytes(const csmri_wgrad_desc* d);
int csmri_wgrad_suggest_splitk(const csmri_wgrad_desc* d);
/* name of the kernel instance csmri_wgrad launches for d, as profilers print it (for reports/tests) */
int csmri_wgrad_kernel_name(const csmri_wgrad_desc* d, char* buf, int n);

/* Fold the gradient w.r.t. a reflect-padded (and optionally x2-upsampled)
 * tensor back onto the un-padded tensor (backward of nn.ReflectionPad2d +
 * nn.Upsample(nearest), models/utils.py:58-72, unet.py:98) and optionally apply
 * the activation derivative of the producer.  gpad: [B, (up?2H:H)+pt+pb,
 * (up?2W:W)+pl+pr, C] dense; out: [B,H,W,C] with out_pix_stride. */
/* border-only companion of the csmri_gconv output window: adds to dx [B,H,W,C] (in place) the
 * gradient that reflection padding (pads pt,pb,pl,pr) mirrors back from the halo positions of
 * gpad_halo [B,H+pt+pb,W+pl+pr,C]; multiplied by lrelu'(g_src) when g_src != NULL.  Touches
 * O(perimeter) pixels.  Needs H >= pt+pb+2 and W >= pl+pr+2. */
int csmri_fold_halo(int dtype, const void* gpad_halo, void* dx, int dx_pix_stride, int B, int H, int W,
                    int C, int pt, int pb, int pl, int pr, const void* g_src, int g_pix_stride,
                    float g_slope, void* stream);
int csmri_fold_pad_grad(int dtype, const void* gpad, void* out, int out_pix_stride,
                        int B, int H, int W, int C, int pt, int pb, int pl, int pr,
                        int upsample, const void* g_src, int g_pix_stride, float g_slope,
                        void* stream);

/* ------------------------------------------------------------------------
 * Data consistency in k-space (fused 2-D FFT + mask merge + inverse FFT).
 * Replaces DataConsistencyInKspace.perform, Fft2d/Ifft2d and data_consistency
 * (data/reconstruction/deep_med_lib/my_pytorch/myfft.py:78-163):
 *     out = orthoIFFT2( (1 - m) * orthoFFT2(x) + k0 )
 * x: interleaved complex fp32, pixel p at x + p*x_pix_stride (2 = dense; 8 reads
 * channels 0,1 of a channel-padded conv output directly); k0, out: dense
 * interleaved complex fp32 [B][H][W][2]; mask: uint8 [B][H][W]
 * (1 = sampled).  k0 == NULL gives the adjoint (backward w.r.t. x,
 * myfft.py:92-102,119-128).  H, W in {32,64,128,256,512}.  work: csmri_dc_work_bytes(B,H,W) bytes
 * (currently 0: the passes run in place on `out`; NULL is accepted then).
 * out_pad (optional): also write the result as a channel-padded NHWC tensor
 * [B][H][W][8] of dtype out_pad_dtype (channels 0,1 = re,im; 2..7 = 0) -- the
 * input layout of the next conv block.
 * ---------------------------------------------------------------------- */
int csmri_dc(const float* x, int x_pix_stride, const float* k0, const uint8_t* mask, float* out,
             void* out_pad, int out_pad_dtype, float* work, int B, int H, int W,
             void* stream);
/* forward model of a training sample on the device (rec_transforms.py:18-57,
 * compressed_sensing.py:460-512): kspace = m * orthoFFT2(img), inp = orthoIFFT2(kspace);
 * img, kspace, inp interleaved complex fp32 [B,H,W,2], mask uint8 [B,H,W]; H, W powers of two
 * in [32, 512]. */
int csmri_undersample(const float* img, const uint8_t* mask, float* kspace, float* inp, int B,
                      int H, int W, void* stream);
size_t csmri_dc_work_bytes(int B, int H, int W);
/* stand-alone batched 2-D FFT / inverse FFT, interleaved complex fp32 [B,H,W,2] -> same (x == out allowed):
 * the transform behind the reference's Fft2d / Ifft2d autograd Functions (myfft.py:78-128; their backward
 * passes, :92-102,119-128, are this call with `inverse` flipped).  ortho != 0: 1/sqrt(HW) in both directions
 * (normalized=True); ortho == 0: forward unscaled, inverse 1/(HW) (pytorch_fft.fft2 / ifft2).
 * H, W powers of two in [32, 512]. */
int csmri_fft2(const float* x, float* out, int B, int H, int W, int inverse, int ortho, void* stream);
/* bf16 image storage ("bf16 cFFT", BASELINE config 5): the same passes with x, out and the intermediate
 * between them held as interleaved complex bf16 (4 bytes per value; x_pix_stride in bf16 elements, 8 reads
 * channels 0,1 of a channel-padded bf16 conv output); arithmetic fp32 in registers / LDS; k0 stays fp32.
 * The reference's transform (myfft.py:78-163) is fp32; stated tolerance of this storage format:
 * 1e-2 of the output's maximum (tests/test_hip_ops.py). */
int csmri_fft2_bf16(const void* x, void* out, int B, int H, int W, int inverse, int ortho, void* stream);
int csmri_dc_bf16(const void* x, int x_pix_stride, const float* k0, const uint8_t* mask, void* out,
                  void* out_pad, int out_pad_dtype, int B, int H, int W, void* stream);
/* csmri_dc (fp32 arithmetic, intermediate and output) with the input image read as bf16, x_pix_stride in bf16
 * elements: the adjoint of DataConsistencyInKspace.perform (myfft.py:145-163 under autograd, k0 = NULL) applied
 * straight to the channel-padded bf16 gradient of the next conv block's input.  x_dtype declares the format of x:
 * CSMRI_BF16 reads channels 0,1 (whatever the pad channels hold), CSMRI_BF16_SPLIT reads channels (0,1) + (2,3) of a
 * pixel of >= 4 channels -- a gradient the fused backward wrote with dx_split (ABI 101: the format used to be inferred
 * from the stride). */
int csmri_dc_in_bf16(const void* x, int x_dtype, int x_pix_stride, const float* k0, const uint8_t* mask, float* out,
                     void* out_pad, int out_pad_dtype, int B, int H, int W, void* stream);

/* layout converters (H2D boundary: batch dict tensors are NCHW fp32,
 * training/base_runner.py:29-41) */
int csmri_nchw_to_nhwc(const float* src, int B, int C, int H, int W, void* dst,
                       int dst_dtype, int dst_pix_stride, int Cpad, void* stream);
int csmri_nhwc_to_nchw(const void* src, int src_dtype, int src_pix_stride, int B,
                       int C, int H, int W, float* dst, void* stream);
/* dst[B,H,W,Cpad] (dst_dtype) = zero-padded NHWC form of src [B,C,H,W] fp32 + add (NULL, or NHWC of add_dtype with
 * >= Cpad channels per pixel): the gradient of a tensor that left the library twice -- as the NCHW fp32 API tensor
 * (the discriminator's `logits`, reference models/discriminators.py:236-247) and as its device-layout feature map --
 * converted, summed and rounded to the conv's dtype in one launch. */
int csmri_nchw_to_nhwc_add(const float* src, int B, int C, int H, int W, void* dst, int dst_dtype,
                           int dst_pix_stride, int Cpad, const void* add, int add_dtype, int add_pix_stride,
                           void* stream);
/* mask [B,2,H,W] fp32 {0,1} -> uint8 [B,H,W] (bit-exact; returns E_ARG semantics
 * are checked on the host side of the binding) */
int csmri_mask_to_u8(const float* mask_nchw, int B, int H, int W, uint8_t* dst, void* stream);

/* ------------------------------------------------------------------------
 * BatchNorm2d (training) + LeakyReLU + Dropout2d mask, NHWC.
 * Replaces nn.BatchNorm2d / nn.LeakyReLU / nn.Dropout2d as sequenced in
 * models/unet.py:48-58,100-118 and models/discriminators.py:129-155.
 *
 * `groups` (>= 1): the batch is `groups` equal, consecutive sub-batches that are
 * normalised independently -- the result equals `groups` separate module calls
 * (own batch statistics each; running statistics updated once per group, in order),
 * which lets the two discriminator passes of training/adversarial_runner.py:333-341
 * (fake, then real) run as one launch sequence.  mean/invstd are [groups][C]; partial
 * sums are channel-major, [2][C][rows] with the rows group-major (the finalize kernels read a
 * channel's rows contiguously).
 * ---------------------------------------------------------------------- */
/* per-channel partial sums of a tensor: partial [2][C][rows] (rows returned by
 * csmri_bn_stats_rows) -- used when the conv epilogue did not produce them */
int csmri_bn_stats_rows(int npix, int C);
int csmri_bn_stats(int dtype, const void* y, int pix_stride, int npix, int C,
                   float* partial, int groups, void* stream);   /* rows = groups*stats_rows(npix/groups) */
/* reduce partials -> mean, invstd (saved for backward); update running stats
 * (momentum, unbiased var) when running_mean != NULL.  C_real channels. */
int csmri_bn_finalize(const float* partial, int rows, int C, int C_real, long long count,
                      float eps, float momentum, float* mean, float* invstd,
                      float* running_mean, float* running_var, int groups, void* stream);
/* z = dropmask[b,c] * lrelu( (y-mean)*invstd*gamma + beta ) ; dropmask NULL = 1.
 * eval mode: pass running stats as mean and 1/sqrt(var+eps) as invstd.
 * affine_snap (NULL or [2][C] fp32) receives gamma and beta as this forward used them.
 * z must not alias y (nor dy an input of the backward passes): E_ARG. */
int csmri_bn_act(int dtype, const void* y, int y_pix_stride, void* z, int z_pix_stride,
                 int B, int HW, int C, int C_real, const float* mean, const float* invstd,
                 const float* gamma, const float* beta, float slope,
                 const float* dropmask, float* affine_snap, int groups, void* stream);
/* backward, pass 1: partial sums of dyh = dz*mask*lrelu'(z) and dyh*xhat.
 * z may be NULL when affine_snap (from csmri_bn_act) is given: the activation sign is then
 * recomputed from y with the forward's own arithmetic, saving one tensor read.
 * dz2 (NULL or a second gradient of z, same dtype; both passes must get the same one): where z has
 * two consumers (the next layer and a feature-matching loss, models/criteria.py of the reference via
 * discriminators.py:118-126) the two gradients are summed in fp32 inside these passes instead of by a
 * separate add launch. */
int csmri_bn_bwd_reduce(int dtype, const void* dz, int dz_pix_stride, const void* y,
                        int y_pix_stride, const void* z, int z_pix_stride, int B, int HW,
                        int C, const float* mean, const float* invstd, float slope,
                        const float* dropmask, float* partial, const float* affine_snap,
                        int groups, const void* dz2, int dz2_pix_stride, void* stream);
/* pass 2: finalize dgamma/dbeta (accumulated into fp32 grads if not NULL) and
 * write dy = gamma*invstd*(dyh - mean(dyh) - xhat*mean(dyh*xhat)).
 * partial: the rows of pass 1 plus `groups` extra rows that receive the totals. */
int csmri_bn_bwd_apply(int dtype, const void* dz, int dz_pix_stride, const void* y,
                       int y_pix_stride, const void* z, int z_pix_stride, void* dy,
                       int dy_pix_stride, int B, int HW, int C, int C_real,
                       const float* mean, const float* invstd, const float* gamma,
                       float slope, const float* dropmask, const float* partial, int rows,
                       float* dgamma, float* dbeta, int accumulate, const float* affine_snap,
                       int groups, const void* dz2, int dz2_pix_stride, void* stream);

/* Small feature maps (one group, C a power of two >= 16, B*HW <= 512: csmri_bn_small_ok != 0 -- the discriminator's
 * 8 x 8 x 1024 layers, reference models/discriminators.py:118-126, on an 8-image pass): the whole training-mode forward
 * (csmri_bn_stats + csmri_bn_finalize + csmri_bn_act) resp. backward (csmri_bn_bwd_reduce + csmri_bn_bwd_apply with
 * affine_snap) as ONE launch each; a workgroup owns 8 or 16 channels and all pixels, so the channel sums need no second
 * stage.  Same arguments and results as the calls they replace (sums in a different, still fixed, order). */
int csmri_bn_small_ok(int npix, int C, int groups);
int csmri_bn_small_fwd(int dtype, const void* y, int y_pix_stride, void* z, int z_pix_stride, int B, int HW,
                       int C, int C_real, const float* gamma, const float* beta, float slope,
                       const float* dropmask, float eps, float momentum, float* mean, float* invstd,
                       float* running_mean, float* running_var, float* affine_snap, void* stream);
int csmri_bn_small_bwd(int dtype, const void* dz, int dz_pix_stride, const void* dz2, int dz2_pix_stride,
                       const void* y, int y_pix_stride, void* dy, int dy_pix_stride, int B, int HW, int C,
                       int C_real, const float* mean, const float* invstd, const float* gamma, float slope,
                       const float* dropmask, const float* affine_snap, float* dgamma, float* dbeta,
                       int accumulate, void* stream);

/* dy = (dz + dz2) * lrelu'(z)  (where no BN follows the conv); dz2: NULL or a second gradient of z */
int csmri_act_bwd(int dtype, const void* dz, int dz_pix_stride, const void* z, int z_pix_stride,
                  void* dy, int dy_pix_stride, long long npix, int C, float slope, const void* dz2,
                  int dz2_pix_stride, void* stream);

/* MaxPool2d(2,2) NHWC (models/unet.py:58, torchvision VGG19) */
int csmri_maxpool2(int dtype, const void* x, int x_pix_stride, void* y, int y_pix_stride,
                   uint8_t* argmax, int B, int H, int W, int C, void* stream);
/* the same (bf16) with an fp8 (e4m3fn) copy of the pooled output for a following fp8 convolution: y_q = what
 * csmri_quantize_fp8 makes of y with scale *q_scale; *amax (optional) = atomicMax of the bit patterns |y| (ABI 102) */
int csmri_maxpool2_q(int dtype, const void* x, int x_pix_stride, void* y, int y_pix_stride,
                     uint8_t* argmax, int B, int H, int W, int C, void* y_q, int y_q_pix_stride,
                     const float* q_scale, float* amax, void* stream);
int csmri_maxpool2_bwd(int dtype, const void* dy, int dy_pix_stride, const uint8_t* argmax,
                       void* dx, int dx_pix_stride, int B, int H, int W, int C, void* stream);
/* same with the neighbouring elementwise steps of the backward folded into the one pass over the
 * full-resolution gradient:  dx = (unpool(dy) + g_add) * act'(g_src).
 * g_add (optional, [B,H,W,C]): the other gradient of the pool's input -- the skip connection of the U-Net
 * encoder (models/unet.py:58-72: the block output feeds both the pool and the decoder's concat);
 * g_src (optional): output of the activated layer that fed the pool, derivative 1 where g_src > 0 else
 * g_slope -- the VGG19 backward's pool -> relu' step (torchvision features: conv, ReLU, MaxPool2d). */
int csmri_maxpool2_bwd_act(int dtype, const void* dy, int dy_pix_stride, const uint8_t* argmax,
                           void* dx, int dx_pix_stride, int B, int H, int W, int C, const void* g_src,
                           int g_pix_stride, float g_slope, const void* g_add, int g_add_pix_stride,
                           void* stream);

/* The activated-producer form with the POOLED tensor as the gate (g_pooled = the pool's output [B,H/2,W/2,C]): where the
 * gradient is routed the producer's output is the pooled value itself, elsewhere dx is zero either way -- the same bits
 * as csmri_maxpool2_bwd_act with the full-resolution g_src, for a quarter of its bytes. */
int csmri_maxpool2_bwd_pooled_gate(int dtype, const void* dy, int dy_pix_stride, const uint8_t* argmax,
                                   void* dx, int dx_pix_stride, int B, int H, int W, int C,
                                   const void* g_pooled, int g_pix_stride, float g_slope, void* stream);

/* ------------------------------------------------------------------------
 * small fused ops of the refinement wrapper / losses / optimizer
 * ---------------------------------------------------------------------- */
/* complex_abs (utils/tensor_transforms.py:62-75): x [B,H,W,2] fp32 -> |x| written
 * to n_out channel-padded outputs.  mode 0: 1 real channel; mode 3: 3 equal
 * channels normalised (v-mean[c])/std[c] (VGG input, models/vgg.py:66-72). */
int csmri_complex_abs(const float* x, long long npix, void* out, int out_dtype,
                      int out_pix_stride, int Cpad, int mode, void* stream);
/* backward: dx[p] = x[p]/|x[p]| * sum_c g[p][c]*scale[c] */
int csmri_complex_abs_bwd(const float* x, long long npix, const void* g, int g_dtype,
                          int g_pix_stride, int nch, int mode, float* dx, int accumulate,
                          void* stream);
/* per-sample min and max-of-shifted of channel 0 of x [B,HW,2]:
 * minmax[b] = (min, max(x-min))   (models/refinement_wrapper.py:51-73).
 * minmax holds csmri_minmax_floats(B) floats: the [B][2] result, then two-stage scratch. */
size_t csmri_minmax_floats(int B);
int csmri_minmax_real(const float* x, int B, long long HW, float* minmax, void* stream);
/* pred = cat( unscale(scale(pre_real) + s*u), pre_imag ); scaled = s*u
 * (models/refinement_wrapper.py:169-194).  u: [B,HW] values with pix stride. */
int csmri_refine_combine(const float* pre, const void* u, int u_dtype, int u_pix_stride,
                         const float* scale_param, const float* minmax, int B, long long HW,
                         float* pred, float* scaled, void* stream);
/* backward: du = gpred_real * s*max/2, dscale_partial[0] = sum gpred_real*u*max/2.  du_pix_stride == 8 (16-byte aligned
 * du): the pixel is written whole -- the value and seven zero pad channels --, wider pixels must be cleared by the caller.
 * The gradient fan-ins around this node are summed here instead of by add launches (each NULL = absent):
 *   gpred2  a second gradient of pred (its two consumers: the discriminator input and the VGG loss), added to gpred;
 *   du2     a second gradient of u, du_dtype with pixel stride du2_pix_stride, channel 0 (the feature penalty on the
 *           raw refinement, reference models/criteria.py feature_penalty), added to du;
 *   dscale  the scale parameter's fp32 gradient: receives the sum (accumulate != 0: is added to). */
int csmri_refine_combine_bwd(const float* gpred, const void* u, int u_dtype, int u_pix_stride,
                             const float* scale_param, const float* minmax, int B,
                             long long HW, void* du, int du_dtype, int du_pix_stride,
                             float* dscale_partial, const float* gpred2, const void* du2,
                             int du2_pix_stride, float* dscale, int accumulate, void* stream);

/* deterministic two-stage reductions; result[0] = mean over n_real elements.
 * kind: 0 = L1 |a-b|, 1 = MSE (a-b)^2.  b == NULL means b = 0.
 * a, b: NHWC [npix][C] with pix strides; only channels < C_real are counted. */
int csmri_loss(int kind, int dtype, const void* a, int a_pix_stride, const void* b,
               int b_pix_stride, long long npix, int C_real, float* result,
               float* work, void* stream);
size_t csmri_loss_work_bytes(void);
/* grad wrt a: g = coeff[0]*w * d/da ; written (or accumulated) into ga.  ga must not alias a or b (E_ARG);
 * npix * C / 4 < 2^31 (E_UNSUPPORTED). */
int csmri_loss_bwd(int kind, int dtype, const void* a, int a_pix_stride, const void* b,
                   int b_pix_stride, long long npix, int C, int C_real, const float* coeff,
                   float weight, void* ga, int ga_pix_stride, int accumulate, void* stream);
/* several mean losses in ONE launch pair: result[0] = sum_i weight_i * mean_i, result[1+i] = mean_i
 * (FeatureMatchingLoss, models/adversarial_loss.py:133-160: one L1 per discriminator layer).
 * items is a HOST array (copied into the kernel arguments), n <= CSMRI_LOSS_MAX_ITEMS;
 * result: 1+n floats; work: csmri_loss_multi_work_bytes(n).  The backward writes
 * ga_i = coeff[0] * weight_i / count_i * d(|a-b| or (a-b)^2)/da into items[i].ga (C padded channels). */
#define CSMRI_LOSS_MAX_ITEMS 16
typedef struct csmri_loss_item {
  const void* a; const void* b;          /* b may be NULL (= 0) */
  int a_pix_stride, b_pix_stride;
  long long npix; int C, C_real;
  float weight;
  void* ga; int ga_pix_stride;           /* backward only */
  int dtype_plus1;                       /* 0: the call's dtype; else CSMRI_F32 + 1 / CSMRI_BF16 + 1 for this item's
                                            tensors (a discriminator's feature list mixes bf16 maps and fp32 logits) */
} csmri_loss_item;
size_t csmri_loss_multi_work_bytes(int n);
int csmri_loss_multi(int kind, int dtype, const csmri_loss_item* items, int n, float* result,
                     float* work, void* stream);
int csmri_loss_multi_bwd(int kind, int dtype, const csmri_loss_item* items, int n, const float* coeff,
                         void* stream);
/* BCE on sigmoid(logits) with constant target t (models/adversarial_loss.py:71-98,
 * F.binary_cross_entropy clamps log at -100): result[0] = mean. */
int csmri_bce_logits(const float* logits, long long n, float target, float* prob,
                     float* result, void* stream);
int csmri_bce_logits_bwd(const float* logits, long long n, float target, const float* coeff,
                         float weight, float* glogits, int accumulate, void* stream);
/* PSNR (metrics/image_metrics.py:7-19 with rec_transforms.py:79-85): per image
 * mse of clamp(|.|,0,1); mse[b] = mse_b for b < B (host takes 10*log10(1/mse)); mse must
 * hold B*33 floats (the tail is the per-image partial-sum workspace). */
int csmri_psnr_mse(const float* pred, const float* target, int B, long long HW,
                   float* mse, void* stream);
/* SSIM validation metric (metrics/image_metrics.py:22-42 -> metrics/pytorch_ssim/__init__.py:22-42,
 * with the output transform of rec_transforms.py:79-85): per image, on clamp(|.|,0,1) of the
 * interleaved-complex [B,H,W,2] fp32 inputs, 11x11 gaussian window (sigma 1.5), zero padding;
 * ssim[b] = mean of the SSIM map.  work: csmri_ssim_work_bytes(B,H,W). */
size_t csmri_ssim_work_bytes(int B, int H, int W);
int csmri_ssim(const float* pred, const float* target, int B, int H, int W, float* ssim, void* work,
               void* stream);

/* Adam (training/optimizers.py:19-22 -> torch.optim.Adam, eps 1e-8, no decay) on a
 * flat fp32 parameter buffer; step is the 1-based step count. */
int csmri_adam(float* p, const float* g, float* m, float* v, long long n, float lr,
               float beta1, float beta2, float eps, int step, float grad_scale,
               void* stream);

/* same update, step count in device memory (*step_dev steps already taken; incremented by
 * the call): graph-capture safe */
int csmri_adam_dev(float* p, const float* g, float* m, float* v, long long n, float lr,
                   float beta1, float beta2, float eps, int* step_dev, float grad_scale,
                   void* stream);

/* csmri_adam_dev with the learning rate in device memory as well (*lr_dev, fp32): nothing a learning-rate scheduler
 * (reference training/lr_schedulers.py, stepped by the runners' epoch hooks, training/adversarial_runner.py:267-305)
 * changes is a launch argument, so a captured hipGraph follows the schedule without being captured again -- the host
 * writes *lr_dev on the stream ahead of the replay. */
int csmri_adam_dev_lr(float* p, const float* g, float* m, float* v, long long n, const float* lr_dev,
                      float beta1, float beta2, float eps, int* step_dev, float grad_scale,
                      void* stream);

/* One query of the discriminator's history pool of generated images (reference utils/image_pool.py:8-60,
 * called from the discriminator input function, training/adversarial_training.py) as ONE launch:
 *   out[i]              = kind[i] == 1 ? pool[pool_idx[i]] : kind[i] == 2 ? x[x_idx[i]] : x[i]
 *   pool[write_slot[j]] = x[write_src[j]]        (every read of the old pool content precedes the writes)
 * x, out: n images of bytes_per_image bytes (any dtype / layout, a multiple of 16 bytes, 16-byte aligned);
 * pool: pool_size + 1 such images (the last one is a write-only dummy); plan: DEVICE int64 [5][n] rows
 * kind, pool_idx, x_idx, write_slot, write_src -- the reference's sequential python-random decisions resolved
 * into indices by the host (utils/image_pool.py ImagePool.plan).  n <= 64. */
int csmri_image_pool_exchange(const void* x, void* pool, void* out, const long long* plan, int n,
                              long long bytes_per_image, void* stream);

/* Scalar glue of a training step, one launch each (they were 3-9 framework launches inside the captured step).
 * csmri_weighted_sum: out[0] = sum_i w[i] * v[i][0] in list order, fp32 -- the runner's
 *   `torch.sum(torch.cat(losses) * weights)` (reference training/adversarial_runner.py:314-320, runner.py:159-161);
 *   csmri_weighted_sum_bwd: out[i] = g[0] * w[i].
 * csmri_bce_logits_pair: result[0] = mean BCE(sigmoid(l[0:n]), t_first) + mean BCE(sigmoid(l[n:2n]), t_second) (torch's
 *   log clamp at -100), result[1], result[2] the two terms: the discriminator GAN loss on the [fake; real] logits of
 *   one batched pass (models/adversarial_loss.py:71-85); _bwd: glogits[i] = coeff[0] * (sigmoid(l[i]) - t) / n.
 * csmri_psnr_mean: out[0] = mean_b 10 log10(1 / mse[b]) in double (metrics/image_metrics.py:7-19).
 * csmri_disc_accuracy: fraction of images whose mean probability is classified correctly, fake images against
 *   label 0 and / or real images against label 1 (metrics/scalar_metrics.py:26-53); either pointer may be NULL;
 *   B <= 64 images of n_per_image probabilities each. */
#define CSMRI_SCALAR_LIST_MAX 16
typedef struct csmri_scalar_list {
  const float* v[CSMRI_SCALAR_LIST_MAX]; float w[CSMRI_SCALAR_LIST_MAX]; int n;
} csmri_scalar_list;
int csmri_weighted_sum(const csmri_scalar_list* items, float* out, void* stream);
int csmri_weighted_sum_bwd(const csmri_scalar_list* items, const float* g, float* out, void* stream);
int csmri_bce_logits_pair(const float* logits, long long n_half, float t_first, float t_second, float* result,
                          void* stream);
int csmri_bce_logits_pair_bwd(const float* logits, long long n_half, float t_first, float t_second,
                              const float* coeff, float* glogits, void* stream);
int csmri_psnr_mean(const float* mse, int B, float* out, void* stream);
int csmri_disc_accuracy(const float* prob_fake, const float* prob_real, int B, int n_per_image, float* out,
                        void* stream);

/* nn.Dropout2d masks of one discriminator forward pass (reference models/discriminators.py:150-152: one
 * Bernoulli(1 - p) draw per (image, channel), survivors scaled by 1 / (1 - p)), all dropout layers in ONE launch:
 *   mask[i] = keep_i / (1 - p),  keep_i = [u_i < 1 - p],
 *   u_i = (Philox4x32-10(counter = (i / 4, 0, call_lo, call_hi), key = (seed_lo, seed_hi))[i % 4] >> 8) * 2^-24
 * state: DEVICE uint64[3] = {seed, call, 0} (ABI 101; two words before: a caller that still passes two gets an
 * out-of-bounds atomic -- check csmri_version() >= 101); the launch increments `call` (hipGraph-replay safe: every replay draws
 * new masks, and an eager run draws the same sequence); the third word counts the launch's workgroups as they finish
 * (the last one advances `call`) and is zero between launches.  mask: 16-byte aligned.  The masks are APPLIED by csmri_bn_act / the BatchNorm
 * backward (`dropmask`); tests inject masks there directly.  n <= 2^24. */
int csmri_dropout2d_mask(float* mask, long long n, float p, unsigned long long* state, void* stream);

/* Gradient-bucket transport of the data-parallel step (replaces the gradient reduction of the reference's
 * nn.DataParallel wrapper, utils/custom_data_parallel.py:26-35 and utils/__init__.py:59-68).  A sub-bucket
 * g[0..n) of a model's flat fp32 gradient buffer travels as bf16 over two RCCL collectives issued by the host
 * side (training/distributed.py GradBucket: all_to_all of the `world` chunks, all_gather of the reduced chunks):
 *   csmri_bucket_pack_bf16    send[i] = bf16(g[i]) (RNE) for i < n, 0 for n <= i < n_padded (n_padded % 8 == 0,
 *                             = world * per: rank r's chunk is send[r*per .. (r+1)*per))
 *   csmri_bucket_reduce       mine[j] = bf16( sum_{r < world} float(recv[r*per + j]) ), the sum carried in fp32
 *                             in rank order (per % 8 == 0): every element of the result is
 *                             bf16(sum_r bf16(g_r)) whatever the rank count
 *   csmri_bucket_unpack_bf16  g[i] = float(src[i]) for i < n
 * All pointers 16-byte aligned device memory; launches are asynchronous on `stream`. */
int csmri_bucket_pack_bf16(const float* g, long long n, void* send, long long n_padded, void* stream);
int csmri_bucket_reduce(const void* recv, int world, long long per, void* mine, void* stream);
int csmri_bucket_unpack_bf16(const void* src, long long n, float* g, void* stream);

/* misc */
int csmri_fill_f32(float* p, long long n, float v, void* stream);
int csmri_cast(const void* src, int src_dtype, void* dst, int dst_dtype, long long n, void* stream);
/* dst[p*dst_ps + c] = (c < C_src ? src[p*src_ps + c] : 0) for c < C_dst: channel
 * slice / zero-pad / dtype cast of an NHWC tensor in one pass */
int csmri_copy_channels(const void* src, int src_dtype, int src_pix_stride, int C_src, void* dst,
                        int dst_dtype, int dst_pix_stride, int C_dst, long long npix, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CSMRI_HIP_H */
