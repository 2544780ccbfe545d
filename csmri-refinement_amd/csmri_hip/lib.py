"""ctypes binding of libcsmri_hip.so (C-ABI declared in include/csmri_hip.h).

The library is the product; there is NO fallback.  Importing this module without
the built shared object raises, and every entry point raises RuntimeError on a
non-zero status.  Build with ``python __graft_entry__.py`` (or ``make -C csrc``).
"""
import ctypes as C
import os

# torch first: the library must bind to the HIP runtime torch has loaded (streams and
# device pointers are shared with it); loading libamdhip64 on our own beforehand gives
# this process a second runtime that sees no device.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# CSMRI_HIP_LIB: diagnostic builds of the SAME library (tools/stamp_*.py: in-kernel phase stamps)
LIB_PATH = os.environ.get('CSMRI_HIP_LIB') or os.path.join(_HERE, 'libcsmri_hip.so')

F32, BF16 = 0, 1
BF16_SPLIT = 3      # CSMRI_BF16_SPLIT: 2-channel image as bf16 hi + lo in a padded pixel (include/csmri_hip.h)
BORDER_ZERO, BORDER_REFLECT = 0, 1

if not os.path.exists(LIB_PATH):
  raise ImportError('libcsmri_hip.so not found at %s -- build it first '
                    '(python __graft_entry__.py); there is no CPU fallback' % LIB_PATH)
_lib = C.CDLL(LIB_PATH)

vp, i32, i64, f32, sz = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_size_t

ABI = 103           # csmri_version() of the library these structures mirror (103: csmri_adam_dev_lr; 102: csmri_gconv_desc.out_q .. out_amax)
_lib.csmri_version.restype = C.c_int
if _lib.csmri_version() != ABI:
  raise RuntimeError('%s reports ABI %d, this binding is for %d: rebuild the library (python __graft_entry__.py)'
                     % (LIB_PATH, _lib.csmri_version(), ABI))


class GConvDesc(C.Structure):
  _fields_ = [
      ('dtype', i32), ('out_dtype', i32),
      ('in0', vp), ('in1', vp),
      ('in0_pix_stride', i32), ('in1_pix_stride', i32), ('c0', i32),
      ('B', i32), ('Hin', i32), ('Win', i32), ('Cin', i32),
      ('upsample', i32), ('border', i32),
      ('TH', i32), ('TW', i32), ('in_s', i32), ('dy0', i32), ('dy_step', i32),
      ('dx0', i32), ('dx_step', i32),
      ('w', vp), ('Kp', i32), ('nclass', i32), ('w_class_stride', i64),
      ('out', vp), ('out_pix_stride', i32), ('Hout_t', i32), ('Wout_t', i32),
      ('Ho', i32), ('Wo', i32),
      ('out_sy', i32), ('out_sx', i32), ('out_oy', i32), ('out_ox', i32),
      ('Cout', i32),
      ('bias', vp), ('act_slope', f32),
      ('g_src', vp), ('g_pix_stride', i32), ('g_slope', f32), ('g_dtype', i32),
      ('stats_partial', vp),
      ('splitk', i32), ('slab', vp), ('flags', i32),
      ('out_halo', vp), ('halo_pix_stride', i32), ('win_y0', i32), ('win_x0', i32), ('win_h', i32),
      ('win_w', i32),
      ('in_dequant', vp), ('w_dequant', vp),
      ('cin_real', i32), ('cout_real', i32),
      ('out_q', vp), ('out_q_pix_stride', i32), ('out_q_scale', vp), ('out_amax', vp),
  ]


class ConvBlockDesc(C.Structure):
  _fields_ = [
      ('dtype', i32),
      ('num_convs', i32), ('num_filters', i32), ('kernel_size', i32), ('num_inputs', i32), ('num_outputs', i32),
      ('border', i32),
      ('x', vp), ('x_pix_stride', i32),
      ('B', i32), ('H', i32), ('W', i32),
      ('w', vp * 3), ('Kp', i32 * 3),
      ('bias', vp * 3),
      ('slope', f32),
      ('act', vp * 2), ('act_pix_stride', i32 * 2),
      ('out', vp), ('out_dtype', i32), ('out_pix_stride', i32),
      ('x_split', i32),
  ]


class ConvBlockBwdDesc(C.Structure):
  _fields_ = [
      ('dtype', i32),
      ('num_convs', i32), ('num_filters', i32), ('kernel_size', i32), ('num_inputs', i32), ('num_outputs', i32),
      ('border', i32),
      ('x', vp), ('x_pix_stride', i32),
      ('B', i32), ('H', i32), ('W', i32),
      ('act', vp * 2), ('act_pix_stride', i32 * 2),
      ('gy', vp), ('gy_dtype', i32), ('gy_pix_stride', i32),
      ('wd', vp * 3), ('Kp', i32 * 3),
      ('slope', f32),
      ('dx', vp), ('dx_pix_stride', i32),
      ('slab', vp * 3), ('splits', i32), ('want_db', i32),
      ('x_split', i32), ('dx_split', i32),
  ]


class ScalarList(C.Structure):
  _fields_ = [('v', vp * 16), ('w', f32 * 16), ('n', i32)]


class PackItem(C.Structure):
  _fields_ = [('w', vp), ('out', vp), ('mode', i32), ('dtype', i32), ('Cout', i32), ('Cin', i32),
              ('KH', i32), ('KW', i32), ('bias', vp), ('bias_out', vp)]


class LossItem(C.Structure):
  _fields_ = [('a', vp), ('b', vp), ('a_pix_stride', i32), ('b_pix_stride', i32), ('npix', i64),
              ('C', i32), ('C_real', i32), ('weight', f32), ('ga', vp), ('ga_pix_stride', i32),
              ('dtype_plus1', i32)]


class WGradDesc(C.Structure):
  _fields_ = [
      ('dtype', i32),
      ('in0', vp), ('in1', vp),
      ('in0_pix_stride', i32), ('in1_pix_stride', i32), ('c0', i32),
      ('B', i32), ('Hin', i32), ('Win', i32), ('Cin', i32),
      ('upsample', i32), ('border', i32),
      ('KH', i32), ('KW', i32), ('stride', i32), ('pad_t', i32), ('pad_l', i32),
      ('dy', vp), ('dy_pix_stride', i32), ('Ho', i32), ('Wo', i32), ('Cout', i32),
      ('Cin_real', i32), ('Cout_real', i32),
      ('dw', vp), ('db', vp),
      ('splitk', i32), ('slab', vp),
      ('accumulate', i32),
      ('defer_finish', i32),
  ]


_SIGS = {
    'csmri_version': (i32, []),
    'csmri_error_string': (C.c_char_p, [i32]),
    'csmri_shutdown': (i32, []),
    'csmri_gconv': (i32, [C.POINTER(GConvDesc), vp]),
    'csmri_gconv_reduce': (i32, [C.POINTER(GConvDesc), vp]),
    'csmri_gconv_kernel_name': (i32, [C.POINTER(GConvDesc), C.c_char_p, i32]),
    'csmri_gconv_stats_rows': (i32, [C.POINTER(GConvDesc)]),
    'csmri_gconv_slab_bytes': (sz, [C.POINTER(GConvDesc)]),
    'csmri_gconv_suggest_splitk': (i32, [C.POINTER(GConvDesc)]),
    'csmri_pack_weight_bytes': (sz, [i32, i32, i32, i32, i32, i32]),
    'csmri_pack_weight': (i32, [i32, i32, vp, i32, i32, i32, i32, vp, C.POINTER(i32),
                                C.POINTER(i64), C.POINTER(i32), vp]),
    'csmri_pack_weight_multi': (i32, [vp, i32, vp]),
    'csmri_absmax': (i32, [i32, vp, i64, vp, vp]),
    'csmri_convblock_fused_supported': (i32, [vp]),
    'csmri_convblock_fused_fwd': (i32, [vp, vp]),
    'csmri_convblock_fused_bwd': (i32, [vp, vp]),
    'csmri_convblock_fused_bwd_splits': (i32, [i32, i32, i32]),
    'csmri_quantize_fp8': (i32, [i32, vp, vp, i64, vp, vp, vp]),
    'csmri_wgrad': (i32, [C.POINTER(WGradDesc), vp]),
    'csmri_wgrad_finish_multi': (i32, [C.POINTER(WGradDesc), i32, vp]),
    'csmri_wgrad_slab_bytes': (sz, [C.POINTER(WGradDesc)]),
    'csmri_wgrad_suggest_splitk': (i32, [C.POINTER(WGradDesc)]),
    'csmri_wgrad_kernel_name': (i32, [C.POINTER(WGradDesc), C.c_char_p, i32]),
    'csmri_fold_pad_grad': (i32, [i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32,
                                  i32, vp, i32, f32, vp]),
    'csmri_fold_halo': (i32, [i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, i32, f32, vp]),
    'csmri_dc': (i32, [vp, i32, vp, vp, vp, vp, i32, vp, i32, i32, i32, vp]),
    'csmri_undersample': (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    'csmri_dc_work_bytes': (sz, [i32, i32, i32]),
    'csmri_fft2': (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    'csmri_fft2_bf16': (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    'csmri_dc_bf16': (i32, [vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    'csmri_dc_in_bf16': (i32, [vp, i32, i32, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    'csmri_nchw_to_nhwc': (i32, [vp, i32, i32, i32, i32, vp, i32, i32, i32, vp]),
    'csmri_nhwc_to_nchw': (i32, [vp, i32, i32, i32, i32, i32, i32, vp, vp]),
    'csmri_nchw_to_nhwc_add': (i32, [vp, i32, i32, i32, i32, vp, i32, i32, i32, vp, i32, i32, vp]),
    'csmri_mask_to_u8': (i32, [vp, i32, i32, i32, vp, vp]),
    'csmri_bn_stats_rows': (i32, [i32, i32]),
    'csmri_bn_stats': (i32, [i32, vp, i32, i32, i32, vp, i32, vp]),
    'csmri_bn_finalize': (i32, [vp, i32, i32, i32, i64, f32, f32, vp, vp, vp, vp, i32, vp]),
    'csmri_bn_act': (i32, [i32, vp, i32, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, f32,
                           vp, vp, i32, vp]),
    'csmri_bn_bwd_reduce': (i32, [i32, vp, i32, vp, i32, vp, i32, i32, i32, i32, vp, vp, f32,
                                  vp, vp, vp, i32, vp, i32, vp]),
    'csmri_bn_bwd_apply': (i32, [i32, vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, i32, i32,
                                 vp, vp, vp, f32, vp, vp, i32, vp, vp, i32, vp, i32, vp, i32, vp]),
    'csmri_bn_small_ok': (i32, [i32, i32, i32]),
    'csmri_bn_small_fwd': (i32, [i32, vp, i32, vp, i32, i32, i32, i32, i32, vp, vp, f32, vp, f32, f32, vp, vp, vp, vp,
                                 vp, vp]),
    'csmri_bn_small_bwd': (i32, [i32, vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, vp, vp, vp, f32, vp, vp,
                                 vp, vp, i32, vp]),
    'csmri_act_bwd': (i32, [i32, vp, i32, vp, i32, vp, i32, i64, i32, f32, vp, i32, vp]),
    'csmri_maxpool2': (i32, [i32, vp, i32, vp, i32, vp, i32, i32, i32, i32, vp]),
    'csmri_maxpool2_bwd_pooled_gate': (i32, [i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp, i32, f32, vp]),
    'csmri_maxpool2_q': (i32, [i32, vp, i32, vp, i32, vp, i32, i32, i32, i32, vp, i32, vp, vp, vp]),
    'csmri_fp8_scales_update': (i32, [vp, vp, i32, i32, vp]),
    'csmri_maxpool2_bwd': (i32, [i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp]),
    'csmri_maxpool2_bwd_act': (i32, [i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp, i32, f32, vp, i32, vp]),
    'csmri_complex_abs': (i32, [vp, i64, vp, i32, i32, i32, i32, vp]),
    'csmri_complex_abs_bwd': (i32, [vp, i64, vp, i32, i32, i32, i32, vp, i32, vp]),
    'csmri_minmax_floats': (sz, [i32]),
    'csmri_minmax_real': (i32, [vp, i32, i64, vp, vp]),
    'csmri_refine_combine': (i32, [vp, vp, i32, i32, vp, vp, i32, i64, vp, vp, vp]),
    'csmri_refine_combine_bwd': (i32, [vp, vp, i32, i32, vp, vp, i32, i64, vp, i32, i32, vp,
                                       vp, vp, i32, vp, i32, vp]),
    'csmri_loss': (i32, [i32, i32, vp, i32, vp, i32, i64, i32, vp, vp, vp]),
    'csmri_loss_work_bytes': (sz, []),
    'csmri_loss_multi_work_bytes': (sz, [i32]),
    'csmri_loss_multi': (i32, [i32, i32, C.POINTER(LossItem), i32, vp, vp, vp]),
    'csmri_loss_multi_bwd': (i32, [i32, i32, C.POINTER(LossItem), i32, vp, vp]),
    'csmri_loss_bwd': (i32, [i32, i32, vp, i32, vp, i32, i64, i32, i32, vp, f32, vp, i32, i32,
                             vp]),
    'csmri_bce_logits': (i32, [vp, i64, f32, vp, vp, vp]),
    'csmri_bce_logits_bwd': (i32, [vp, i64, f32, vp, f32, vp, i32, vp]),
    'csmri_psnr_mse': (i32, [vp, vp, i32, i64, vp, vp]),
    'csmri_ssim_work_bytes': (sz, [i32, i32, i32]),
    'csmri_ssim': (i32, [vp, vp, i32, i32, i32, vp, vp, vp]),
    'csmri_adam': (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, i32, f32, vp]),
    'csmri_adam_dev': (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, vp, f32, vp]),
    'csmri_adam_dev_lr': (i32, [vp, vp, vp, vp, i64, vp, f32, f32, f32, vp, f32, vp]),
    'csmri_image_pool_exchange': (i32, [vp, vp, vp, vp, i32, i64, vp]),
    'csmri_weighted_sum': (i32, [C.POINTER(ScalarList), vp, vp]),
    'csmri_weighted_sum_bwd': (i32, [C.POINTER(ScalarList), vp, vp, vp]),
    'csmri_bce_logits_pair': (i32, [vp, i64, f32, f32, vp, vp]),
    'csmri_bce_logits_pair_bwd': (i32, [vp, i64, f32, f32, vp, vp, vp]),
    'csmri_psnr_mean': (i32, [vp, i32, vp, vp]),
    'csmri_disc_accuracy': (i32, [vp, vp, i32, i32, vp, vp]),
    'csmri_dropout2d_mask': (i32, [vp, i64, f32, vp, vp]),
    'csmri_bucket_pack_bf16': (i32, [vp, i64, vp, i64, vp]),
    'csmri_bucket_reduce': (i32, [vp, i32, i64, vp, vp]),
    'csmri_bucket_unpack_bf16': (i32, [vp, i64, vp, vp]),
    'csmri_fill_f32': (i32, [vp, i64, f32, vp]),
    'csmri_cast': (i32, [vp, i32, vp, i32, i64, vp]),
    'csmri_copy_channels': (i32, [vp, i32, i32, i32, vp, i32, i32, i32, i64, vp]),
}

EXPORTS = sorted(_SIGS)

for _name, (_res, _args) in _SIGS.items():
  _fn = getattr(_lib, _name)          # AttributeError if the .so lacks a declared symbol
  _fn.restype = _res
  _fn.argtypes = _args


def error_string(code):
  s = _lib.csmri_error_string(code)
  return s.decode() if s else 'code %d' % code


# bench.py: HIP-event brackets around the HBM-bound entry points, with their ALGORITHMIC bytes
# (the bytes any implementation has to move: each operand tensor once).
HBM_PROFILE = None      # list of (label, bytes, start_event, end_event) while profiling


def _es(dt):
  return 4 if dt == 0 else 2


def _dc_bytes(a):
  # csmri_dc(x, xs, k0, mask, out, out_pad, pad_dt, work, B, H, W): read x, k0 (forward only), the
  # uint8 mask; write out (+ the channel-padded copy when requested)
  b, h, w = a[8], a[9], a[10]
  n = b * h * w
  return n * 8 * (3 if a[2] else 2) + n + (n * 8 * _es(a[6]) if a[5] else 0)


HBM_BYTES = {
    'csmri_dc': ('dc (3 passes)', _dc_bytes),
    # (dt, y, ys, z, zs, b, hw, cp, ...): read y, write z
    'csmri_bn_act': ('bn_act_kernel', lambda a: 2 * a[5] * a[6] * a[7] * _es(a[0])),
    # (dt, gz, gzs, y, ys, r, rs, b, hw, cp, ...): read gz, y
    'csmri_bn_bwd_reduce': ('bn_bwd_reduce_kernel', lambda a: 2 * a[7] * a[8] * a[9] * _es(a[0])),
    # (dt, gz, gzs, y, ys, r, rs, gy, gys, b, hw, cp, ...): read gz, y; write gy
    'csmri_bn_bwd_apply': ('bn_bwd_apply_kernel', lambda a: 3 * a[9] * a[10] * a[11] * _es(a[0])),
    # (p, g, m, v, n, ...): read p, g, m, v; write p, m, v
    'csmri_adam_dev': ('adam_dev_kernel', lambda a: 7 * a[4] * 4),
    'csmri_adam_dev_lr': ('adam_dev_kernel', lambda a: 7 * a[4] * 4),
    'csmri_adam': ('adam_kernel', lambda a: 7 * a[4] * 4),
}


class CsmriError(RuntimeError):
  """A non-zero status of a C-ABI entry point; ``code`` is the status (CSMRI_E_* or a HIP error code)."""

  def __init__(self, name, code):
    super(CsmriError, self).__init__('%s failed: %s (%d)' % (name, error_string(code), code))
    self.entry, self.code = name, int(code)


E_UNSUPPORTED = -2      # CSMRI_E_UNSUPPORTED (include/csmri_hip.h)


def call(name, *args):
  """Call a status-returning entry point; raise CsmriError (a RuntimeError) on failure."""
  if HBM_PROFILE is not None and name in HBM_BYTES:
    import torch
    label, fn = HBM_BYTES[name]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = getattr(_lib, name)(*args)
    e1.record()
    HBM_PROFILE.append((label, int(fn(args)), e0, e1))
  else:
    rc = getattr(_lib, name)(*args)
  if rc != 0:
    raise CsmriError(name, rc)


def raw(name):
  return getattr(_lib, name)
