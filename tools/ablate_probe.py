"""Critical-path sensitivity of the C3 step: time it with one kernel class removed (numerically INVALID runs, timing only).
Which classes, if made free, would shorten the step -- i.e. where kernel work is exposed rather than overlapped."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, json
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "csmri-refinement_amd"))
import torch
from csmri_hip import ops, lib
what = os.environ.get("ABLATE", "")
if what == "wgrad":
  def no_wgrad(layer, x0, x1, gy, accumulate=True):
    wgt = layer.weight
    if wgt.grad is None: wgt.grad = torch.zeros_like(wgt)
    if layer.bias is not None and layer.bias.grad is None: layer.bias.grad = torch.zeros_like(layer.bias)
  ops.conv_wgrad = no_wgrad
orig_call = lib.call
skip = {"bn": ("csmri_bn_finalize",), "reduce": ("csmri_gconv_reduce",), "adam": ("csmri_adam",),
        "pack": ("csmri_pack_weight_multi", "csmri_pack_weight"), "dc": ("csmri_dc",),
        "loss": ("csmri_loss",), "pool": ("csmri_maxpool2", "csmri_maxpool2_bwd")}.get(what)
if skip:
  def call(name, *a):
    if any(name.startswith(s) for s in skip): return 0
    return orig_call(name, *a)
  lib.call = call
import bench
sys.argv = ["bench.py", "--steps", "150", "--no-cpu-baseline", "--no-roofline"]
bench.main()
''' % (ROOT, ROOT)
for what in ('', 'wgrad', 'bn', 'adam', 'pack', 'pool', 'dc'):
  env = dict(os.environ, ABLATE=what)
  out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True)
  try:
    d = json.loads(out.stdout.strip().splitlines()[-1])
    print('%-8s %8.3f ms/step' % (what or 'full', d['ms_per_step']))
  except Exception as e:
    print(what, 'failed', out.stderr[-300:])
