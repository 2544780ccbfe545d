"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/csmri_hip.h declares, rejects bad arguments without touching a GPU, and the
product fails loudly (no CPU fallback)."""
import os
import re

import pytest
import torch

from conftest import ROOT, PKG


def header_symbols():
  txt = open(os.path.join(ROOT, 'include', 'csmri_hip.h')).read()
  txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
  return sorted(set(re.findall(r'\b(csmri_[a-z0-9_]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
  import csmri_hip
  syms = header_symbols()
  assert len(syms) >= 40
  for s in syms:
    assert hasattr(csmri_hip.lib._lib, s), 'libcsmri_hip.so lacks %s' % s
  # and the binding covers the whole header
  assert set(syms) == set(csmri_hip.lib.EXPORTS), set(syms) ^ set(csmri_hip.lib.EXPORTS)
  assert csmri_hip.lib.raw('csmri_version')() >= 100


def test_device_code_has_no_packed_fp32_instructions(tmp_path):
  """csrc/Makefile builds the gfx950 code without v_pk_*_f32 (such kernels
  return wrong lanes next to MFMA kernels of another hipGraph branch).  Disassemble what ships."""
  import glob
  import shutil
  import subprocess
  objdump = '/opt/rocm/lib/llvm/bin/llvm-objdump'
  if not os.path.exists(objdump):
    pytest.skip('llvm-objdump not available')
  lib = shutil.copy(os.path.join(PKG, 'csmri_hip', 'libcsmri_hip.so'), str(tmp_path / 'lib.so'))
  subprocess.run([objdump, '--offloading', 'lib.so'], cwd=str(tmp_path), check=True,
                 stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
  objs = glob.glob(str(tmp_path / 'lib.so.*gfx950*'))
  assert len(objs) >= 8, objs                       # one code object per .hip source
  mfma = 0
  for o in objs:
    asm = subprocess.run([objdump, '-d', o], check=True, capture_output=True, text=True).stdout
    assert not re.search(r'\bv_pk_(add|mul|fma)_f32\b', asm), 'packed fp32 VALU op in %s' % os.path.basename(o)
    mfma += len(re.findall(r'\bv_mfma_f32_16x16x32_bf16\b', asm))
  assert mfma > 500                                 # the disassembly is the real thing


def test_argument_validation_needs_no_gpu():
  import ctypes as C
  import csmri_hip
  lib = csmri_hip.lib
  d = lib.GConvDesc()
  assert lib.raw('csmri_gconv')(C.byref(d), None) == -1            # CSMRI_E_ARG
  w = lib.WGradDesc()
  assert lib.raw('csmri_wgrad')(C.byref(w), None) == -1
  assert lib.raw('csmri_dc')(None, 2, None, None, None, None, 0, None, 1, 64, 64, None) == -1
  assert lib.raw('csmri_adam')(None, None, None, None, 10, 1e-3, 0.9, 0.999, 1e-8, 1, 1.0, None) == -1
  assert b'bad argument' in lib.raw('csmri_error_string')(-1)
  # Cin 2 -> 8 channels, KW 3 -> 4 taps (32/8 taps per K chunk): K = 3*4*8 = 96 -> 128
  assert lib.raw('csmri_pack_weight_bytes')(0, lib.BF16, 32, 2, 3, 3) == 128 * 128 * 2
  assert lib.raw('csmri_pack_weight_bytes')(2, lib.F32, 64, 16, 4, 4) == 4 * 128 * 256 * 4
  assert lib.raw('csmri_bn_stats_rows')(524288, 32) == 1024
  assert lib.raw('csmri_bn_stats_rows')(1024, 1024) == 64


def test_no_cpu_fallback():
  import csmri_hip
  x = torch.zeros(1, 2, 8, 8)
  with pytest.raises(RuntimeError):
    csmri_hip.ops.nchw_to_nhwc(x, torch.bfloat16)
  import utils
  with pytest.raises(RuntimeError):
    utils.device_for('')


def test_product_never_imports_the_oracle():
  bad = []
  for root, _, files in os.walk(PKG):
    for fn in files:
      if fn.endswith('.py'):
        src = open(os.path.join(root, fn)).read()
        if 'csmri_oracle' in src or 'import oracle' in src:
          bad.append(os.path.join(root, fn))
  assert not bad, bad


def test_configuration_semantics(tmp_path):
  from utils.config import Configuration
  base = tmp_path / 'base.json'
  base.write_text('{"seed": 3, "a": 1, "nested": {"x": 1}}')
  top = tmp_path / 'top.json'
  top.write_text('{"#include": "base.json", "b": [1, 2], "include": {"sub": "base.json"}}')
  conf = Configuration.from_json(str(top))
  assert conf.seed == 3 and conf.a == 1 and conf.b == [1, 2] and conf.sub['a'] == 1
  conf.update({'lr': '0.5', 'flag': 'True', 'n': '7', 'lst': '[1, 2.5, x]', 'seed': '9'})
  assert conf.lr == 0.5 and conf.flag is True and conf.n == 7 and conf.lst == [1, 2.5, 'x'] and conf.seed == 9
  assert conf.get_attr('missing', default=4) == 4 and conf.get_attr('missing', alternative='a') == 1
  assert conf.to_param_dict(['a'], ['b', 'zzz'], {'a': 'alpha'}) == {'alpha': 1, 'b': [1, 2]}
  for name in ('1-recnet.json', '2-refinement.json'):
    c = Configuration.from_json(os.path.join(PKG, 'configs', name))
    assert c.runner_type in ('standard', 'adversarial')
  ref = '/root/reference/configs/2-refinement.json'
  if os.path.exists(ref):       # the reference's own config loads unchanged (build container only)
    c = Configuration.from_json(ref)
    assert c.generator_model['learnable_model']['encode_filters'] == [32, 64, 128]


def test_model_state_dict_key_space():
  """SURVEY App. A-12 key space of the full-width models (constructed on CPU)."""
  from utils.config import Configuration
  from models import construct_model
  conf = Configuration.from_json(os.path.join(PKG, 'configs', '2-refinement.json'))
  g = construct_model(Configuration.from_dict(conf.generator_model, conf), 'RefinementWrapper', cuda='0')
  d = construct_model(Configuration.from_dict(conf.discriminator_model, conf), 'CNNDiscriminator')
  gk, dk = set(g.state_dict()), set(d.state_dict())
  assert 'scale' in gk and 'pretrained_model.conv_blocks.2.layers.7.bias' in gk
  assert 'learnable_model.concat_decode_units.1.decode.0.encode.5.weight' in gk
  assert 'learnable_model.encode_units.0.encode.2.running_var' in gk and 'learnable_model.head.0.bias' in gk
  assert {'convs.%d.weight' % i for i in (1, 4, 8, 12, 17, 22)} <= dk and 'convs.1.bias' in dk
  assert {'convs.%d.running_mean' % i for i in (5, 9, 13, 18, 23)} <= dk
  assert {'final_conv.0.weight', 'final_conv.0.bias'} <= dk and 'convs.4.bias' not in dk
  assert sum(p.numel() for p in g.parameters()) == 920034 and len(list(g.parameters())) == 39
  assert sum(p.numel() for p in d.parameters()) == 27941697
  assert float(g.scale) == 0.0
  assert g.pretrained_model.conv_blocks[0].layers['1'].weight.shape == (32, 2, 3, 3)


def test_image_pool_semantics():
  import random
  from utils.image_pool import ImagePool
  import csmri_oracle as O
  random.seed(5)
  a, b = ImagePool(4), O.ImagePool(4)
  for step in range(6):
    x = torch.arange(3 * 2, dtype=torch.float32).reshape(3, 1, 2, 1) + 100 * step
    dec = a.decide(3)
    ra = a.query(x, dec)
    rb = b.query(x, [(u, i) for u, i in dec])
    assert torch.equal(ra, rb), step


def test_image_pool_query_into_preallocated_output():
  """ImagePool.query(..., out=) (the runner hands the fake half of the [fake; real] discriminator batch): same
  result and pool state as the plain query, written into the given tensor (host formulation here; the device
  formulation is csmri_image_pool_exchange, tests/test_hip_ops.py)."""
  import random
  from utils.image_pool import ImagePool
  random.seed(11)
  a, b = ImagePool(3), ImagePool(3)
  for step in range(5):
    x = torch.arange(2 * 4, dtype=torch.float32).reshape(2, 1, 2, 2) + 10 * step
    dec = a.decide(2)
    buf = torch.full((4, 1, 2, 2), -1.0)
    ra = a.query(x, dec, out=buf[:2])
    rb = b.query(x, list(dec))
    assert ra.data_ptr() == buf.data_ptr() and torch.equal(buf[:2], rb) and float(buf[2:].max()) == -1.0
    assert torch.equal(a.buffer, b.buffer) and a.count == b.count


def test_lr_schedulers_match_torch_closed_forms():
  """training/lr_schedulers.py (FlatAdam has no torch optimizer to hand to torch's schedulers) against
  torch.optim.lr_scheduler.MultiStepLR / LambdaLR driven the way the reference drives them
  (reference training/lr_schedulers.py:26-44: constructed with initial_epoch -1, one step() per epoch)."""
  import sys
  import torch
  sys.path.insert(0, PKG)
  from training import lr_schedulers as L

  class Opt(object):
    def __init__(self, lr):
      self.param_groups = [{'lr': lr}]

  class Conf(dict):
    __getattr__ = dict.__getitem__

    def get_attr(self, k, default=None):
      return self.get(k, default)

  for name, conf in (('multistep', Conf(decay_steps=[2, 5], decay_factor=0.3)),
                     ('linear', Conf(learning_rate=2e-4, end_learning_rate=1e-5, decay_steps=6, start_decay=2)),
                     ('polynomial', Conf(learning_rate=2e-4, end_learning_rate=0.0, decay_steps=5, decay_power=2.0))):
    mine = Opt(2e-4)
    sm = L.get_lr_scheduler(conf, name, mine)
    p = torch.nn.Parameter(torch.zeros(1))
    theirs = torch.optim.Adam([p], lr=2e-4)
    if name == 'multistep':
      st = torch.optim.lr_scheduler.MultiStepLR(theirs, conf.decay_steps, conf.decay_factor, -1)
    else:
      lam = L._get_polynomial_decay(conf.learning_rate, conf.end_learning_rate, conf.decay_steps,
                                    conf.get_attr('start_decay', 0), 1.0 if name == 'linear' else conf.decay_power)
      st = torch.optim.lr_scheduler.LambdaLR(theirs, lam, -1)
    for epoch in range(10):
      assert abs(mine.param_groups[0]['lr'] - theirs.param_groups[0]['lr']) < 1e-18 + 1e-12 * 2e-4, (name, epoch)
      assert L.is_pre_epoch_scheduler(sm) and not L.is_post_epoch_scheduler(sm)
      sm.step()
      theirs.step()
      st.step()


def test_weight_init_reproduces_the_reference_bit_for_bit():
  """SURVEY a19 against fixture F12 (written by the reference's own initialize_weights, models/weight_inits.py
  :5-114, under fixed seeds): RecNet conv kaiming_normal(a = 0.01) with zero biases EXCEPT the first conv of each
  block -- xavier-uniform weight and torch's default U(+-1/sqrt(fan_in)) bias, recnet.py:54-59 -- the U-Net's
  orthogonal(gain sqrt 2) convs with BatchNorm (1, 0), the discriminator's N(0, 0.02) convs and N(1, 0.02)
  BatchNorm weights.  Construction draws from torch's CPU generator in the reference's order with the
  reference's initialisers, so every tensor is bit-identical: first 8 values, float64 sum, sum of squares and an
  order-sensitive checksum all equal."""
  import sys
  import numpy as np
  import torch
  sys.path.insert(0, PKG)
  from utils.config import Configuration
  from models import construct_model
  import utils
  f = np.load(os.path.join(ROOT, 'tests', 'golden', 'F12_weight_init.npz'))
  conf = Configuration.from_json(os.path.join(PKG, 'configs', '2-refinement.json'))
  utils.set_random_seeds(7)
  gc = Configuration.from_dict(conf.generator_model, conf)
  gen = construct_model(gc, gc.name)
  utils.set_random_seeds(8)
  dc = Configuration.from_dict(conf.discriminator_model, conf)
  disc = construct_model(dc, dc.get_attr('name', default='CNNDiscriminator'))
  rconf = Configuration.from_json(os.path.join(PKG, 'configs', '1-recnet.json'))
  rconf.model['num_blocks'] = 3          # the reference's shipped value (this repo's copy is set up for config C2: 5)
  utils.set_random_seeds(9)
  rc = Configuration.from_dict(rconf.model, rconf)
  rec = construct_model(rc, rc.name)
  checked = 0
  for tag, model in (('G', gen), ('D', disc), ('R', rec)):
    sd = model.state_dict()
    keys = [k[len(tag) + 1:-len('.digest')] for k in f.files if k.startswith(tag + '.') and k.endswith('.digest')]
    assert sorted(keys) == sorted(k for k in sd if 'num_batches' not in k), tag
    for k in keys:
      v = sd[k].detach().float().cpu()
      assert list(v.shape) == list(f['%s.%s.shape' % (tag, k)]), k
      assert np.array_equal(v.reshape(-1)[:8].numpy(), f['%s.%s.head' % (tag, k)]), (tag, k)
      a = v.double().reshape(-1).numpy()
      i = np.arange(a.size, dtype=np.float64)
      d = np.array([a.sum(), (a * a).sum(), (a * np.cos(0.37 * i)).sum(), a.min(), a.max()])
      assert np.array_equal(d, f['%s.%s.digest' % (tag, k)]), (tag, k, d, f['%s.%s.digest' % (tag, k)])
      checked += 1
  assert checked >= 127
  # the quirk itself, visible in the numbers: first conv of a RecNet block keeps a non-zero default bias
  sd = rec.state_dict()
  assert float(sd['conv_blocks.0.layers.1.bias'].abs().max()) > 0.05
  assert float(sd['conv_blocks.0.layers.4.bias'].abs().max()) == 0.0


def test_reference_written_checkpoint_loads_and_checkpoint_helpers(tmp_path):
  """SURVEY 8f-1 on the CPU side: F13_reference_checkpoint.pth was written by the reference's own
  save_checkpoint (utils/checkpoints.py:9-16).  It unpickles with this package on the path (its Configuration
  becomes utils.config.Configuration here), its model state dicts carry exactly this package's key space, and
  the remaining helpers behave as the reference's: load_model_state_dict errors, inference checkpoints,
  pruning, the run-directory file names."""
  import sys
  import torch
  sys.path.insert(0, PKG)
  from utils import checkpoints as CK, checkpoint_paths as CP
  from utils.config import Configuration
  from models import construct_model
  path = os.path.join(ROOT, 'tests', 'golden', 'F13_reference_checkpoint.pth')
  ck = torch.load(path, map_location='cpu', weights_only=False)
  assert isinstance(ck['conf'], Configuration) and ck['epoch'] == 4 and ck['best_val_metrics'] == {'psnr': 31.5}
  conf = ck['conf']
  gen = construct_model(Configuration.from_dict(conf.generator_model, conf), conf.generator_model['name'])
  disc = construct_model(Configuration.from_dict(conf.discriminator_model, conf), 'CNNDiscriminator')
  gs = CK.load_model_state_dict(path, 'generator')
  ds = CK.load_model_state_dict(path, 'discriminator')
  assert set(gs) == set(k for k in gen.state_dict() if 'num_batches' not in k) | set(k for k in gs if 'num_batches' in k)
  assert set(k for k in ds if 'num_batches' not in k) == set(k for k in disc.state_dict() if 'num_batches' not in k)
  gen.load_state_dict(gs)
  disc.load_state_dict(ds)
  assert abs(float(gen.scale) - 0.25) < 1e-3          # one Adam step away from the preset 0.25
  with pytest.raises(ValueError):
    CK.load_model_state_dict(path, 'model')
  inf = CK.inference_checkpoint_from_training_checkpoint(ck, 'adversarial')
  assert sorted(inf) == ['conf', 'runner'] and list(inf['runner']) == ['generator']
  with pytest.raises(AssertionError):
    CK.inference_checkpoint_from_training_checkpoint(ck, 'standard')
  # run-directory names and pruning
  run = CP.get_run_dir(str(tmp_path), 'refine')
  os.makedirs(run)
  assert os.path.basename(run).startswith('refine_20') and CP.get_run_dir(str(tmp_path), 'refine').endswith('.2')
  names = []
  for epoch in (1, 2, 3):
    p = CP.get_periodic_checkpoint_path(run, epoch)
    assert os.path.basename(p).startswith('periodic-chkpt_20') and p.endswith('_%d.pth' % epoch)
    p = os.path.join(run, 'periodic-chkpt_2026-01-0%d-00-00-00_%d.pth' % (epoch, epoch))
    open(p, 'w').close()
    names.append(os.path.basename(p))
  open(os.path.join(run, 'config_x.json'), 'w').close()
  assert CP.is_checkpoint_path('a.pth') and CP.is_checkpoint_path('a.pth.2') and not CP.is_checkpoint_path('a.json')
  assert os.path.basename(CP.get_best_checkpoint_path(run, 7, 31.23456)).endswith('_7_31.2346.pth')
  CK.prune_checkpoints(run, num_checkpoints_to_retain=1)
  assert sorted(f for f in os.listdir(run) if f.endswith('.pth')) == [names[-1]]
  assert os.path.exists(os.path.join(run, 'config_x.json'))


def test_product_radial_mask_equals_the_reference_sample_for_sample():
  """data.synthetic.radial_mask (the product's own generator for BASELINE config 5 batches) against F10,
  written by the reference's radial_sampling (compressed_sensing.py:568-647): identical sample indices for
  golden-angle and uniform spokes at 64^2, 128^2 and 512^2."""
  import sys
  import numpy as np
  sys.path.insert(0, PKG)
  from data.synthetic import radial_mask, synth_batch_radial
  f = np.load(os.path.join(ROOT, 'tests', 'golden', 'F10_radial.npz'))
  for tag in ('g512', 'u128', 'g64'):
    n, nx, lines, golden = (int(v) for v in f['args_' + tag])
    m = radial_mask((n, nx, nx), lines, rng=np.random.RandomState(4321), golden_angle=bool(golden))
    assert np.array_equal(np.flatnonzero(m), f['idx_' + tag]), tag
  b = synth_batch_radial(2, 64, 64, spokes=12, seed=3)
  assert b['mask'].shape == (2, 2, 64, 64) and set(np.unique(b['mask'].numpy())) == {0.0, 1.0}
  k = b['kspace'].numpy()
  assert float(np.abs(k[b['mask'].numpy() == 0]).max()) == 0.0
