"""checkpoint save / restore and the pretrained-weights hand-off that chains the
RecNet run (1-recnet) into the refinement run (2-refinement), reference utils/checkpoints.py:9-41,
96-121 and configs/2-refinement.json:29 -- same pickle layout {'conf','runner','epoch',
'best_val_metrics'}, runner state under 'model' / 'generator','discriminator','gen_optimizer',
'disc_optimizer', state-dict keys of the reference (SURVEY A-12)."""
import os

import pytest
import torch

import csmri_oracle as O
from conftest import PKG, ROOT

pytestmark = pytest.mark.gpu


class Loader(list):
  batch_size = 2


def _confs(tmp_path, dtype='fp32'):
  from utils.config import Configuration
  from models.utils import set_default_compute_dtype
  set_default_compute_dtype(dtype)
  c1 = Configuration.from_json(os.path.join(PKG, 'configs', '1-recnet.json'))
  c1.batch_size = 2
  c1.model['num_blocks'], c1.model['num_filters'], c1.model['compute_dtype'] = 3, 8, dtype
  c2 = Configuration.from_json(os.path.join(PKG, 'configs', '2-refinement.json'))
  c2.batch_size = 2
  c2.vgg_loss = {'seed': 19}
  g, d = c2.generator_model, c2.discriminator_model
  g['pretrained_model'].update(num_filters=8, compute_dtype=dtype)
  g['learnable_model'].update(encode_filters=[8, 16, 32], decode_filters=[16, 8], compute_dtype=dtype)
  d.update(num_filters_per_layer=[8, 16, 32, 64, 64, 64], compute_dtype=dtype)
  return c1, c2


def test_recnet_checkpoint_feeds_refinement_pretrained_model(tmp_path):
  import csmri_hip  # noqa: F401
  from training import build_runner
  from utils.checkpoints import save_checkpoint
  c1, c2 = _confs(tmp_path)
  r1 = build_runner(c1, 'standard', '0', 'train')
  r1.train_epoch(Loader([O.synth_batch(2, 128, 128, acc=4, seed=5)]), 1)
  path = str(tmp_path / 'recnet_0001.pth')
  save_checkpoint(path, c1, r1, 1, None)
  ck = torch.load(path, map_location='cpu', weights_only=False)
  assert sorted(ck) == ['best_val_metrics', 'conf', 'epoch', 'runner'] and 'model' in ck['runner']
  c2.generator_model['pretrained_model']['pretrained_weights'] = [path, 'model']
  r2 = build_runner(c2, 'adversarial', '0', 'train')
  want = r1.model.state_dict()
  got = r2.gen.pretrained_model.state_dict()
  assert sorted(want) == sorted(got)
  for k in want:
    assert torch.equal(want[k].cpu(), got[k].cpu()), k
  assert all(not p.requires_grad for p in r2.gen.pretrained_model.parameters())


def test_adversarial_checkpoint_resume_is_exact(tmp_path):
  """train 1 step -> save -> restore into a fresh runner -> both continue with the same batch
  and the same injected randomness: identical losses and parameters (kernels are deterministic)."""
  import csmri_hip  # noqa: F401
  from training import build_runner
  from utils.checkpoints import save_checkpoint, restore_checkpoint
  _, c2 = _confs(tmp_path)
  batches = [O.synth_batch(2, 128, 128, acc=4, seed=s) for s in (6, 7)]
  a = build_runner(c2, 'adversarial', '0', 'train')
  a.train_epoch(Loader(batches[:1]), 1)
  path = str(tmp_path / 'gan_0001.pth')
  save_checkpoint(path, c2, a, 1, {'psnr': 1.0})
  ck = torch.load(path, map_location='cpu', weights_only=False)
  assert sorted(ck['runner']) == ['disc_optimizer', 'discriminator', 'gen_optimizer', 'generator']
  b = build_runner(c2, 'adversarial', '0', 'train')
  state = restore_checkpoint(path, b)
  assert state['start_epoch'] == 1 and state['best_val_metrics'] == {'psnr': 1.0}
  g = torch.Generator().manual_seed(4)
  chans = [f for _, bn, drop, f in a.disc._layers if bn is not None and drop]
  masks = [(torch.rand(2, c, generator=g) < 0.5).float() * 2.0 for _ in range(3) for c in chans]
  outs = []
  for r in (a, b):
    r.disc.injected_dropout = [m.clone() for m in masks]
    pool = getattr(r.disc_input_fn, 'image_pool', None)
    if pool is not None:
      import random
      random.seed(123)
    torch.manual_seed(99)
    losses, _ = r.train_epoch(Loader(batches[1:]), 2)
    outs.append({k: v.value for k, v in losses.items()})
  assert outs[0] == outs[1], (outs[0], outs[1])
  for (n, p), (_, q) in zip(list(a.gen.state_dict().items()) + list(a.disc.state_dict().items()),
                            list(b.gen.state_dict().items()) + list(b.disc.state_dict().items())):
    assert torch.equal(p.cpu(), q.cpu()), n


@pytest.mark.gpu
def test_resume_from_a_checkpoint_written_by_the_reference(tmp_path):
  """SURVEY 8f-1 against fixtures F13: restore_checkpoint of a file the reference's own save_checkpoint wrote
  (weights, BatchNorm statistics, both Adam states, epoch, best metrics), then what the reference computed
  after saving: the generator's eval-mode prediction (2e-5) and the losses / parameters of the NEXT training
  step (fp32 compute; 1e-4 relative -- the restored exp_avg / exp_avg_sq / step enter this update)."""
  import numpy as np
  import csmri_hip  # noqa: F401
  from csmri_hip import ops
  from utils.config import Configuration
  from utils.checkpoints import restore_checkpoint
  from models.utils import set_default_compute_dtype
  from training import build_runner
  set_default_compute_dtype('fp32')
  golden = os.path.join(ROOT, 'tests', 'golden')
  f = np.load(os.path.join(golden, 'F13_checkpoint_expected.npz'))
  conf = Configuration.from_json(os.path.join(PKG, 'configs', '2-refinement.json'))
  conf.batch_size = 2
  conf.vgg_loss = {'seed': 19}
  g, d = conf.generator_model, conf.discriminator_model
  for m in (g['pretrained_model'], g['learnable_model'], d):
    m['compute_dtype'] = 'fp32'
  g['pretrained_model']['num_filters'] = 8
  g['learnable_model']['encode_filters'] = [8, 16, 32]
  g['learnable_model']['decode_filters'] = [16, 8]
  d['num_filters_per_layer'] = [8, 16, 32, 64, 64, 64]
  runner = build_runner(conf, 'adversarial', '0', 'train')
  state = restore_checkpoint(os.path.join(golden, 'F13_reference_checkpoint.pth'), runner, '0')
  assert state['start_epoch'] == 4 and state['best_val_metrics'] == {'psnr': 31.5}
  assert runner.gen_optimizer.step_count == 1 and runner.disc_optimizer.step_count == 1
  runner._set_test()
  vb = O.synth_batch(2, 128, 128, acc=4, seed=901)
  with torch.no_grad():
    pred = runner.gen(vb['inp'].cuda(), vb['kspace'].cuda(), vb['mask'].cuda())['pred'].float().cpu()
  ref = torch.from_numpy(f['eval_pred'])
  assert float((pred - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
  runner.disc.injected_dropout = [torch.from_numpy(f['step1.mask%d' % j]) for j in range(9)]
  losses, metrics = runner.train_epoch(Loader([O.synth_batch(2, 128, 128, acc=4, seed=902)]), 2)
  for k, v in zip([str(s) for s in f['loss_names']], f['step1.losses']):
    assert abs(losses[k].value - v) < 1e-4 * max(1.0, abs(v)), (k, losses[k].value, v)
  assert abs(metrics['gen_psnr'].value - float(f['step1.gen_psnr'])) < 1e-3
  sd = runner.gen.state_dict()
  for k in f.files:
    if not k.startswith('G2.') or 'num_batches' in k or 'running' in k:
      continue
    v = torch.from_numpy(f[k]).float()
    dd = (sd[k[3:]].cpu().float() - v).abs()
    # second Adam step: every parameter moved by ~lr; agreement to a small fraction of that
    assert float((dd > 2e-5 * max(1.0, float(v.abs().max()))).float().mean()) < 0.02, k
    assert float(dd.max()) < 4.1e-4 * max(1.0, float(v.abs().max())), (k, float(dd.max()))


def test_train_cli_periodic_checkpoint_resumes_at_the_next_epoch(tmp_path):
  """train.py stores the NEXT epoch in its periodic checkpoints, under the reference's file name
  (reference train.py:286-296), so `--resume` continues AFTER the last finished epoch: two epochs, then a resumed run
  whose first epoch is 3 and whose Adam step counter continues where the first run stopped."""
  import glob
  import train
  from utils.checkpoints import restore_checkpoint
  from training import build_runner
  from utils.config import Configuration
  conf = os.path.join(PKG, 'configs', '1-recnet.json')
  run = str(tmp_path / 'run')
  over = ['--conf', 'num_epochs=2', 'steps_per_epoch=2', 'image_size=64', 'batch_size=2', 'compute_dtype=fp32']
  assert train.main([conf, '-c', '0', '--run-dir', run] + over) == 0
  files = sorted(glob.glob(os.path.join(run, 'periodic-chkpt_*.pth')))
  assert len(files) == 2 and files[0].endswith('_1.pth') and files[1].endswith('_2.pth'), files
  ck = torch.load(files[1], map_location='cpu', weights_only=False)
  assert ck['epoch'] == 3                                      # saved_epoch + 1
  c = Configuration.from_json(conf)
  c.update({'batch_size': '2'})
  runner = build_runner(c, 'standard', '0', 'train')
  state = restore_checkpoint(files[1], runner, '0')
  assert state['start_epoch'] == 3 and runner.optimizer.step_count == 4
  # the resumed CLI run trains epoch 3 only (num_epochs=3) and writes exactly one more checkpoint, which stores 4
  over3 = ['--conf', 'num_epochs=3', 'steps_per_epoch=2', 'image_size=64', 'batch_size=2', 'compute_dtype=fp32']
  assert train.main([conf, '-c', '0', '--run-dir', run, '--resume', files[1]] + over3) == 0
  files3 = sorted(glob.glob(os.path.join(run, 'periodic-chkpt_*.pth')))
  assert len(files3) == 3 and len(glob.glob(os.path.join(run, 'periodic-chkpt_*_3.pth'))) == 1, files3
  new = [f for f in files3 if f not in files][0]
  ck3 = torch.load(new, map_location='cpu', weights_only=False)
  assert ck3['epoch'] == 4
  assert int(ck3['runner']['optimizer']['state'][0]['step']) == 6          # 3 epochs x 2 steps, no epoch repeated
