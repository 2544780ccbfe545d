"""Adversarial (GAN refinement) runner -- reference training/adversarial_runner.py.

One training step (reference :322-389):
  out_gen = G(inp, kspace, mask)
  D phase : D(|pred|.detach() via image pool), D(|target|)  -> GAN_disc
  G phase : D(|pred|) again with gradient               -> GAN_gen, FeatureMatching
            VGG19 perceptual loss, FeaturePenalty
  update D (zero_grad, backward, Adam), then update G (backward THROUGH the already
  updated D, Adam).
Semantics kept (SURVEY A-3/A-4/A-5/A-7): three train-mode D forwards (BN running
stats move three times, three dropout draws); the generator backward sees D's
post-update weights and BN affine parameters with the pre-update saved activations
and batch statistics (torch-0.3.1 behaviour, "faithful" ordering (A)); D weight
gradients of the generator pass are not computed (the reference discards them).

Scheduling differs (results do not): the D backward is issued as soon as the D losses
exist, its gradient bucket is all-reduced on RCCL's stream while the third D forward
and the two VGG forwards run, and loss scalars stay on the device (one readback per
logging interval instead of one blocking .data[0] per loss)."""
from collections import OrderedDict
import contextlib
import logging

import torch

import utils
from metrics import get_metric_fn, get_loss_metric, MaxMetric, SharedVec, VecRef
from models import construct_model
from models.criteria import get_criterion
from training.adversarial_training import get_discriminator_input_fn
from training.base_runner import BaseRunner
from training.optimizers import get_optimizer
from training.lr_schedulers import get_lr_scheduler, is_pre_epoch_scheduler, is_post_epoch_scheduler
from csmri_hip import ops
from training import distributed as dist_utils
from utils.checkpoints import initialize_pretrained_model
from utils.config import Configuration


def build_runner(conf, cuda, mode):
  gen_conf = Configuration.from_dict(conf.generator_model, conf)
  gen_model = construct_model(gen_conf, gen_conf.name, cuda)
  val_metric_fns = {name: get_metric_fn(conf, name, cuda, 'test')
                    for name in conf.get_attr('validation_metrics', default=[])}
  if mode != 'train':
    gen_model = utils.cudaify(gen_model, cuda)
    return AdversarialRunner(gen_model, cuda=cuda, val_metric_fns=val_metric_fns)

  disc_dict = dict(conf.discriminator_model)
  disc_dict.setdefault('name', 'CNNDiscriminator')      # shipped config lacks it (SURVEY A-1)
  disc_conf = Configuration.from_dict(disc_dict, conf)
  disc_model = construct_model(disc_conf, disc_conf.name, cuda)

  gen_adv = OrderedDict((n, get_criterion(conf, n, cuda, loss_type='gen'))
                        for n in conf.generator_adversarial_losses)
  gen_crit = OrderedDict((n, get_criterion(conf, n, cuda)) for n in conf.generator_losses)
  disc_adv = OrderedDict((n, get_criterion(conf, n, cuda, loss_type='disc'))
                         for n in conf.discriminator_losses)
  gen_model, disc_model = utils.cudaify([gen_model, disc_model], cuda)
  for crit in list(gen_adv.values()) + list(gen_crit.values()) + list(disc_adv.values()):
    utils.cudaify(crit, cuda)
  if gen_conf.has_attr('pretrained_weights'):
    initialize_pretrained_model(gen_conf, gen_model, cuda, conf.file)
  if disc_conf.has_attr('pretrained_weights'):
    initialize_pretrained_model(disc_conf, disc_model, cuda, conf.file)
  dist_utils.broadcast_module(gen_model)
  dist_utils.broadcast_module(disc_model)
  dist_utils.decorrelate_rng_streams(conf.seed)

  gen_opt_conf = Configuration.from_dict(conf.generator_optimizer, conf)
  disc_opt_conf = Configuration.from_dict(conf.discriminator_optimizer, conf)
  # Important: construct optimizers after moving the models to the GPU
  gen_optimizer = get_optimizer(gen_opt_conf, gen_opt_conf.name, gen_model.parameters())
  disc_optimizer = get_optimizer(disc_opt_conf, disc_opt_conf.name, disc_model.parameters())
  gen_lr_scheduler = disc_lr_scheduler = None      # reference adversarial_runner.py:58-73
  if gen_opt_conf.has_attr('lr_scheduler'):
    gen_lr_scheduler = get_lr_scheduler(gen_opt_conf, gen_opt_conf.lr_scheduler, gen_optimizer)
  if disc_opt_conf.has_attr('lr_scheduler'):
    disc_lr_scheduler = get_lr_scheduler(disc_opt_conf, disc_opt_conf.lr_scheduler, disc_optimizer)

  train_gen_metric_fns = {n: get_metric_fn(conf, n, cuda, 'train')
                          for n in conf.get_attr('train_generator_metrics', default=[])}
  train_disc_metric_fns = {n: get_metric_fn(conf, n, cuda, 'train')
                           for n in conf.get_attr('train_discriminator_metrics', default=[])}
  disc_input_fn = get_discriminator_input_fn(conf, disc_conf, dtype_fn=lambda: disc_model.dtype)
  val_disc_input_fn = get_discriminator_input_fn(conf, disc_conf, no_pool=True,
                                                 dtype_fn=lambda: disc_model.dtype)
  return AdversarialRunner(gen_model, disc_model, gen_optimizer, disc_optimizer,
                           gen_lr_scheduler, disc_lr_scheduler,
                           gen_adv, gen_crit, disc_adv,
                           conf.get_attr('generator_loss_weights', {}),
                           conf.get_attr('discriminator_loss_weights', {}), cuda,
                           train_gen_metric_fns, train_disc_metric_fns, val_metric_fns, {},
                           gen_updates_per_step=gen_opt_conf.get_attr('updates_per_step', 1),
                           disc_updates_per_step=disc_opt_conf.get_attr('updates_per_step', 1),
                           disc_input_fn=disc_input_fn, val_disc_input_fn=val_disc_input_fn,
                           pretrain_generator_epochs=conf.get_attr('pretrain_generator_epochs'),
                           pretrain_discriminator_epochs=conf.get_attr('pretrain_discriminator_epochs'))


def _split_disc_output(out, b):
  """Halves of a discriminator output computed on a stacked [first; second] batch."""
  first, second = {}, {}
  if torch.is_tensor(out.get('logits')) and out['logits'].shape[0] == 2 * b:
    first['_pair_logits'] = second['_pair_logits'] = out['logits']
    first['_pair_half'], second['_pair_half'] = 0, 1       # (GANLoss assigns its labels by these, not by position)
  for k, v in out.items():
    if torch.is_tensor(v):
      first[k], second[k] = v[:b], v[b:]
    elif isinstance(v, list) and v and torch.is_tensor(v[0]):
      first[k], second[k] = [t[:b] for t in v], [t[b:] for t in v]
    else:
      first[k] = second[k] = v
  return first, second


# hipGraph executables this process no longer replays are PARKED here, not destroyed.  Destroying a captured graph and
# then instantiating and launching a new one whose buffers have the same addresses (the caching allocator hands a second
# runner of the same shapes the blocks the first one freed) crashes inside hipGraphLaunch of this runtime -- a stale
# entry of the executable's stream list, libamdhip64 +0xaee41, fault address 0x1d8; reproduced 28 times in 45 runs with
# runners c3 -> c5 -> c5 in one process, 0 in 45 with the first c5 graphs kept alive (tools/segv_hunt.sh, DESIGN.md
# section 4).  A training process captures once (the learning rate is a device scalar); only benchmarks and tests build
# several runners, and they are short-lived.  CSMRI_DESTROY_GRAPHS=1 restores the destruction (the reproduction).
GRAPH_GRAVEYARD = []


def retire_graphs(g):
  import os
  if os.environ.get('CSMRI_DESTROY_GRAPHS') != '1':
    GRAPH_GRAVEYARD.append([g.get('graphs'), g.get('pf_graph'), g.get('graph'), g.get('graph_adam')])


def _epoch_window(spec):
  """Half-open epoch window [first, end) of a pretraining phase from its config value: a count n means the first n
  epochs (1-based), a pair is a window as given, None an empty window (semantics pinned by fixture F11 /
  oracle.pretraining_flags; reference config keys pretrain_generator / pretrain_discriminator)."""
  if spec is None:
    return (-1, -1)
  if isinstance(spec, int):
    return (1, 1 + spec)
  first, end = spec
  if not first < end:
    raise AssertionError('Starting epoch must be smaller than ending epoch')
  return (first, end)


class AdversarialRunner(BaseRunner):
  def __init__(self, gen_model, disc_model=None, gen_optimizer=None, disc_optimizer=None,
               gen_lr_scheduler=None, disc_lr_scheduler=None, gen_adv_criteria=None,
               gen_criteria=None, disc_adv_criteria=None, gen_loss_weights=None,
               disc_loss_weights=None, cuda='', train_gen_metric_fns=None,
               train_disc_metric_fns=None, val_metric_fns=None, val_disc_metric_fns=None,
               output_transform=None, train_input_batch_transform=None,
               test_input_batch_transform=None, gen_updates_per_step=1, disc_updates_per_step=1,
               disc_input_fn=None, val_disc_input_fn=None, pretrain_generator_epochs=None,
               pretrain_discriminator_epochs=None):
    super(AdversarialRunner, self).__init__(cuda)
    self.gen, self.disc = gen_model, disc_model
    self.gen_optimizer, self.disc_optimizer = gen_optimizer, disc_optimizer
    self.gen_lr_scheduler, self.disc_lr_scheduler = gen_lr_scheduler, disc_lr_scheduler
    # a step of one network's optimizer leaves the packed weights of the other current
    from models.utils import trainable_pack_groups
    if gen_optimizer is not None and hasattr(gen_optimizer, 'pack_groups'):
      gen_optimizer.pack_groups = lambda: trainable_pack_groups(self.gen)
    if disc_optimizer is not None and hasattr(disc_optimizer, 'pack_groups'):
      disc_optimizer.pack_groups = lambda: trainable_pack_groups(self.disc)
    # every discriminator gradient is written by library kernels (conv / BatchNorm backward), so the fill of its
    # 112 MB flat gradient buffer per step is replaced by "the first write overwrites" (FlatAdam.lazy_zero; identical
    # losses after 250 steps, 6.13 -> 6.04 ms: the fill sat in front of D's backward on the main chain).  The
    # generator likewise (the wrapper's scale, an autograd-accumulated gradient, is still zeroed).
    for opt in (disc_optimizer, gen_optimizer):
      if opt is not None and hasattr(opt, 'lazy_zero'):
        opt.lazy_zero = True
    self._grad_hook = None
    if dist_utils.exchange_active() and gen_optimizer is not None and disc_optimizer is not None:
      # data parallelism: a sub-bucket of gradients leaves as soon as the backward has issued its last layer.  The
      # hook is installed around this runner's own steps only (_hooked), never process-wide
      def _ready(layer, opts=(disc_optimizer, gen_optimizer)):
        for o in opts:
          o.grad_ready(layer)
      self._grad_hook = _ready
    self.train_gen_metric_fns = train_gen_metric_fns or {}
    self.train_disc_metric_fns = train_disc_metric_fns or {}
    self.val_metric_fns = val_metric_fns or {}
    self.val_disc_metric_fns = val_disc_metric_fns or {}
    self.train_model_input_fn = self._get_model_input_fn(self.gen, train_input_batch_transform)
    self.test_model_input_fn = self._get_model_input_fn(self.gen, test_input_batch_transform)
    self.gen_updates_per_step, self.disc_updates_per_step = gen_updates_per_step, disc_updates_per_step
    # reference adversarial_runner.py:176-180
    if gen_updates_per_step == 1 and disc_updates_per_step == 1:
      self._train_step = self._train_single_step
    else:
      self._train_step = self._train_multiple_steps
    self.disc_input_fn, self.val_disc_input_fn = disc_input_fn, val_disc_input_fn
    self.gen_adv_criteria = OrderedDict(gen_adv_criteria or {})
    self.gen_criteria = OrderedDict(gen_criteria or {})
    self.disc_adv_criteria = OrderedDict(disc_adv_criteria or {})
    self.gen_loss_weights = self._get_loss_weights(gen_loss_weights or {}, self.gen_adv_criteria,
                                                   self.gen_criteria)
    self.disc_loss_weights = self._get_loss_weights(disc_loss_weights or {},
                                                    self.disc_adv_criteria)
    self.discriminator_enabled = True
    self.generator_enabled = True
    self._host_weights = {}

    self.generator_pretraining_schedule = _epoch_window(pretrain_generator_epochs)
    self.discriminator_pretraining_schedule = _epoch_window(pretrain_discriminator_epochs)
    self.pool_decisions = None            # optional injected image-pool decisions (tests)
    self._graph = None
    self._last_metrics = None
    self.overlap_streams = False
    self._side_stream = None
    # software pipelining of the FROZEN pretrained reconstruction: its forward for the next batch
    # runs on a side stream while this step trains (exact reordering; bench.py turns it on)
    self.prefetch_pretrained = False
    self._pf = None
    self._pf_stream = None
    self.third_pass_early = True          # with vgg_early (single GPU)
    self._side_stream3 = None
    self._metric_stream = None
    self.vgg_early = None                 # None: decided at the first step (True on a single GPU)
    # graph mode: launch the look-ahead graph before (True) or after (False) the step's own graph(s)
    self.lookahead_first = True
    self.batch_disc_passes = True         # D(fake) and D(real) of the D phase as one grouped pass
    self.batch_three_disc_passes = True   # ... together with the generator phase's D(fake): one pass of three groups

  # -- reference surface -------------------------------------------------------
  def get_named_outputs(self, data):
    batch, out_gen = data[0], data[1]
    pred = out_gen['pred'] if isinstance(out_gen, dict) else out_gen
    return {'input': batch['inp'], 'prediction': pred, 'target': batch['target'],
            'disc_fake': data[2]}

  def get_named_models(self):
    return {'generator': self.gen, 'discriminator': self.disc}

  def state_dict(self):
    return {'generator': self.gen.state_dict(), 'discriminator': self.disc.state_dict(),
            'gen_optimizer': self.gen_optimizer.state_dict(),
            'disc_optimizer': self.disc_optimizer.state_dict()}

  def load_state_dict(self, state_dict):
    from csmri_hip import ops
    self.gen.load_state_dict(state_dict['generator'])
    if self.disc is not None:
      assert 'discriminator' in state_dict, 'Incompatible checkpoint'
      self.disc.load_state_dict(state_dict['discriminator'])
    ops.bump_weight_epoch()
    from models.utils import refresh_packs
    for net in (self.gen, self.disc):
      if net is not None:
        refresh_packs(net)
    if self.gen_optimizer is not None:
      assert 'gen_optimizer' in state_dict, 'Incompatible checkpoint'
      self.gen_optimizer.load_state_dict(state_dict['gen_optimizer'])
    if self.disc_optimizer is not None:
      assert 'disc_optimizer' in state_dict, 'Incompatible checkpoint'
      self.disc_optimizer.load_state_dict(state_dict['disc_optimizer'])

  def __str__(self):
    s = 'Generator:\n' + str(self.gen)
    if self.disc is not None:
      s += '\nDiscriminator:\n' + str(self.disc)
    return s

  def predict(self, batch):
    return self.gen(*self.train_model_input_fn(batch, use_batch_transform=False))

  def _weighted_total(self, losses, weights):
    if losses[0].is_cuda and len(losses) <= 16:
      key = id(weights)
      host = self._host_weights.get(key)
      if host is None:                     # (loss weights are configuration constants: read back once)
        host = self._host_weights[key] = [float(w) for w in weights.detach().cpu()]
      return ops.weighted_sum(losses, host)
    return torch.sum(torch.stack(losses) * weights)

  def _update_step(self, optimizer, losses, weights):
    """zero_grad; backward; all-reduce; Adam (reference :314-320)."""
    total = self._weighted_total(losses, weights)
    optimizer.zero_grad()
    ops.GRAD_READY_HOOK = self._grad_hook
    try:
      ops.backward_scalar(total)
      ops.join_wgrad_stream()
    finally:
      ops.GRAD_READY_HOOK = None
    optimizer.start_allreduce()
    optimizer.step()
    return total.detach()

  # ---- the training step, in four segments separated by the two collectives --------
  # S1: G fwd, D(fake.detach via pool), D(real), D losses, D backward
  #     -> all-reduce of D's gradient bucket starts (async, RCCL stream)
  # S2: third D forward (with gradient to G), G losses (GAN, FM, VGG, FeaturePenalty)
  #     -> wait for D's bucket
  # S3: D Adam step, then G backward THROUGH the updated D (faithful ordering A)
  #     -> all-reduce of G's bucket
  # S4: G Adam step, training metrics, all scalars stacked into one device vector
  # Eager mode runs the segments back to back; graph mode (enable_graphs) captures each
  # segment into a hipGraph once and replays them, the collectives staying eager between.

  def _seg1(self, st):
    batch = st['batch']
    if self.vgg_early is None:
      self.vgg_early = not dist_utils.exchange_active()
    gen_inp = self.train_model_input_fn(batch)
    st['gen_inp0'] = gen_inp[0]
    if st.get('pre_cur') is not None:
      st['pre_cur'].record_stream(torch.cuda.current_stream())   # produced on the prefetch stream
      out_gen = self.gen.forward_with_pre(*gen_inp, pre=st['pre_cur'])
    else:
      out_gen = self.gen(*gen_inp)
    st['out_gen'] = out_gen
    st['side_results'] = {}
    # ISSUE ORDER (a hipGraph replays its nodes in capture order at the host's enqueue rate, and only a node's
    # first-issued successor stays on its hardware queue; measured alternatives: DESIGN 9.0): the VGG branch is forked
    # right behind the generator (deferring it on an event costs 3-7 %), the third discriminator pass right behind the
    # two D-phase passes, the look-ahead at the END of this segment (in front of the U-Net forward its ~35 nodes kept
    # the critical chain waiting 0.3 ms at every step start).
    if self.overlap_streams and self.vgg_early:
      self._fork_vgg(st, out_gen, batch)
    pair = None
    tgt = batch['target']
    st['out_disc_fake_early'] = None
    can_stack = tgt.is_cuda and tgt.dim() == 4 and tgt.shape[1] == 2 and hasattr(self.disc_input_fn, 'out_dtype')
    if self.batch_disc_passes and self.batch_three_disc_passes and can_stack:
      # ALL THREE discriminator forwards of the step (reference :332, :338, :354 -- same weights, no dependence on
      # each other) as ONE pass over [pool-fake; real; current-fake]: per-group BatchNorm statistics, running-statistics
      # updates and dropout draws in the reference's order.  The discriminator loss then differentiates groups 0-1
      # w.r.t. the weights, the generator losses group 2 w.r.t. its input (CNNDiscriminator.forward_grouped); |pred|
      # is computed once and feeds both the history pool and group 2.
      b, _, h, w = tgt.shape
      x_all = torch.empty(3 * b, h, w, 8, dtype=self.disc_input_fn.out_dtype(), device=tgt.device)
      fn = self.disc_input_fn
      mag = fn.magnitude(out_gen, x_all[2 * b:])
      fn(out_gen, gen_inp[0], out_gen, is_real_input=False, detach=True, pool_decisions=self.pool_decisions,
         out=x_all[:b], mag=mag)
      fn(tgt, gen_inp[0], out_gen, is_real_input=True, detach=True, out=x_all[b:2 * b])
      live_stream = None
      if self.overlap_streams and self.vgg_early and self.third_pass_early:
        # the generator-loss graph of group 2 is BUILT on its own stream (its forward launches only the tiny logit
        # conversions), so autograd runs its backward there, next to the VGG branch's (as the separate third pass did)
        if self._side_stream3 is None:
          self._side_stream3 = ops.named_stream('third')
        live_stream = self._side_stream3
      out_pair, out_cur = self.disc.forward_grouped(
          x_all, 3, [(0, 2, x_all[:2 * b], True, None),
                     (2, 3, lambda: fn.link(out_gen, x_all[2 * b:]), False, live_stream)])
      out_fake_d, out_real = _split_disc_output(out_pair, b)
      st['out_disc_fake_early'] = out_cur
      st['_live_stream'] = live_stream
    else:
      if self.batch_disc_passes and can_stack:
        # [fake; real] of the batched discriminator pass: the two inputs are written straight into the halves of
        # one tensor (no torch.cat of their results)
        b, _, h, w = tgt.shape
        pair = torch.empty(2 * b, h, w, 8, dtype=self.disc_input_fn.out_dtype(), device=tgt.device)
      in_fake = self.disc_input_fn(out_gen, gen_inp[0], out_gen, is_real_input=False, detach=True,
                                   pool_decisions=self.pool_decisions, **({} if pair is None else {'out': pair[:b]}))
      in_real = self.disc_input_fn(tgt, gen_inp[0], out_gen, is_real_input=True, detach=True,
                                   **({} if pair is None else {'out': pair[b:]}))
      if self.batch_disc_passes:
        # the two passes of reference :333-341 as ONE pass over [fake; real] with per-half BatchNorm
        # statistics and dropout draws (identical results, half the launches on D's small maps)
        out_fake_d, out_real = _split_disc_output(
            self.disc(nhwc=pair if pair is not None else torch.cat([in_fake, in_real], 0), groups=2), in_fake.shape[0])
      else:
        out_fake_d = self.disc(nhwc=in_fake)
        out_real = self.disc(nhwc=in_real)
      if self.overlap_streams and self.vgg_early and self.third_pass_early:
        # single GPU: the third D forward (reference :354-357; same D weights, it only has to
        # follow the two passes above for the BatchNorm running statistics) runs on its own stream
        # next to the D loss and backward below -- two chains of small kernels share the chip
        if self._side_stream3 is None:
          self._side_stream3 = ops.named_stream('third')
        self._side_stream3.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._side_stream3):
          st['out_disc_fake_early'] = self._third_disc_pass(st)
        st['_live_stream'] = self._side_stream3
    st['out_disc_real'] = out_real
    names, vals, disc_losses = [], [], []
    for name, criterion in self.disc_adv_criteria.items():
      loss = criterion(out_fake_d, out_real)
      disc_losses.append(loss)
      names.append('disc_loss_' + name)
      vals.append(loss.detach())
    total_disc = self._weighted_total(disc_losses, self.disc_loss_weights)
    self.disc_optimizer.zero_grad()
    ops.enable_wgrad_stream(self.overlap_streams)
    ops.backward_scalar(total_disc)
    # (Joining D's weight-gradient stream only at D's Adam -- so that the look-ahead fork and the generator's losses
    # run next to the weight-gradient tail of D's backward -- measured 9 % slower on the bench and crashed the graph
    # replay of a small configuration inside the runtime: removed)
    ops.join_wgrad_stream()
    names.append('disc_loss')
    vals.append(total_disc.detach())
    st['names'], st['vals'] = names, vals
    if st['side_results']:                       # a segment (graph) ends with every stream joined
      torch.cuda.current_stream().wait_stream(self._side_stream)
    if st.get('_live_stream') is not None:
      torch.cuda.current_stream().wait_stream(st['_live_stream'])
    self._fork_prefetch(st)
    self._join_prefetch(st, 1)

  def _fork_prefetch(self, st):
    if st.get('batch_next') is None:
      return
    if self._pf_stream is None:
      self._pf_stream = ops.named_stream('lookahead')
    self._pf_stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(self._pf_stream):
      st['pre_next'] = self.gen.precompute(*self.train_model_input_fn(st['batch_next']))
    st['_pf_pending'] = True

  def _join_prefetch(self, st, seg):
    """Join the look-ahead stream.  A captured segment must end with every stream joined: with one
    graph per step (single GPU) the frozen forward of batch t+1 may run until the end of step t,
    with one graph per segment it is joined inside segment 1."""
    if not st.get('_pf_pending'):
      return
    last = 1 if dist_utils.exchange_active() else 4
    if seg >= last:
      torch.cuda.current_stream().wait_stream(self._pf_stream)
      st['_pf_pending'] = False

  def _third_disc_pass(self, st):
    self.disc.set_wgrad(False)     # D's weight gradients of this pass are discarded (A-5)
    out_fake = self.disc(nhwc=self.disc_input_fn(st['out_gen'], st['gen_inp0'], st['out_gen'],
                                                 is_real_input=False, detach=False))
    self.disc.set_wgrad(True)
    return out_fake

  def _fork_vgg(self, st, out_gen, batch):
    """The VGG perceptual branch (big GEMMs) only needs the generator output: run it on a side
    stream next to chains of small kernels -- the discriminator passes and backward of segment 1
    on a single GPU (vgg_early), or the third D forward of segment 2 when segment 2 has to hide
    the D-bucket all-reduce.  Autograd replays the stream assignment in the backward, so the two
    gradient chains into `pred` overlap as well.  (Cutting the branch's backward out of the generator backward and
    issuing it earlier measured neutral to 6 % slower, DESIGN 9.0: removed.)"""
    if self._side_stream is None:
      self._side_stream = ops.named_stream('vgg')
    self._side_stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(self._side_stream):
      for name, criterion in self.gen_criteria.items():
        if name == 'VGG19':
          st['side_results'][name] = criterion(out_gen, batch)

  def _seg2(self, st):
    batch, out_gen = st['batch'], st['out_gen']
    forked_here = False
    if self.overlap_streams and not st['side_results']:
      self._fork_vgg(st, out_gen, batch)
      forked_here = True
    out_fake = st.get('out_disc_fake_early')
    if out_fake is None:
      out_fake = self._third_disc_pass(st)
    st['out_disc_fake'] = out_fake
    gen_losses = []
    for name, criterion in self.gen_adv_criteria.items():
      loss = criterion(out_fake, st['out_disc_real'])
      gen_losses.append(loss)
      st['names'].append('gen_loss_' + name)
      st['vals'].append(loss.detach())
    side = st.get('side_results', {})
    for name, criterion in self.gen_criteria.items():
      loss = side[name] if name in side else criterion(out_gen, batch)
      gen_losses.append(loss)
      st['names'].append('gen_loss_' + name)
      st['vals'].append(loss.detach())
    if side and forked_here:
      torch.cuda.current_stream().wait_stream(self._side_stream)
    st['total_gen'] = self._weighted_total(gen_losses, self.gen_loss_weights)
    self._join_prefetch(st, 2)

  def _repack_disc(self):
    from models.utils import trainable_pack_groups
    for g in trainable_pack_groups(self.disc) or []:
      g.repack_stale()

  def _fork_train_metrics(self, st, event=None):
    """The training metrics only read forward results: compute them on their own stream next to
    the generator backward instead of as a serial tail of ~20 tiny launches after the G step.
    ``event``: what the metrics depend on was complete at that event (issue the launches later without
    making them wait for what was issued in between)."""
    data = (st['batch'], st['out_gen'], st['out_disc_fake'], st['out_disc_real'])
    if not self.overlap_streams:
      with torch.no_grad():
        st['train_metrics'] = self._compute_train_metrics(data)
      return
    if self._metric_stream is None:
      self._metric_stream = ops.named_stream('metrics')
    if event is not None:
      self._metric_stream.wait_event(event)
    else:
      self._metric_stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(self._metric_stream), torch.no_grad():
      st['train_metrics'] = self._compute_train_metrics(data)
    st['_metrics_pending'] = True

  def _seg3(self, st):
    # the training metrics only read forward results that are complete HERE (event); their ~20 launches are issued
    # after the generator backward so that they do not sit in front of it in the replay's enqueue order (+0.5-1 %)
    ev_m = None
    if self.overlap_streams:
      ev_m = torch.cuda.Event()
      ev_m.record(torch.cuda.current_stream())
    else:
      self._fork_train_metrics(st)
    live = st.get('_live_stream')
    if live is not None:
      # the generator-phase D pass lives on its own stream: run D's Adam there too.  Only that pass's
      # backward (data gradients through the UPDATED D, ordering A) needs the new weights; the
      # VGG backward and the rest of the generator backward start without waiting for it.
      live.wait_stream(torch.cuda.current_stream())
      with torch.cuda.stream(live):
        self.disc_optimizer.apply()
    else:
      self.disc_optimizer.apply()
    self.gen_optimizer.zero_grad()
    ops.enable_wgrad_stream(self.overlap_streams)
    ops.backward_scalar(st['total_gen'])
    if ev_m is not None:
      self._fork_train_metrics(st, ev_m)
    ops.join_wgrad_stream()
    st['names'].append('gen_loss')
    st['vals'].append(st['total_gen'].detach())
    if live is not None:
      # D's forward-mode packs for the NEXT step, behind the third pass's backward on its stream
      # (next to the rest of the generator backward) instead of in front of the next D forward
      with torch.cuda.stream(live):
        self._repack_disc()
      torch.cuda.current_stream().wait_stream(live)
    if st.pop('_metrics_pending', False):
      torch.cuda.current_stream().wait_stream(self._metric_stream)
    self._join_prefetch(st, 3)

  def _seg4(self, st):
    self.gen_optimizer.apply()
    self._join_prefetch(st, 4)
    metrics = st.pop('train_metrics')
    st['metric_names'] = list(metrics.keys())
    vec = [v.float().reshape(()) for v in st['vals']] + \
          [m.sum_values.float().reshape(()) if torch.is_tensor(m.sum_values)
           else torch.tensor(float(m.sum_values), device=self.device) for m in metrics.values()]
    st['vec'] = torch.stack(vec)

  def _run_segments_eager(self, st):
    ops.GRAD_READY_HOOK = self._grad_hook        # (sub-buckets start from inside the backward passes)
    try:
      self._seg1(st)
      self.disc_optimizer.start_allreduce()
      self._seg2(st)
      self.disc_optimizer.wait_allreduce()
      self._seg3(st)
      self.gen_optimizer.start_allreduce()
      self.gen_optimizer.wait_allreduce()
      self._seg4(st)
    finally:
      ops.GRAD_READY_HOOK = None

  def enable_graphs(self, example_batch, warmup=3):
    """Capture the four segments as hipGraphs (shared memory pool) for this batch shape.
    ``example_batch``: device batch dict; its values seed the static input buffers."""
    assert self.pool_decisions is None and not self.disc.injected_dropout, \
        'graph mode draws its own randomness'
    static = {k: v.detach().clone() for k, v in example_batch.items()}
    static_next = static_pre = None
    if self.prefetch_pretrained:
      static_next = {k: v.detach().clone() for k, v in example_batch.items()}
      static_pre = self.gen.precompute(*self.train_model_input_fn(static)).clone()
    self._set_train()
    side = ops.named_stream('warmup')
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
      for _ in range(warmup):
        self._run_segments_eager({'batch': static, 'batch_next': static_next, 'pre_cur': static_pre})
    torch.cuda.current_stream().wait_stream(side)
    from models.utils import prepare_packs_for_capture
    for net in (self.gen, self.disc):
      prepare_packs_for_capture(net)
    torch.cuda.synchronize()
    pool = getattr(self.disc_input_fn, 'image_pool', None)
    bns = [m for net in (self.gen, self.disc) for m in net.modules() if hasattr(m, 'batches_tracked')]
    before = [m.batches_tracked for m in bns]
    if pool is not None:
      # the capture pass executes nothing: its plan must not mark slots as filled nor consume
      # python-random draws (the first replay plans again from this very state)
      import random as _random
      pool_count, rnd_state = pool.count, _random.getstate()
      pool.prepare(torch.empty((static['inp'].shape[0],) + tuple(pool.buffer.shape[1:]),
                               dtype=pool.buffer.dtype, device=self.device))
      pool.count = pool_count
      _random.setstate(rnd_state)
      pool.external_plan = True
    # The look-ahead (frozen reconstruction of batch t + 1) is captured as a graph OF ITS OWN, replayed on its own stream
    # next to the step's graph: inside one graph a forked branch is a second successor and is serialised against the
    # main chain by this runtime (it ran alone for ~0.22 ms of every step wherever it was forked), two
    # independent graph launches on two streams simply share the chip.  Same arithmetic, same inputs: an exact reordering.
    st = {'batch': static, 'batch_next': None, 'pre_cur': static_pre}
    graphs = []
    # a cyclic-GC pass in the middle of a capture may destroy graphs/events of an earlier
    # runner, which the runtime rejects while a stream is capturing: collect now, pause GC
    import gc
    gc.collect()
    gc_was_enabled = gc.isenabled()
    gc.disable()
    from training import distributed as dist_utils
    if not dist_utils.exchange_active():
      # no collectives between the segments on a single GPU: one graph, three launch gaps fewer
      def whole(st_):
        for seg in (self._seg1, self._seg2, self._seg3, self._seg4):
          seg(st_)
      segments = (whole,)
    else:
      segments = (self._seg1, self._seg2, self._seg3, self._seg4)
    cap_stream = ops.named_stream('capture')
    self.disc_optimizer.sync_lr()
    self.gen_optimizer.sync_lr()
    ops.GRAD_READY_HOOK = self._grad_hook        # same launch plan as the eager steps (the hook itself is a no-op while capturing)
    try:
      for seg in segments:
        g = torch.cuda.CUDAGraph()
        # thread_local: RCCL's watchdog thread may touch the runtime while this thread captures
        with torch.cuda.graph(g, pool=graphs[0].pool() if graphs else None, stream=cap_stream,
                              capture_error_mode='thread_local'):
          seg(st)
        graphs.append(g)
    except Exception:
      if pool is not None:
        pool.external_plan = False
      raise
    finally:
      ops.GRAD_READY_HOOK = None
      if gc_was_enabled:
        gc.enable()
    pf_graph = pre_next = None
    if self.prefetch_pretrained:
      if self._pf_stream is None:
        self._pf_stream = ops.named_stream('lookahead')
      gc.collect()
      gc.disable()
      try:
        pf_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(pf_graph, stream=self._pf_stream, capture_error_mode='thread_local'):
          pre_next = self.gen.precompute(*self.train_model_input_fn(static_next))
      finally:
        if gc_was_enabled:
          gc.enable()
      st['pre_next'] = pre_next
    self._graph = {'graphs': graphs, 'static': static, 'static_next': static_next, 'static_pre': static_pre,
                   'st': st, 'pool': pool, 'pf_graph': pf_graph, 'pf_done': None,
                   'bn_delta': [(m, m.batches_tracked - b) for m, b in zip(bns, before)]}
    # the capture pass did not execute anything: optimizer host mirrors advanced, undo
    self.disc_optimizer.step_count -= 1
    self.gen_optimizer.step_count -= 1
    for m, d in self._graph['bn_delta']:
      m.batches_tracked -= d
    return self

  def disable_graphs(self):
    g = getattr(self, '_graph', None)
    if g is not None and g['pool'] is not None:
      g['pool'].external_plan = False
    if g is not None:
      retire_graphs(g)
    self._graph = None

  def __del__(self):
    try:
      g = getattr(self, '_graph', None)
      if g is not None:
        retire_graphs(g)
    except Exception:       # interpreter shutdown: module globals are already gone, and so is the need
      pass

  def _run_segments_graphed(self, batch, batch_next=None, pre_cur=None):
    from csmri_hip import ops
    G = self._graph
    if G.get('pf_done') is not None:                             # pre_cur is the look-ahead graph's output buffer
      torch.cuda.current_stream().wait_event(G['pf_done'])
    dst = list(G['static'].values())
    src = [batch[k] for k in G['static']]
    if G['static_next'] is not None:
      nb = batch_next if batch_next is not None else batch       # last step of an epoch: nothing to prefetch
      dst += list(G['static_next'].values()) + [G['static_pre']]
      src += [nb[k] for k in G['static_next']] + [pre_cur]
    if all(t.is_cuda for t in src):
      torch._foreach_copy_(dst, src)                             # one multi-tensor launch
    else:
      for d, t in zip(dst, src):
        d.copy_(t, non_blocking=True)
    if G['pool'] is not None:
      G['pool'].external_plan = False
      G['pool'].prepare(G['pool'].buffer[:G['static']['inp'].shape[0]])
      G['pool'].external_plan = True
    ready = None
    if G.get('pf_graph') is not None:
      ready = torch.cuda.Event()           # the copies of the look-ahead's inputs are complete here
      ready.record()

    def launch_lookahead():
      # the look-ahead graph on its own stream, behind the copies of its inputs; its result is read by the NEXT step's
      # copy into static_pre (which waits for pf_done)
      if ready is None:
        return
      self._pf_stream.wait_event(ready)
      with torch.cuda.stream(self._pf_stream):
        G['pf_graph'].replay()
        done = torch.cuda.Event()
        done.record(self._pf_stream)
      G['pf_done'] = done
    if self.lookahead_first:
      launch_lookahead()
    self.disc_optimizer.sync_lr()                  # (a scheduler may have moved the rates: device scalars, not graph nodes)
    self.gen_optimizer.sync_lr()
    if len(G['graphs']) == 1:
      G['graphs'][0].replay()
      if not self.lookahead_first:
        launch_lookahead()
    else:
      g1, g2, g3, g4 = G['graphs']
      g1.replay()
      self.disc_optimizer.start_allreduce()
      g2.replay()
      self.disc_optimizer.wait_allreduce()
      g3.replay()
      self.gen_optimizer.start_allreduce()
      self.gen_optimizer.wait_allreduce()
      g4.replay()
      if not self.lookahead_first:
        launch_lookahead()
    self.disc_optimizer.step_count += 1
    self.gen_optimizer.step_count += 1
    for m, d in G['bn_delta']:
      m.batches_tracked += d
    ops.bump_weight_epoch()
    return G['st']

  def train_epoch(self, loader, epoch, *args, **kwargs):
    self._pf = None                      # the look-ahead never crosses an epoch boundary
    return super(AdversarialRunner, self).train_epoch(loader, epoch, *args, **kwargs)

  def _train_single_step(self, loader):
    if not (self.discriminator_enabled and self.generator_enabled):
      self._pf = None
      return self._train_single_step_general(loader)
    batch_next = pre_cur = None
    if self.prefetch_pretrained:
      # this step's batch was fetched (and its pretrained reconstruction issued) one step ago
      if self._pf is None:
        batch = self._request_data(loader)
        if batch is None:
          return 0, None, None
        pre_cur = self.gen.precompute(*self.train_model_input_fn(batch))
      else:
        batch, pre_cur = self._pf
        if batch is None:
          return 0, None, None
      batch_next = self._request_data(loader) if self.data_iter is not None else None
    else:
      batch = self._request_data(loader)
      if batch is None:
        return 0, None, None
    if getattr(self, '_graph', None) is not None:
      st = self._run_segments_graphed(batch, batch_next, pre_cur)
      vec = st['vec'].clone()
    else:
      st = {'batch': batch, 'batch_next': batch_next, 'pre_cur': pre_cur}
      self._run_segments_eager(st)
      vec = st['vec']
    if self.prefetch_pretrained:
      self._pf = (batch_next, st['pre_next'] if batch_next is not None else None)
    n = len(st['names'])
    shared = SharedVec(vec)          # epoch accumulation adds the whole vector once per step
    loss_metrics = {name: get_loss_metric(VecRef(shared, i)) for i, name in enumerate(st['names'])}
    self._last_metrics = {name: MaxMetric(VecRef(shared, n + j)) for j, name in enumerate(st['metric_names'])}
    return 1, loss_metrics, (st['batch'], st['out_gen'], st['out_disc_fake'], st['out_disc_real'])

  # -- epoch hooks: learning-rate schedulers and the pretraining windows (behaviour pinned by fixture F11:
  #    tests/test_hip_path.py::test_f11_*; the reference's hooks are adversarial_runner.py:267-305) --
  def _learning_rates(self):
    return [o.param_groups[0]['lr'] for o in (self.gen_optimizer, self.disc_optimizer) if o is not None]

  def _step_schedulers(self, due):
    for scheduler in (self.gen_lr_scheduler, self.disc_lr_scheduler):
      if due(scheduler):
        scheduler.step()

  def _networks_enabled(self, epoch):
    """(discriminator_enabled, generator_enabled) in ``epoch``: inside the discriminator's pretraining window only the
    discriminator trains; else inside the generator's only the generator; else both.  (Where the two windows overlap
    the discriminator's wins: oracle.pretraining_flags, F11.)"""
    in_window = lambda w: w[0] <= epoch < w[1]
    if in_window(self.discriminator_pretraining_schedule):
      return True, False
    if in_window(self.generator_pretraining_schedule):
      return False, True
    return True, True

  def epoch_beginning(self, epoch):
    lrs = self._learning_rates()
    self._step_schedulers(is_pre_epoch_scheduler)
    self.discriminator_enabled, self.generator_enabled = self._networks_enabled(epoch)
    for trained, frozen, (first, end) in (('generator', 'discriminator', self.generator_pretraining_schedule),
                                          ('discriminator', 'generator', self.discriminator_pretraining_schedule)):
      if epoch == first:
        logging.info('Epoch %d: %s pretraining begins (%s frozen until epoch %d)', epoch, trained, frozen, end)
      elif epoch == end:
        logging.info('Epoch %d: %s pretraining is over', epoch, trained)
      elif first < epoch < end:
        logging.debug('Epoch %d: %s pretraining, %s frozen', epoch, trained, frozen)
    self._after_lr_change(lrs)

  def epoch_finished(self, epoch):
    lrs = self._learning_rates()
    self._step_schedulers(is_post_epoch_scheduler)
    self._after_lr_change(lrs)

  def _after_lr_change(self, old_lrs):
    """Nothing to re-capture: the fused Adam kernel reads its learning rate from device memory (csmri_adam_dev_lr,
    FlatAdam.lr_dev), which FlatAdam.sync_lr() refreshes in front of the next eager step or graph replay.  (Until round 5
    the rate was a launch argument and a scheduler step forced a new capture -- and dropping a captured graph to capture
    its successor is what the HIP runtime does not survive reliably)"""
    return None

  # -- steps outside the fused one-update-each case: a network disabled by a pretraining window, or several updates
  #    per step.  Semantics pinned by fixture F11 against oracle.gan_train_step / oracle.gan_train_multi_step
  #    (tests/test_hip_path.py::test_f11_*); the reference's versions are adversarial_runner.py:322-525.  Built from
  #    three pieces: a PHASE evaluates one network's named losses on a generator output, `_apply_phase` turns a phase
  #    into an optimizer update and records its values, and a step is a short schedule of phases. ------------------
  def _disc_forward_pair(self, out_gen, gen_inp0, target):
    out_fake = self.disc(nhwc=self.disc_input_fn(out_gen, gen_inp0, out_gen, is_real_input=False, detach=True,
                                                 pool_decisions=self.pool_decisions))
    out_real = self.disc(nhwc=self.disc_input_fn(target, gen_inp0, out_gen, is_real_input=True, detach=True))
    return out_fake, out_real

  def _general_update(self, optimizer, losses, weights):
    assert len(losses) == int(weights.numel()), \
        'got %d losses for %d loss weights (the reference fails the same way: adversarial generator ' \
        'losses are weighted but not computed while the discriminator is disabled)' % (len(losses), weights.numel())
    return self._update_step(optimizer, losses, weights)

  @staticmethod
  def _evaluate(prefix, criteria, *operands):
    return [(prefix + name, criterion(*operands)) for name, criterion in criteria.items()]

  def _disc_phase(self, batch, gen_inp0, out_gen):
    """D(pool(fake).detach()), D(real) -> the discriminator's named losses."""
    fake, real = self._disc_forward_pair(out_gen, gen_inp0, batch['target'])
    return self._evaluate('disc_loss_', self.disc_adv_criteria, fake, real), fake, real

  def _gen_phase(self, batch, gen_inp0, out_gen, real=None, fresh_real=False):
    """The generator's named losses: adversarial ones through D(fake) with gradient (only while the discriminator
    takes part; ``real`` = a D(real) output to match features against, recomputed when ``fresh_real``), then the
    ordinary criteria on (out_gen, batch)."""
    named, fake = [], None
    if self.discriminator_enabled:
      fake = self.disc(nhwc=self.disc_input_fn(out_gen, gen_inp0, out_gen, is_real_input=False, detach=False))
      if fresh_real:
        real = self.disc(nhwc=self.disc_input_fn(batch['target'], gen_inp0, out_gen, is_real_input=True, detach=True))
      named += self._evaluate('gen_loss_', self.gen_adv_criteria, fake, real)
    else:
      real = None
    return named + self._evaluate('gen_loss_', self.gen_criteria, out_gen, batch), fake, real

  def _apply_phase(self, who, named, record):
    optimizer, weights = ((self.disc_optimizer, self.disc_loss_weights) if who == 'disc' else
                          (self.gen_optimizer, self.gen_loss_weights))
    for name, loss in named:
      record(name, get_loss_metric(loss.detach()))
    record(who + '_loss', get_loss_metric(self._general_update(optimizer, [loss for _, loss in named], weights)))

  def _train_single_step_general(self, loader):
    """One batch, ONE generator forward shared by both phases; every loss is evaluated before either network moves,
    then the discriminator is updated, then the generator (its backward runs through the updated discriminator:
    ordering A, SURVEY A-4).  The D(real) pass of the discriminator phase also serves feature matching."""
    batch = self._request_data(loader)
    if batch is None:
      return 0, None, None
    gen_inp = self.train_model_input_fn(batch)
    out_gen = self.gen(*gen_inp)
    phases, fake, real = [], None, None
    if self.discriminator_enabled:
      named, fake, real = self._disc_phase(batch, gen_inp[0], out_gen)
      phases.append(('disc', named))
    if self.generator_enabled:
      named, fake_g, real = self._gen_phase(batch, gen_inp[0], out_gen, real=real)
      fake = fake_g if fake_g is not None else fake
      phases.append(('gen', named))
    loss_metrics = {}
    for who, named in phases:
      self._apply_phase(who, named, loss_metrics.__setitem__)
    if not self.discriminator_enabled:
      fake = real = None
    return 1, loss_metrics, (batch, out_gen, fake, real)

  def _train_multiple_steps(self, loader):
    """disc_updates_per_step / gen_updates_per_step > 1: as many batches as the larger count are drawn first; the
    discriminator trains on the leading disc_updates_per_step of them, THEN the generator on the leading
    gen_updates_per_step -- every update with forward passes of its own (the generator's through the discriminator
    as updated so far; D(real) is run again only when feature matching needs it).  Returns the batches consumed, the
    per-name mean of the recorded values, and the tensors of the last update."""
    from itertools import islice
    from metrics import accumulate_metric
    wanted = max(self.disc_updates_per_step, self.gen_updates_per_step)
    batches = list(islice(iter(lambda: self._request_data(loader), None), wanted))
    if not batches:
      return 0, None, None
    schedule = [('disc', b) for b in batches[:self.disc_updates_per_step] if self.discriminator_enabled] + \
               [('gen', b) for b in batches[:self.gen_updates_per_step] if self.generator_enabled]
    matches_features = 'FeatureMatching' in self.gen_adv_criteria
    totals, last = {}, (None, None, None, None)
    for who, batch in schedule:
      gen_inp = self.train_model_input_fn(batch)
      out_gen = self.gen(*gen_inp)
      if who == 'disc':
        named, fake, real = self._disc_phase(batch, gen_inp[0], out_gen)
      else:
        named, fake, real = self._gen_phase(batch, gen_inp[0], out_gen, fresh_real=matches_features)
      self._apply_phase(who, named, lambda name, m: accumulate_metric(totals, name, m))
      last = (batch, out_gen, fake, real)
    if not self.discriminator_enabled:
      last = last[:2] + (None, None)
    return len(batches), {name: m.average() for name, m in totals.items()}, last

  def _val_step(self, loader, compute_metrics=True):
    batch = self._request_data(loader, volatile=True)
    if batch is None:
      return None, None
    gen_inp = self.test_model_input_fn(batch)
    out_gen = self.gen(*gen_inp)
    out_disc_fake = out_disc_real = None
    if self.disc is not None and self.val_disc_input_fn is not None:
      out_disc_fake = self.disc(nhwc=self.val_disc_input_fn(out_gen, gen_inp[0], out_gen,
                                                            is_real_input=False, detach=True))
      out_disc_real = self.disc(nhwc=self.val_disc_input_fn(batch['target'], gen_inp[0], out_gen,
                                                            is_real_input=True, detach=True))
    loss_metrics = {}
    if compute_metrics:
      for name, criterion in self.gen_criteria.items():
        loss_metrics['gen_loss_' + name] = get_loss_metric(criterion(out_gen, batch).detach())
    return loss_metrics, (batch, out_gen, out_disc_fake, out_disc_real)

  def _compute_train_metrics(self, data):
    cached = getattr(self, '_last_metrics', None)
    if cached is not None:            # computed inside the step (segment 4)
      self._last_metrics = None
      return cached
    metrics = {}
    for name, fn in self.train_gen_metric_fns.items():
      metrics['gen_' + name] = fn(data[1], data[0])
    if data[2] is not None:
      for name, fn in self.train_disc_metric_fns.items():
        metrics['disc_' + name] = fn(data[2]['prob'], data[3]['prob'], transform=False)
    return metrics

  def _compute_test_metrics(self, data):
    metrics = {}
    for name, fn in self.val_metric_fns.items():
      metrics['gen_' + name] = fn(data[1], data[0])
    if data[2] is not None:
      for name, fn in self.val_disc_metric_fns.items():
        metrics['disc_' + name] = fn(data[2]['prob'], data[3]['prob'], transform=False)
    return metrics

  def _set_train(self):
    self.gen.train()
    self.disc.train()

  def _set_test(self):
    self.gen.eval()
    if self.disc is not None:
      self.disc.eval()
