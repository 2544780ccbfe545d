#!/usr/bin/env python3
"""Training entry point with the reference's CLI surface (train.py:41-62):

  python train.py [-c CUDA] [-v] [--conf k=v ...] [--run-dir D] [--resume CKPT] config.json

``-c 0`` one GPU; ``-c 0,1,..`` spawns one process per listed GPU (RCCL data
parallel).  The proprietary dataset is not available, so batches come from
data.synthetic (``--conf image_size=256 steps_per_epoch=50``).  Validation,
TensorBoard and early stopping of the reference's loop are outside the hot path
(SURVEY 8f)."""
import argparse
import logging
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
  sys.path.insert(0, HERE)


def parse_args(argv=None):
  p = argparse.ArgumentParser(description='Train CS-MRI RecNet / GAN refinement on MI355X')
  p.add_argument('-c', '--cuda', default='0', help="GPU ids, e.g. '0' or '0,1,2,3'")
  p.add_argument('-v', '--verbose', action='store_true')
  p.add_argument('-p', '--print-model', action='store_true')
  p.add_argument('--print-parameters', action='store_true')
  p.add_argument('--dry', action='store_true', help='build everything, train nothing')
  p.add_argument('--conf', nargs='+', default=[], help='key=value overrides')
  p.add_argument('--data-dir', default=None)
  p.add_argument('--log-dir', default=None)
  p.add_argument('--run-dir', default=None)
  p.add_argument('--resume', default=None)
  p.add_argument('config')
  return p.parse_args(argv)


def main(argv=None):
  args = parse_args(argv)
  gpus = [g for g in args.cuda.split(',') if g != '']
  if len(gpus) > 1 and 'LOCAL_RANK' not in os.environ:
    # one process per GPU (the reference wraps the model in nn.DataParallel instead)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node', str(len(gpus)), '--master-addr', '127.0.0.1',
           '--master-port', os.environ.get('MASTER_PORT', '29511'), os.path.abspath(__file__)]
    cmd += (argv if argv is not None else sys.argv[1:])
    env = dict(os.environ, HIP_VISIBLE_DEVICES=','.join(gpus))
    return subprocess.call(cmd, env=env)

  import torch
  import utils
  from utils.config import Configuration
  from utils.checkpoints import restore_checkpoint, save_checkpoint
  from utils.checkpoint_paths import get_periodic_checkpoint_path
  from models.utils import set_default_compute_dtype
  from training import build_runner
  from training import distributed as dist_utils
  from data.synthetic import SyntheticLoader

  logging.basicConfig(level=logging.DEBUG if args.verbose else logging.INFO,
                      format='%(asctime)s %(message)s')
  conf = Configuration.from_json(args.config)
  conf.update(dict(kv.split('=', 1) for kv in args.conf))
  dist_utils.init_from_env()
  utils.set_random_seeds(conf.seed)
  set_default_compute_dtype(conf.get_attr('compute_dtype', default='bf16'))
  runner = build_runner(conf, conf.runner_type, args.cuda, 'train')
  if args.print_model:
    print(runner)
  start_epoch = 1
  if args.resume:
    state = restore_checkpoint(args.resume, runner, args.cuda)
    # the reference stores the NEXT epoch in periodic checkpoints (train.py:295: save(..., epoch + 1, ...))
    start_epoch = state.get('start_epoch', 1)
  if args.dry:
    return 0
  size = conf.get_attr('image_size', default=512 // conf.get_attr('downscale', default=1))
  ws = dist_utils.world_size()
  loader = SyntheticLoader(conf.batch_size * ws, size, size,
                           conf.get_attr('steps_per_epoch', default=20),
                           acc=conf.undersampling['acceleration_factor'], seed=conf.seed,
                           shard=(dist_utils.rank(), ws))      # each rank synthesises its own rows only
  for epoch in range(start_epoch, conf.num_epochs + 1):
    runner.epoch_beginning(epoch)
    t0 = time.time()
    losses, metrics = runner.train_epoch(loader, epoch, None, conf.get_attr('steps_per_train_summary', 1),
                                         args.verbose)
    torch.cuda.synchronize()
    runner.epoch_finished(epoch)
    if dist_utils.rank() == 0:
      logging.info('Epoch %d: %.1fs  %s  %s', epoch, time.time() - t0,
                   ', '.join('%s: %s' % kv for kv in losses.items()),
                   ', '.join('%s: %s' % kv for kv in metrics.items()))
      if args.run_dir:
        os.makedirs(args.run_dir, exist_ok=True)
        # the stored epoch is the NEXT one to train (reference train.py:286-296 save_periodic_checkpoint(..., epoch + 1))
        save_checkpoint(get_periodic_checkpoint_path(args.run_dir, epoch), conf, runner, epoch + 1)
  return 0


if __name__ == '__main__':
  sys.exit(main())
