"""Runner registry -- same plugin contract as the reference (training/__init__.py:3-17):
``build_runner(conf, runner_type, cuda, mode)`` -> module.build_runner(conf, cuda, mode)."""
import importlib

RUNNER_MODULES = {
    'standard': 'training.runner',
    'adversarial': 'training.adversarial_runner',
}


def build_runner(conf, runner_type, cuda, mode, *args, **kwargs):
  assert runner_type in RUNNER_MODULES, 'Unknown runner {}'.format(runner_type)
  module = importlib.import_module(RUNNER_MODULES[runner_type])
  return module.build_runner(conf, cuda, mode, *args, **kwargs)
