"""Host-side control logic that needs no GPU: the pretraining windows of the adversarial runner against the oracle
restatement pinned by fixture F11 (oracle.pretraining_flags <- reference training/adversarial_runner.py:195-209,273-298)."""
import itertools
import types

import pytest

import csmri_oracle as O


def _runner_module():
  from training import adversarial_runner
  return adversarial_runner


SPECS = [None, 1, 2, 3, (1, 2), (2, 4), (3, 5), (1, 6)]


def test_epoch_window_of_a_config_value():
  m = _runner_module()
  assert m._epoch_window(None) == (-1, -1)
  assert m._epoch_window(3) == (1, 4)
  assert m._epoch_window([2, 5]) == (2, 5)
  with pytest.raises(AssertionError):
    m._epoch_window((4, 4))


@pytest.mark.parametrize('gen_spec,disc_spec', list(itertools.product(SPECS, SPECS)))
def test_networks_enabled_equals_the_oracle_flags(gen_spec, disc_spec):
  """Every combination of generator / discriminator pretraining windows (overlapping ones included), epochs 1..7:
  the runner's stateless rule gives the flags the reference's stateful hook leaves behind."""
  m = _runner_module()
  me = types.SimpleNamespace(generator_pretraining_schedule=m._epoch_window(gen_spec),
                             discriminator_pretraining_schedule=m._epoch_window(disc_spec))
  for epoch in range(1, 8):
    got = m.AdversarialRunner._networks_enabled(me, epoch)
    assert got == O.pretraining_flags(epoch, gen_spec, disc_spec), (epoch, gen_spec, disc_spec, got)
