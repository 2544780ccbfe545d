"""bench.py with A/B switches of csmri_hip.ops turned off: CSMRI_OFF=FANIN_TAPS,BN_SMALL python tools/bench_toggle.py [bench args]
CSMRI_SET="WGRAD_DEFER=4": integer module attributes of csmri_hip.ops set to a value."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from csmri_hip import ops  # noqa: E402

for name in filter(None, os.environ.get('CSMRI_OFF', '').split(',')):
  assert hasattr(ops, name), name
  setattr(ops, name, False)
for kv in filter(None, os.environ.get('CSMRI_SET', '').split(',')):
  k, v = kv.split('=')
  assert hasattr(ops, k), k
  setattr(ops, k, int(v))
# CSMRI_RUNNER="vgg_fork_late=1,other=0": class attributes of the adversarial runner (its A/B switches)
if os.environ.get('CSMRI_RUNNER'):
  from training.adversarial_runner import AdversarialRunner
  for kv in os.environ['CSMRI_RUNNER'].split(','):
    k, v = kv.split('=')
    setattr(AdversarialRunner, k, type(getattr(AdversarialRunner, k, 0))(int(v)) if hasattr(AdversarialRunner, k) else int(v))
bench.main()
