import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'csmri-refinement_amd')
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
for p in (PKG, os.path.join(ROOT, 'oracle'), ROOT):
  if p not in sys.path:
    sys.path.insert(0, p)


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def golden_dir():
  return GOLDEN
