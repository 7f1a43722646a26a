import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
  sys.path.insert(0, REPO)

GOLDEN_DIR = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def _has_gpu():
  try:
    import torch
    return torch.cuda.is_available()
  except Exception:
    return False


def pytest_collection_modifyitems(config, items):
  if _has_gpu():
    return
  skip = pytest.mark.skip(reason='no HIP device in this container')
  for item in items:
    if 'gpu' in item.keywords:
      item.add_marker(skip)


@pytest.fixture(scope='session')
def golden():
  import numpy as np

  def load(name):
    with np.load(os.path.join(GOLDEN_DIR, name + '.npz')) as f:
      return {k: f[k] for k in f.files}
  return load


def took_about(seconds, limit, what):
  """Wall-clock expectations of CPU tests: a WARNING past `limit` (the figure the documentation
  quotes, measured on idle cores), a failure only past ten times that (a real regression) - the
  suite also runs on machines that are doing other things (four pytest workers x eight torch threads
  once stretched a 20 s tabulation to 894 s)."""
  import warnings
  assert seconds < 10.0 * limit, '{} took {:.1f} s (expected about {:.1f})'.format(what, seconds, limit)
  if seconds > limit:
    warnings.warn('{} took {:.1f} s (expected under {:.1f} on an idle machine)'.format(what, seconds, limit))

