"""The update kernels come in two workgroup sizes (256 and 512 environments); the library
takes the larger from one workgroup per CU up, i.e. B >= 131 072 on an MI355X, which the
full-size tests cover at whole multiples.  Here: the same kernels at ragged batches."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_big_update_workgroups_at_ragged_batches():
  run = subprocess.run([sys.executable, os.path.join(HERE, 'big_workgroups_check.py')],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
  assert run.returncode == 0, run.stdout[-3000:]
  assert run.stdout.count('ok ') == 12, run.stdout[-3000:]      # 4 games x 3 batches
