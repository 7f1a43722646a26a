"""Import the reference (OpenMined/CampX at /root/reference) under modern torch.

Build-owned test infrastructure; runs only where /root/reference exists (the
build container), never on the GPU box.  Nothing from the reference is copied:
its modules are imported from where they lie, and notebook classes are obtained
by exec-ing cell source read from the .ipynb at run time.

The reference targets torch 0.3.1 / numpy 1.15 and imports PySyft and pycolab at
module load.  The shims below make it importable without
editing it:

* stub modules `syft`, `syft.core.frameworks.torch.utils`,
  `pycolab.protocols.logging`;
* `np.vstack` accepts a generator, `np.fromstring(str)` decodes bytes;
* tensor comparisons return uint8 (0.3.1 ByteTensor semantics) and uint8 masks
  index like bool masks.

IMPORTANT: the reference package is called `campx`, the same name as this repo's
alias package.  Use this module only from a process whose sys.path puts
/root/reference first and which never imports the repo's `campx` alias
(`make_golden.py` is such a process).
"""

import json
import os
import sys
import types

REFERENCE_ROOT = '/root/reference'


def available():
  return os.path.isdir(os.path.join(REFERENCE_ROOT, 'campx'))


def _install_stub_modules():
  syft = types.ModuleType('syft')
  syft._PointerTensor = type('_PointerTensor', (), {})
  syft._SNNTensor = type('_SNNTensor', (), {})
  utils = types.ModuleType('syft.core.frameworks.torch.utils')
  import torch
  utils.is_tensor = torch.is_tensor
  chain = ['syft', 'syft.core', 'syft.core.frameworks',
           'syft.core.frameworks.torch']
  mods = {'syft': syft}
  for name in chain[1:]:
    mods[name] = types.ModuleType(name)
  mods['syft.core.frameworks.torch.utils'] = utils
  mods['syft.core.frameworks.torch'].utils = utils
  pycolab = types.ModuleType('pycolab')
  protocols = types.ModuleType('pycolab.protocols')
  logging = types.ModuleType('pycolab.protocols.logging')
  logging.log = lambda the_plot, message: the_plot.setdefault(
      'log_messages', []).append(message)
  protocols.logging = logging
  pycolab.protocols = protocols
  mods.update({'pycolab': pycolab, 'pycolab.protocols': protocols,
               'pycolab.protocols.logging': logging})
  for name, mod in mods.items():
    sys.modules.setdefault(name, mod)


def _install_numpy_shims():
  import numpy as np
  if getattr(np, '_campx_shimmed', False):
    return
  real_vstack = np.vstack

  def vstack(tup, *a, **k):
    if isinstance(tup, types.GeneratorType):
      tup = list(tup)
    return real_vstack(tup, *a, **k)

  def fromstring(s, dtype=float, *a, **k):
    if isinstance(s, str):
      s = s.encode('latin-1')
    return np.frombuffer(s, dtype=dtype).copy()

  np.vstack = vstack
  np.fromstring = fromstring
  np._campx_shimmed = True


def _install_torch_shims():
  import torch
  T = torch.Tensor
  if getattr(T, '_campx_shimmed', False):
    return

  def as_byte(fn):
    def wrapped(self, other):
      out = fn(self, other)
      if torch.is_tensor(out) and out.dtype == torch.bool:
        out = out.to(torch.uint8)
      return out
    return wrapped

  for name in ('__eq__', '__ne__', '__ge__', '__le__', '__gt__', '__lt__'):
    setattr(T, name, as_byte(getattr(T, name)))
  T.__hash__ = lambda self: id(self)

  def boolify(index):
    if torch.is_tensor(index) and index.dtype == torch.uint8:
      return index.bool()
    if isinstance(index, tuple):
      return tuple(boolify(i) for i in index)
    return index

  real_get, real_set = T.__getitem__, T.__setitem__
  T.__getitem__ = lambda self, idx: real_get(self, boolify(idx))
  T.__setitem__ = lambda self, idx, val: real_set(self, boolify(idx), val)
  T._campx_shimmed = True


_loaded = None


def load():
  """Import the reference and return a namespace of its modules."""
  global _loaded
  if _loaded is not None:
    return _loaded
  if not available():
    raise RuntimeError('reference tree not present at ' + REFERENCE_ROOT)
  for p in (os.path.join(REFERENCE_ROOT, 'examples'), REFERENCE_ROOT):
    if p in sys.path:
      sys.path.remove(p)
    sys.path.insert(0, p)
  if 'campx' in sys.modules and not sys.modules['campx'].__file__.startswith(
      REFERENCE_ROOT):
    raise RuntimeError('the repo\'s `campx` alias is already imported; run the '
                       'reference harness in its own process')
  _install_stub_modules()
  _install_numpy_shims()
  _install_torch_shims()
  import campx
  from campx import things, engine, ascii_art, plot, rendering
  import boat_race
  assert campx.__file__.startswith(REFERENCE_ROOT)
  _loaded = types.SimpleNamespace(
      campx=campx, things=things, engine=engine, ascii_art=ascii_art,
      plot=plot, rendering=rendering, boat_race=boat_race)
  return _loaded


def notebook_cells(name):
  """Code cells of an example notebook: list of (cell_index, source, outputs)."""
  path = os.path.join(REFERENCE_ROOT, 'examples', name)
  with open(path) as f:
    nb = json.load(f)
  cells = []
  for i, cell in enumerate(nb['cells']):
    if cell['cell_type'] == 'code':
      cells.append((i, ''.join(cell['source']), cell.get('outputs', [])))
  return cells


def notebook_namespace(name, cell_indices):
  """Exec the given cells of a notebook; returns the resulting globals."""
  ref = load()
  import collections
  import itertools
  import numpy as np
  import six
  import torch
  ns = dict(torch=torch, np=np, six=six, itertools=itertools,
            collections=collections, things=ref.things, engine=ref.engine,
            ascii_art_to_game=ref.ascii_art.ascii_art_to_game,
            Partial=ref.ascii_art.Partial)
  sources = {i: src for i, src, _ in notebook_cells(name)}
  for i in cell_indices:
    exec(compile(sources[i], '{}[cell {}]'.format(name, i), 'exec'), ns)
  return ns
