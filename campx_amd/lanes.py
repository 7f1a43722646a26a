"""`Lanes`: a tensor that is N tensors at once, for running a game's own `update()` on many
states per call.

The reference's game classes are written in arithmetic on `[H, W]` tensors - `+ - * >= <=`,
`cat`, `sum`, `set_` (README.md:3; examples/boat_race.py:40-57: four `torch.cat` shifts blended
by the one-hot action, `gate = (b * (1 - layers[c])).sum()`, `self.curtain.set_(b)`) - precisely
so that they run on other tensor types.  `Lanes` is such a type: physically `[N, *shape]`,
logically `shape`, every torch function applied lane by lane (`torch.func.vmap` of the very
function the game called, so `dim=` arguments, slices, broadcasting and type promotion keep
their single-tensor meaning), in-place methods and `set_` redirected to the lanes' storage, and
Python-level reads - `bool()`, `int()`, `.item()`, a branch on a tensor - answered where every
lane agrees; where they do not, `Diverged` (a `CannotBatch`) carries what each lane reads, and
the tabulator runs the frame again for each group of lanes that agree
(campx_amd/tabulate_batched.py); anything else without a lane-by-lane meaning is a plain
`CannotBatch`: the host tabulator then falls back to one frame of Python per state and action
(campx_amd/tabulate.py).

Host logic; used by campx_amd/tabulate_batched.py only.
"""

import torch
from torch.utils import _pytree as pytree


class CannotBatch(BaseException):
  """The game did something that has no lane-by-lane meaning (the message says what).
  (A BaseException: a game class that wraps its own code in `except Exception:` must not swallow
  it - least of all `Diverged`, after which the frame would go on down the wrong branch for some
  of its states.)"""


class Diverged(CannotBatch):
  """A Python-level read - `if x:`, `int(x)`, `.item()` - of a value that is not the same in every
  lane.  `values` [N] says which lane reads what: whoever runs the frame may run it again for
  each group of lanes that agree (campx_amd/tabulate_batched.py does; anyone else sees a
  `CannotBatch`)."""

  def __init__(self, message, values):
    CannotBatch.__init__(self, message)
    self.values = values


def _guard():
  return torch._C.DisableTorchFunctionSubclass()


def plain(x):
  """The physical `[N, ...]` tensor of a `Lanes` (an alias: same storage)."""
  with _guard():
    return x.as_subclass(torch.Tensor)


def wrap(phys):
  """`phys` `[N, ...]` as N lanes of `phys.shape[1:]` (shares its storage)."""
  if phys.dim() < 1:
    raise ValueError('lanes need a leading axis')
  return torch.Tensor._make_subclass(Lanes, phys.detach())


def lanes_of(x):
  with _guard():
    return int(torch.Tensor.size(x, 0))


def _logical_shape(x):
  with _guard():
    return torch.Size(tuple(torch.Tensor.size(x))[1:])


def _name_of(func):
  owner = getattr(func, '__self__', None)          # a property getter: Tensor.shape.__get__
  if getattr(func, '__name__', '') in ('__get__', '__set__') and hasattr(owner, '__name__'):
    return owner.__name__
  return getattr(func, '__name__', repr(func))


def _uniform_value(x, what):
  """The one value every lane of a one-element `Lanes` holds, or `CannotBatch`."""
  p = plain(x)
  if p[0].numel() != 1:
    raise CannotBatch('{} of a tensor with {} elements'.format(what, p[0].numel()))
  flat = p.reshape(p.shape[0])
  if not bool((flat == flat[0]).all()):
    raise Diverged('{} of a value that differs between states (a data-dependent branch or '
                   'Python number: no lane-by-lane meaning)'.format(what), flat.clone())
  return flat[0]


def _generic(func, args, kwargs):
  """`func` lane by lane: vmap over every `Lanes` among the arguments."""
  leaves, spec = pytree.tree_flatten((args, kwargs))
  in_dims, flat, n = [], [], None
  for leaf in leaves:
    if isinstance(leaf, Lanes):
      p = plain(leaf)
      if n is None:
        n = p.shape[0]
      elif p.shape[0] != n:
        raise CannotBatch('tensors of {} and {} lanes in one operation'.format(n, p.shape[0]))
      flat.append(p)
      in_dims.append(0)
    else:
      flat.append(leaf)
      in_dims.append(None)

  def call(*xs):
    a, k = pytree.tree_unflatten(list(xs), spec)
    return func(*a, **k)

  try:
    with _guard():
      out = torch.vmap(call, in_dims=tuple(in_dims))(*flat)
  except CannotBatch:
    raise
  except (RuntimeError, ValueError, TypeError, NotImplementedError) as e:
    raise CannotBatch('{} has no lane-by-lane form ({}: {})'.format(
        _name_of(func), type(e).__name__, str(e).splitlines()[0][:160] if str(e) else ''))
  return pytree.tree_map(lambda o: wrap(o) if isinstance(o, torch.Tensor) else o, out)


# ---- fast paths: the handful of operations the reference's arithmetic-only game classes are
# made of (vmap costs ~0.3 ms per call on the host; these cost a few microseconds).  Each keeps
# the single-tensor semantics exactly - in particular type promotion, where a 0-d operand does
# not widen a tensor of its own category: the common dtype is taken from lane 0's operands and
# both sides are cast to it first, which is what the scalar operation does.
_BINARY = {}
for _n, _f in (('add', torch.add), ('sub', torch.sub), ('mul', torch.mul), ('ge', torch.ge),
               ('le', torch.le), ('gt', torch.gt), ('lt', torch.lt), ('eq', torch.eq), ('ne', torch.ne),
               ('maximum', torch.maximum), ('minimum', torch.minimum),
               ('bitwise_and', torch.bitwise_and), ('bitwise_or', torch.bitwise_or),
               ('bitwise_xor', torch.bitwise_xor)):
  _BINARY[_n] = (_f, False)
  _BINARY['__{}__'.format(_n)] = (_f, False)
  _BINARY['__r{}__'.format(_n)] = (_f, True)
_BINARY['rsub'] = (torch.sub, True)
_BINARY['__and__'] = (torch.bitwise_and, False)
_BINARY['__rand__'] = (torch.bitwise_and, True)
_BINARY['__or__'] = (torch.bitwise_or, False)
_BINARY['__ror__'] = (torch.bitwise_or, True)
_BINARY['__xor__'] = (torch.bitwise_xor, False)
_BINARY['__rxor__'] = (torch.bitwise_xor, True)
_UNARY = ('byte', 'char', 'short', 'int', 'long', 'half', 'float', 'double', 'bool', 'clone', 'detach',
          'contiguous', 'abs', 'neg', '__neg__', 'logical_not', 'bitwise_not', '__invert__', 'sign')


def _binary(f, a, b):
  """f(a, b) lane by lane for operands that are `Lanes`, plain tensors or Python numbers."""
  def exemplar(x):
    return plain(x)[0] if isinstance(x, Lanes) else x
  rank = max(len(_logical_shape(x)) if isinstance(x, Lanes) else (x.dim() if torch.is_tensor(x) else 0)
             for x in (a, b))
  rt = torch.result_type(exemplar(a), exemplar(b))

  def physical(x):
    if isinstance(x, Lanes):
      p = plain(x)
      return p.reshape((p.shape[0],) + (1,) * (rank - (p.dim() - 1)) + tuple(p.shape[1:])).to(rt)
    return x.to(rt) if torch.is_tensor(x) else x
  with _guard():
    return wrap(f(physical(a), physical(b)))


def _fast(name, func, args, kwargs):
  """The result, or NotImplemented where only the general path will do."""
  first = args[0] if args else None
  if name in _BINARY and len(args) == 2 and not kwargs:
    f, swapped = _BINARY[name]
    a, b = (args[1], args[0]) if swapped else args
    if all(isinstance(x, (Lanes, int, float, bool)) or (torch.is_tensor(x) and type(x) is torch.Tensor)
           for x in (a, b)):
      n = {lanes_of(x) for x in (a, b) if isinstance(x, Lanes)}
      if len(n) == 1:
        return _binary(f, a, b)
  if name == '__getitem__' and isinstance(first, Lanes) and len(args) == 2:
    index = args[1] if isinstance(args[1], tuple) else (args[1],)
    if all(i is None or i is Ellipsis or isinstance(i, (int, slice)) for i in index):
      with _guard():
        return wrap(plain(first)[(slice(None),) + index])
  if name == 'sum' and isinstance(first, Lanes) and len(args) == 1 and not kwargs:
    p = plain(first)
    with _guard():
      return wrap(p.reshape(p.shape[0], -1).sum(dim=1))
  if name in _UNARY and isinstance(first, Lanes) and len(args) == 1 and not kwargs:
    with _guard():
      return wrap(getattr(torch.Tensor, name)(plain(first)))
  if name in ('cat', 'concat', 'concatenate') and args and isinstance(args[0], (list, tuple)):
    parts = list(args[0])
    dim = args[1] if len(args) > 1 else kwargs.get('dim', 0)
    if (isinstance(dim, int) and set(kwargs) <= {'dim'} and
        all(isinstance(x, Lanes) or type(x) is torch.Tensor for x in parts)):
      n = {lanes_of(x) for x in parts if isinstance(x, Lanes)}
      if len(n) == 1:
        n = n.pop()
        phys = [plain(x) if isinstance(x, Lanes) else x.unsqueeze(0).expand((n,) + tuple(x.shape))
                for x in parts]
        with _guard():
          return wrap(torch.cat(phys, dim=dim + 1 if dim >= 0 else dim))
  return NotImplemented


_META = {
    'shape': lambda x: _logical_shape(x),
    'ndim': lambda x: len(_logical_shape(x)),
    'dtype': lambda x: plain(x).dtype,
    'device': lambda x: plain(x).device,
    'layout': lambda x: plain(x).layout,
    'is_cuda': lambda x: plain(x).is_cuda,
    'is_sparse': lambda x: False,
    'is_quantized': lambda x: False,
    'is_meta': lambda x: False,
    'requires_grad': lambda x: False,
    'grad': lambda x: None,
    'grad_fn': lambda x: None,
    'is_leaf': lambda x: True,
    'names': lambda x: (None,) * len(_logical_shape(x)),
}

_INPLACE_DUNDER = {'__iadd__': 'add', '__isub__': 'sub', '__imul__': 'mul', '__itruediv__': 'div',
                   '__ifloordiv__': 'floor_divide', '__imod__': 'remainder', '__iand__': 'bitwise_and',
                   '__ior__': 'bitwise_or', '__ixor__': 'bitwise_xor', '__ipow__': 'pow',
                   '__ilshift__': 'bitwise_left_shift', '__irshift__': 'bitwise_right_shift'}


def _store(dst, value):
  """Lanes `dst` <- `value` (a `Lanes` of as many lanes, or anything every lane gets), by copy
  into the lanes' own storage; `dst` keeps its identity, shape and dtype."""
  p = plain(dst)
  if isinstance(value, Lanes):
    v = plain(value)
    if v.shape[0] != p.shape[0]:
      raise CannotBatch('tensors of {} and {} lanes in one operation'.format(p.shape[0], v.shape[0]))
    extra = (p.dim() - 1) - (v.dim() - 1)
    if extra < 0:
      raise CannotBatch('in-place result does not fit its destination')
    v = v.reshape((v.shape[0],) + (1,) * extra + tuple(v.shape[1:]))
  else:
    v = value
  with _guard():
    p.copy_(v if torch.is_tensor(v) else torch.as_tensor(v))
  return dst


class Lanes(torch.Tensor):
  """See the module docstring."""

  def set_(self, source=None, *rest, **kwargs):
    """`Tensor.set_` (which does not pass through __torch_function__): these lanes now ARE
    `source` - a `Lanes`, or one tensor for every lane - and keep their identity, as a
    curtain re-bound with `self.curtain.set_(new)` (examples/boat_race.py:57) does."""
    if rest or kwargs or source is None or not torch.is_tensor(source):
      raise CannotBatch('set_ with a storage / offset / strides')
    if isinstance(source, Lanes):
      new = plain(source)
    else:
      new = source.unsqueeze(0).expand((lanes_of(self),) + tuple(source.shape)).contiguous()
    with _guard():
      torch.Tensor.set_(self, new)
    return self

  @classmethod
  def __torch_function__(cls, func, types, args=(), kwargs=None):
    kwargs = kwargs or {}
    name = _name_of(func)
    first = args[0] if args else None

    # ---- what a tensor says about itself: the logical tensor's answer
    if name in _META and isinstance(first, Lanes) and len(args) == 1:
      return _META[name](first)
    if name == 'size' and isinstance(first, Lanes):
      shape = _logical_shape(first)
      dim = args[1] if len(args) > 1 else kwargs.get('dim')
      return shape if dim is None else shape[dim]
    if name in ('dim', 'ndimension') and isinstance(first, Lanes):
      return len(_logical_shape(first))
    if name in ('numel', 'nelement') and isinstance(first, Lanes):
      return int(plain(first)[0].numel())
    if name == '__len__' and isinstance(first, Lanes):
      shape = _logical_shape(first)
      if not shape:
        raise TypeError('len() of a 0-d tensor')
      return shape[0]
    if name in ('is_floating_point', 'is_complex', 'is_signed', 'element_size', 'is_contiguous',
                'get_device', 'type') and isinstance(first, Lanes) and len(args) == 1 and not kwargs:
      return getattr(plain(first), name)()
    if name == '__iter__' and isinstance(first, Lanes):
      return iter([first[i] for i in range(len(first))])
    if name in ('__repr__', '__str__', '__format__'):
      return 'Lanes(n={}, shape={}, dtype={})'.format(lanes_of(first), tuple(_logical_shape(first)),
                                                      plain(first).dtype)
    if name == '__deepcopy__':
      return wrap(plain(first).clone())
    if name == '__hash__':
      return id(first)

    # ---- Python-level reads: only where every lane agrees
    if name in ('__bool__', '__int__', '__float__', '__index__', 'item', '__complex__'):
      v = _uniform_value(first, name.strip('_') + '()')
      with _guard():
        return getattr(v, name)()
    if name in ('tolist', 'numpy', '__array__') and isinstance(first, Lanes):
      # the whole tensor as a Python / numpy value: the one every lane holds - a READ-ONLY copy (a
      # write through it would have gone to the tensor's storage on plain tensors: not here, so
      # it must not pass silently) - or, where lanes differ, a `Diverged` whose groups are the
      # lanes with the same contents (a one-cell curtain: one group per cell it can be in)
      p = plain(first)
      flat = p.reshape(p.shape[0], -1)
      if flat.shape[0] > 1 and not bool((flat == flat[:1]).all()):
        _, inverse = torch.unique(flat, dim=0, return_inverse=True)
        raise Diverged('{}() of a tensor that differs between states'.format(name), inverse)
      with _guard():
        value = p[0].clone()
        if name == 'tolist':
          return value.tolist()
        array = value.numpy() if name == 'numpy' else value.numpy().__array__(*args[1:], **kwargs)
      array.flags.writeable = False
      return array
    if name in ('tolist', 'numpy', '__array__', '__array_wrap__', 'data_ptr', 'storage',
                'untyped_storage', '__reduce_ex__', '__dlpack__', 'cpu_', 'share_memory_'):
      raise CannotBatch('{}() of a tensor that stands for many states'.format(name))

    # ---- writes: into the lanes' own storage
    if name == '__setitem__':
      if not isinstance(first, Lanes):
        raise CannotBatch('a lane-varying value written into a tensor every state shares')
      index, value = args[1], args[2]
      index = index if isinstance(index, tuple) else (index,)
      if any(torch.is_tensor(i) for i in index):
        raise CannotBatch('item assignment through a tensor index')
      p = plain(first)
      if isinstance(value, Lanes):
        v = plain(value)
        with _guard():
          target = p[(slice(None),) + index]
        v = v.reshape((v.shape[0],) + (1,) * (target.dim() - v.dim()) + tuple(v.shape[1:]))
      else:
        v = value
      with _guard():
        p[(slice(None),) + index] = v
      return None
    inplace = _INPLACE_DUNDER.get(name)
    if inplace is None and name.endswith('_') and not name.endswith('__') and hasattr(torch.Tensor, name[:-1]):
      inplace = name[:-1]
    if inplace is not None:
      if not isinstance(first, Lanes):
        raise CannotBatch('a lane-varying value written into a tensor every state shares '
                          '({})'.format(name))
      if inplace == 'copy':
        return _store(first, args[1])
      if inplace in ('zero', 'fill'):
        value = 0 if inplace == 'zero' else args[1]
        return _store(first, value)
      result = _generic(getattr(torch.Tensor, inplace), args, kwargs)
      return _store(first, result)

    fast = _fast(name, func, args, kwargs)
    if fast is not NotImplemented:
      return fast
    return _generic(func, args, kwargs)
