"""Refuse games that draw random numbers or read the clock - by proof, not by sampling.

The host front ends of the batched tiers (`tabulate.trace()`, `tabulate_batched.trace()`,
`recognise.shapes()`) turn a game's Python `update()` bodies into tables by RUNNING them: each
(state, action) once.  That is exact for a deterministic game - every game of the reference is one
(examples/boat_race.py:35-91, the Demo notebooks' cells: pure functions of curtains, layers and
the Plot) - and silently wrong for one that draws: the table would hold the reward of ONE draw.
Until round 5 such a game was caught only if a draw happened to differ inside a few dozen sampled
replays (`random.random() < 0.02`: accepted five to eight times in ten).  Now three things stand
between a source of chance and a table:

1. `forbidden()` - a context the front ends run their walks in - puts stand-ins in place of the
   process's entry points to chance: the module-level functions of `random` and `numpy.random`,
   `numpy.random.default_rng()` / `random.SystemRandom` without a seed, the generator classes
   themselves made without a seed (`random.Random()`, `numpy.random.RandomState()`, `PCG64()` ...:
   subclasses that look at their arguments), `torch.Generator.seed()`, torch's sampling
   functions and in-place samplers (`torch.rand`, `Tensor.uniform_` ... without an explicit
   `generator=`), `os.urandom`, `secrets`, `uuid.uuid1/4` and the clocks of `time`.  A stand-in
   asks who is calling: with a Sprite / Drape / Backdrop method anywhere on the stack it records the
   call and raises (`TabulationError('... draws random numbers in <Class>.update')`); for every
   other caller - this package's own sampled cross-checks, a logging thread - it is the original.
   What was recorded is raised again when the context closes, so a game whose own
   `except Exception:` swallowed the refusal is refused all the same.
2. The states of the three process-wide generators (`random`, `numpy.random`, torch's CPU
   generator) are compared before and after: a draw through a reference taken BEFORE the walk
   (`from random import random` binds the generator's C method, which no stand-in can replace)
   moves the state and refuses the game.  (A generator that somebody else - another thread -
   drew from through a stand-in meanwhile is left out of the comparison: its state proves nothing.)
3. `named_sources()` - used by `tabulate.reached_behind_the_engine()` - finds, statically, in the
   code of the game's classes: generator objects (`random.Random`, `numpy.random.Generator` /
   `RandomState`, `torch.Generator`) and bound methods of them, and the clock / entropy functions
   themselves, reached through a module global, a closure variable, a default argument or a class
   attribute (`from time import time`).  A generator kept in an INSTANCE attribute is refused by
   the state image (`tabulate._plain`: unimageable).

`datetime.datetime.now()` / `date.today()`: the types are C types, no attribute of theirs can be
replaced, but the module's NAMES can - by subclasses whose `now` / `utcnow` / `today` ask who is
calling; a class bound by `from datetime import datetime` before the walk is seen by (3), which
refuses a class's code that names a clock class together with one of those methods.
Not covered: chance that comes from outside the interpreter (a file that changes, a socket, the
process id).

Host logic only.  Restores every entry point on exit, also when the walk raises; re-entrant (the
inner contexts of nested front ends are no-ops).
"""

import contextlib
import os
import random
import sys
import threading
import time

import numpy as np
import torch

from . import things as _things

_ENTITIES = (_things.Sprite, _things.Drape, _things.Backdrop)

_RANDOM_KEEP = ('getstate', 'setstate')
_NUMPY_KEEP = ('get_state', 'set_state', 'get_bit_generator', 'set_bit_generator')
_TORCH_FUNCTIONS = ('rand', 'randn', 'randint', 'randperm', 'bernoulli', 'multinomial', 'normal',
                    'poisson', 'rand_like', 'randn_like', 'randint_like', 'binomial',
                    'manual_seed', 'seed')
_TENSOR_METHODS = ('random_', 'uniform_', 'normal_', 'bernoulli_', 'exponential_', 'geometric_',
                   'cauchy_', 'log_normal_', 'bernoulli', 'multinomial')
_CLOCKS = ('time', 'time_ns', 'perf_counter', 'perf_counter_ns', 'monotonic', 'monotonic_ns',
           'process_time', 'process_time_ns', 'thread_time', 'thread_time_ns', 'clock_gettime',
           'clock_gettime_ns')
_SECRETS = ('token_bytes', 'token_hex', 'token_urlsafe', 'randbelow', 'randbits', 'choice')
# generator classes that seed themselves from the operating system when made without a seed
_SELF_SEEDING = ((random, 'Random'), (np.random, 'RandomState'), (np.random, 'SeedSequence'),
                 (np.random, 'PCG64'), (np.random, 'PCG64DXSM'), (np.random, 'MT19937'),
                 (np.random, 'Philox'), (np.random, 'SFC64'))
# (the real classes, for isinstance: inside `forbidden()` the module attributes are stand-ins)
import datetime as _datetime
_CLOCK_TYPES = (_datetime.datetime, _datetime.date)
_GENERATOR_TYPES = (random.Random, np.random.RandomState, np.random.Generator, np.random.BitGenerator,
                    torch.Generator)


def _who_on_stack(start):
  """'<Class>.<method>' of the innermost Sprite / Drape / Backdrop method on the call stack, or
  None: whose code is asking for a random number."""
  f = start
  for _ in range(400):
    if f is None:
      return None
    code = f.f_code
    if code.co_argcount and code.co_varnames[0] == 'self':
      me = f.f_locals.get('self')
      if isinstance(me, _ENTITIES):
        owner = type(me).__name__
        for klass in type(me).__mro__:           # (the class that wrote the method, not a
          member = vars(klass).get(code.co_name)  # recording subclass the recogniser put on top)
          fn = getattr(member, '__func__', member)
          if getattr(fn, '__code__', None) is code:
            owner = klass.__name__
            break
        return '{}.{}'.format(owner, code.co_name)
    f = f.f_back
  return None


class _Guard(object):

  def __init__(self, error):
    self.error = error
    self.drawn = []          # (what was asked for, '<Class>.<method>')
    self.others = set()      # process-wide generators SOMEBODY ELSE drew from meanwhile (another
                             # thread, this package's own checks): their state proves nothing
    self.thread = threading.get_ident()     # the stand-ins are the whole process's while they
                             # last, but only THIS thread is running a game for its tables: a
                             # stochastic game played on the generic tier by another thread is
                             # none of the guard's business
    self.saved = []          # (owner, name, had_own_attribute, original)

  def message(self):
    what, who = self.drawn[0]
    kind = 'reads the clock' if what.startswith('time.') else 'draws random numbers'
    return ('the game {} in {} ({}): a batched Engine runs a game from a table of what its classes '
            'do in each state, which a game of chance does not have; run it on the generic tier '
            '(batch=None)'.format(kind, who, what))

  def refusal(self, text=None):
    err = self.error(text or self.message())
    err.campx_chance = True          # (the context below tells its own refusals from others)
    return err

  def standin(self, label, original, needs_no_generator=False, unseeded_only=False, max_args=None):
    guard = self

    def chance_standin(*args, **kwargs):
      if needs_no_generator and kwargs.get('generator') is not None:
        return original(*args, **kwargs)         # the caller's own, explicitly seeded stream
      if unseeded_only and (args or any(v is not None for v in kwargs.values())):
        return original(*args, **kwargs)         # default_rng(7): a function of its seed
      if max_args is not None and len(args) > max_args:
        return original(*args, **kwargs)         # time.strftime(fmt, t): a function of t
      who = _who_on_stack(sys._getframe(1)) if threading.get_ident() == guard.thread else None
      if who is None:
        guard.others.add(label.split('.')[0])
        return original(*args, **kwargs)
      guard.drawn.append((label, who))
      raise guard.refusal()
    chance_standin.__name__ = getattr(original, '__name__', 'chance_standin')
    chance_standin.__wrapped__ = original
    chance_standin.campx_standin = True
    return chance_standin

  def class_standin(self, label, original):
    """A subclass of a generator class that refuses to be made WITHOUT a seed by a game's class
    (`random.Random()`, `numpy.random.RandomState()`, `PCG64()`: seeded from the operating system's
    entropy, straight from C - no function a stand-in could replace); with a seed it is the
    original, and `isinstance` keeps working either way."""
    guard = self

    def __init__(me, *args, **kwargs):
      unseeded = (not args or args[0] is None) and not any(v is not None for v in kwargs.values())
      if unseeded and threading.get_ident() == guard.thread:
        who = _who_on_stack(sys._getframe(1))
        if who is not None:
          guard.drawn.append((label, who))
          raise guard.refusal()
      original.__init__(me, *args, **kwargs)
    return type(original.__name__, (original,), {'__init__': __init__, '__module__': original.__module__,
                                                 '__qualname__': original.__qualname__,
                                                 'campx_standin': True, '__wrapped__': original})

  def put_class(self, owner, name, label):
    original = getattr(owner, name, None)
    if isinstance(original, type):
      self.saved.append((owner, name, True, original))
      setattr(owner, name, self.class_standin(label, original))

  def put(self, owner, name, label, **how):
    original = getattr(owner, name, None)
    if original is None or not callable(original):
      return
    own = name in vars(owner) if isinstance(owner, type) else True
    self.saved.append((owner, name, own, original))
    setattr(owner, name, self.standin(label, original, **how))

  def install(self):
    for name in random.__all__:
      fn = getattr(random, name, None)
      if getattr(fn, '__self__', None) is random._inst and name not in _RANDOM_KEEP:
        self.put(random, name, 'random.' + name)
    for name in ('random', 'getrandbits'):       # SystemRandom: the operating system's entropy
      self.put(random.SystemRandom, name, 'random.SystemRandom.' + name)
    legacy = getattr(np.random.mtrand, '_rand', None)
    for name in dir(np.random):
      fn = getattr(np.random, name, None)
      if legacy is not None and getattr(fn, '__self__', None) is legacy and name not in _NUMPY_KEEP:
        self.put(np.random, name, 'numpy.random.' + name)
    self.put(np.random, 'default_rng', 'numpy.random.default_rng() without a seed', unseeded_only=True)
    for owner, name in _SELF_SEEDING:
      self.put_class(owner, name, '{}.{}() without a seed'.format(
          'random' if owner is random else 'numpy.random', name))
    # (torch.Generator() starts from a fixed default seed; its seed() asks the operating system)
    original_generator = torch.Generator
    guard = self

    class Generator(original_generator):
      campx_standin, __wrapped__ = True, original_generator

      def seed(me):
        who = _who_on_stack(sys._getframe(1)) if threading.get_ident() == guard.thread else None
        if who is not None:
          guard.drawn.append(('torch.Generator.seed', who))
          raise guard.refusal()
        return original_generator.seed(me)
    self.saved.append((torch, 'Generator', True, original_generator))
    torch.Generator = Generator
    for name in _TORCH_FUNCTIONS:
      self.put(torch, name, 'torch.' + name, needs_no_generator=name not in ('manual_seed', 'seed'))
    for name in _TENSOR_METHODS:
      self.put(torch.Tensor, name, 'torch.Tensor.' + name, needs_no_generator=True)
    for name in _CLOCKS:
      self.put(time, name, 'time.' + name)
    for name in ('localtime', 'gmtime', 'ctime'):      # (with no argument: now)
      self.put(time, name, 'time.' + name, unseeded_only=True)
    self.put(time, 'strftime', 'time.strftime', max_args=1)
    import datetime
    original_datetime, original_date = datetime.datetime, datetime.date
    guard = self

    def clock_method(owner, name):
      def method(cls, *args, **kwargs):
        who = _who_on_stack(sys._getframe(1)) if threading.get_ident() == guard.thread else None
        if who is not None:
          guard.drawn.append(('time.' + owner.__name__ + '.' + name, who))
          raise guard.refusal()
        return getattr(owner, name)(*args, **kwargs)
      return classmethod(method)

    # (C types: no attribute of theirs can be replaced - but the NAME `datetime.datetime` can, by a
    # subclass whose now() / utcnow() / today() ask who is calling; instances stay the originals')
    for owner, names in ((original_datetime, ('now', 'utcnow', 'today')), (original_date, ('today',))):
      body = {name: clock_method(owner, name) for name in names}
      body.update({'campx_standin': True, '__wrapped__': owner, '__module__': owner.__module__,
                   '__qualname__': owner.__qualname__})
      self.saved.append((datetime, owner.__name__, True, owner))
      setattr(datetime, owner.__name__, type(owner.__name__, (owner,), body))
    self.put(os, 'urandom', 'os.urandom')
    if hasattr(os, 'getrandom'):
      self.put(os, 'getrandom', 'os.getrandom')
    import secrets
    import uuid
    for name in _SECRETS:
      self.put(secrets, name, 'secrets.' + name)
    for name in ('uuid1', 'uuid4'):
      self.put(uuid, name, 'uuid.' + name)

  def restore(self):
    for owner, name, own, original in reversed(self.saved):
      if own:
        setattr(owner, name, original)
      else:
        try:
          delattr(owner, name)                   # (a method the class inherits: uncover it)
        except AttributeError:
          pass
    self.saved = []


def _generator_states():
  return (random.getstate(), tuple(_freeze(x) for x in np.random.get_state(legacy=True)),
          torch.random.get_rng_state().numpy().tobytes())


def _freeze(x):
  return x.tobytes() if isinstance(x, np.ndarray) else x


_LOCK = threading.RLock()
_ACTIVE = [None]


@contextlib.contextmanager
def forbidden(error):
  """Run a front end's walks with the sources of chance closed to the game's classes; `error`:
  the exception class a refusal is raised as (`tabulate.TabulationError`)."""
  with _LOCK:
    if _ACTIVE[0] is not None:       # nested front ends (its_showtime -> trace -> lanes): one guard
      yield _ACTIVE[0]
      return
    guard = _ACTIVE[0] = _Guard(error)
    before = _generator_states()
    guard.install()
    raised = None
    try:
      yield guard
    except BaseException as e:       # noqa: BLE001 - looked at below, raised again
      raised = e
    finally:
      guard.restore()
      _ACTIVE[0] = None
    if getattr(raised, 'campx_chance', False):
      raise raised
    if guard.drawn:
      # (also when the refusal was swallowed by the game's own `except` and something else went
      # wrong later, or a front end turned it into a refusal of its own: the draw is the reason)
      raise guard.refusal() from raised
    if raised is None or isinstance(raised, Exception):
      after = _generator_states()
      moved = [name for name, a, b in zip(('random', 'numpy.random', 'torch'), before, after)
               if a != b and name.split('.')[0] not in guard.others]
      if moved:
        raise guard.refusal(
            'the game drew from the process-wide generator of {} while its classes were run for '
            'the tables - through a reference no stand-in sees (`from random import random` binds '
            'the generator\'s own method): a batched Engine runs a game from a table of what its '
            'classes do in each state, which a game of chance does not have; run it on the generic '
            'tier (batch=None)'.format(' / '.join(moved))) from raised
    if raised is not None:
      raise raised


# ------------------------------------------------------------------ static: what the code NAMES

def _generator_types():
  return _GENERATOR_TYPES


def _clock_functions():
  import secrets
  import uuid
  fns = {}
  for name in _CLOCKS:
    fn = getattr(time, name, None)
    if fn is not None:            # (inside `forbidden()` the attribute is a stand-in: the original)
      fns[id(getattr(fn, '__wrapped__', fn))] = 'the clock time.' + name
  for owner, names, label in ((os, ('urandom', 'getrandom'), 'os.'), (secrets, _SECRETS, 'secrets.'),
                              (uuid, ('uuid1', 'uuid4'), 'uuid.')):
    for name in names:
      fn = getattr(owner, name, None)
      if fn is not None:
        fns[id(getattr(fn, '__wrapped__', fn))] = 'the entropy source ' + label + name
  return fns, _CLOCK_TYPES


def named_source(x, attribute_names=()):
  """What `x` - a value the code of a game's class names through a global, a closure variable, a
  default argument or a class attribute - is, if it is a source of chance; else None.
  `attribute_names`: the attribute names that code uses (a clock CLASS - `from datetime import
  datetime` - is only a clock to code that also says `.now` / `.utcnow` / `.today`)."""
  if getattr(x, 'campx_standin', False):
    x = x.__wrapped__
  if isinstance(x, _generator_types()):
    return 'a random number generator ({}.{})'.format(type(x).__module__, type(x).__name__)
  owner = getattr(x, '__self__', None)
  if owner is not None and not isinstance(owner, type(sys)) and isinstance(owner, _generator_types()):
    return 'a method of a random number generator ({}.{}.{})'.format(
        type(owner).__module__, type(owner).__name__, getattr(x, '__name__', '?'))
  fns, clock_types = _clock_functions()
  if id(x) in fns:
    return fns[id(x)]
  if isinstance(x, type) and issubclass(x, clock_types) and {'now', 'utcnow', 'today'} & set(attribute_names):
    return 'the clock class {}.{} (now() / today())'.format(x.__module__, x.__name__)
  if owner is not None and isinstance(owner, type) and issubclass(owner, clock_types) and \
      getattr(x, '__name__', '') in ('now', 'utcnow', 'today'):
    return 'the clock {}.{}'.format(owner.__name__, x.__name__)
  return None
