"""User-style games that poke at the lane tabulator's edges (tests/test_tabulate_batched.py): things a
game class may legally do on plain tensors - `torch.where`, writes through views, gathers, another
thing's curtain re-bound, the board read as characters, NaN rewards, Python counters, Plot entries,
random draws, `nonzero()`, branches on a tensor, z-order changes.  Each is either tabulated on lanes
identically to the one-frame-per-play walk, or handed to that walk (which takes it or refuses it
with its own message); never tabulated differently."""

import torch

from campx import things
from campx.ascii_art import ascii_art_to_game

ART = ['######', '#A   #', '#  G #', '# B  #', '######']
def shifted(b):
  return [torch.cat([b[:, 1:], b[:, :1]], dim=1), torch.cat([b[:, -1:], b[:, :-1]], dim=1),
          torch.cat([b[1:], b[:1]], dim=0), torch.cat([b[-1:], b[:-1]], dim=0), b]
def moved(act, b):
  s = shifted(b)
  return sum(act[i] * s[i] for i in range(5))

class Base(things.Drape):
  def step(self, act, layers, solid='#'):
    there = moved(act, self.curtain)
    hit = (there * layers[solid]).sum()
    free = (1 - (hit >= 1).long()).byte()
    return (free * there) + ((1 - free) * self.curtain)

class Where(Base):       # torch.where on layers
  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None: return
    act = actions.byte()
    there = moved(act, self.curtain)
    blocked = (there * layers['#']).sum() >= 1
    self.curtain.set_(torch.where(blocked, self.curtain, there))
    the_plot.add_reward((self.curtain * layers['G']).sum().float())

class Counter(Base):     # python int attribute
  def __init__(self, curtain, character):
    super().__init__(curtain, character); self.n = 0
  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None: return
    self.n = min(self.n + 1, 3)
    self.curtain.set_(self.step(actions.byte(), layers))
    the_plot.add_reward(float(self.n))

class PlotMem(Base):     # remembers the previous curtain in the plot (a clone) and rewards on standing still
  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None: return
    prev = the_plot.get('prev')
    self.curtain.set_(self.step(actions.byte(), layers))
    if prev is not None:
      the_plot.add_reward((prev * self.curtain).sum().float())
    the_plot['prev'] = self.curtain.clone()

class Rand(Base):
  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None: return
    self.curtain.set_(self.step(actions.byte(), layers))
    the_plot.add_reward(float(torch.rand(()) > 0.5))

class Nonzero(Base):
  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None: return
    self.curtain.set_(self.step(actions.byte(), layers))
    pos = self.curtain.nonzero()
    the_plot.add_reward(float(pos[0, 0] * 10 + pos[0, 1]))

class Roll(Base):
  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None: return
    a = int(actions.argmax())
    shift, axis = ((-1, 1), (1, 1), (-1, 0), (1, 0), (0, 0))[a]
    there = torch.roll(self.curtain, shift, axis)
    hit = (there * layers['#']).sum()
    free = (1 - (hit >= 1).long()).byte()
    self.curtain.set_((free * there) + ((1 - free) * self.curtain))
    the_plot.add_reward(self.curtain.float().mul(torch.arange(30.).reshape(5, 6)).sum())

class Terminator(Base):   # terminates on G: state-dependent termination through a tensor
  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None: return
    self.curtain.set_(self.step(actions.byte(), layers))
    on = (self.curtain * layers['G']).sum()
    the_plot.add_reward(on.float())
    if on > 0:
      the_plot.terminate_episode()

class Grower(Base):   # curtain grows: multi-cell drape (trail)
  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None: return
    head = self.step(actions.byte(), layers)
    self.curtain.set_(((self.curtain + head) >= 1).byte())

class Indexer(Base):  # gather with a lane tensor index
  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None: return
    self.curtain.set_(self.step(actions.byte(), layers))
    flat = self.curtain.reshape(-1).long()
    idx = (flat * torch.arange(30)).sum()
    table = torch.arange(30.) * 0.5
    the_plot.add_reward(table[idx])

class Float(Base):   # float curtain arithmetic / division
  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None: return
    self.curtain.set_(self.step(actions.byte(), layers))
    m = self.curtain.float()
    rows = m.sum(dim=1)
    the_plot.add_reward((rows * torch.tensor([0., 1., 2., 3., 4.])).sum() / 3)

class ChangeZ(Base):
  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None: return
    self.curtain.set_(self.step(actions.byte(), layers))
    if actions[4] == 1:
      the_plot.change_z_order('A', 'G')

class ViewWrite(Base):
  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None: return
    new = self.step(actions.byte(), layers)
    row = self.curtain[1]
    row.zero_()
    self.curtain[2:] = 0
    self.curtain.add_(new)
    the_plot.add_reward((self.curtain * layers['G']).sum().float() * 2)

class Pusher(Base):     # moves ANOTHER thing's curtain
  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None: return
    act = actions.byte()
    there = moved(act, self.curtain)
    b = all_things['B'].curtain
    push = (there * b).sum().byte()
    b_there = moved(act, b)
    b_free = (1 - ((b_there * layers['#']).sum() >= 1).long()).byte()
    ok = push * b_free
    b.set_((ok * b_there) + ((1 - ok) * b))
    blocked = ((there * layers['#']).sum() >= 1).long().byte() + (push * (1 - b_free))
    free = (1 - (blocked >= 1).long()).byte()
    self.curtain.set_((free * there) + ((1 - free) * self.curtain))

class Still(things.Drape):
  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    pass

class BoardReader(Base):   # reads `board` (int64 ords) and backdrop
  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None: return
    there = moved(actions.byte(), self.curtain)
    wall = (board == ord('#')).byte()
    hit = (there * wall).sum() + (there * (board == ord('B')).byte()).sum()
    free = (1 - (hit >= 1).long()).byte()
    self.curtain.set_((free * there) + ((1 - free) * self.curtain))
    the_plot.add_reward(((board == ord('G')).byte() * self.curtain).sum().float() + (backdrop.curtain == ord(' ')).sum().float() * 0)

class Chaser(Base):      # B chases A: reads all_things['A'].curtain
  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None: return
    a = all_things['A'].curtain.float()
    me = self.curtain.float()
    cols = torch.arange(6.); rows = torch.arange(5.)
    dx = (a.sum(0) * cols).sum() - (me.sum(0) * cols).sum()
    dy = (a.sum(1) * rows).sum() - (me.sum(1) * rows).sum()
    go_right = (dx > 0).byte(); go_left = (dx < 0).byte()
    s = shifted(self.curtain)
    there = go_left * s[0] + go_right * s[1] + (1 - go_left - go_right) * s[4]
    hit = (there * layers['#']).sum() + (there * layers['A']).sum()
    free = (1 - (hit >= 1).long()).byte()
    self.curtain.set_((free * there) + ((1 - free) * self.curtain))
    the_plot.add_reward(-(dx.abs() + dy.abs()))

class Mover(Base):
  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None: return
    self.curtain.set_(self.step(actions.byte(), layers))

class NanReward(Base):
  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None: return
    self.curtain.set_(self.step(actions.byte(), layers))
    on = (self.curtain * layers['G']).sum().float()
    the_plot.add_reward(on / on)       # nan off the goal, 1 on it

class Discounter(Base):
  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None: return
    self.curtain.set_(self.step(actions.byte(), layers))
    if actions[0] == 1:
      the_plot.terminate_episode(0.5)


class IntIndex(Base):   # int() of a value that differs between states, as an index into a Python list
  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    self.curtain.set_(self.step(actions.byte(), layers))
    where = int((self.curtain.reshape(-1).long() * torch.arange(30)).sum())
    the_plot.add_reward([0.25 * (i % 7) for i in range(30)][where])


class TwoBranches(Base):   # a branch in 'A' and, later in the same frame, one in 'B': groups split again
  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    new = self.step(actions.byte(), layers)
    if (new * layers['G']).sum() > 0:
      the_plot.add_reward(5.0)
      if actions[4] == 1:
        the_plot.terminate_episode(0.5)
    self.curtain.set_(new)


class Follower(Base):      # 'B': steps towards the left whenever 'A' stands in the upper half
  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    if all_things['A'].curtain[:2].sum() > 0:
      act = torch.tensor([1, 0, 0, 0, 0], dtype=torch.uint8)
    else:
      act = torch.tensor([0, 0, 0, 0, 1], dtype=torch.uint8)
    there = moved(act, self.curtain)
    hit = (there * layers['#']).sum() + (there * all_things['A'].curtain).sum()
    if hit == 0:
      self.curtain.set_(there)
      the_plot.add_reward(-0.5)


class LeavesAMark(Base):   # sets a Plot entry, THEN something later in the frame branches
  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    the_plot['mark'] = 1
    self.curtain.set_(self.step(actions.byte(), layers))


class NumpyReader(Base):   # finds itself through numpy views, as tests/traced_games.py's classes do
  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    import numpy as np
    a = int(actions.argmax())
    dr, dc = ((0, -1), (0, 1), (-1, 0), (1, 0), (0, 0))[a]
    (r,), (c,) = np.nonzero(self.curtain.numpy())
    wall = all_things['#'].curtain.numpy()
    if not wall[r + dr, c + dc] and not all_things['B'].curtain[r + dr, c + dc]:
      self.curtain.zero_()
      self.curtain[r + dr, c + dc] = 1
    the_plot.add_reward(float(all_things['G'].curtain.numpy()[r, c]) - 0.125)


class NumpyWriter(Base):   # moves by writing THROUGH the numpy view of its curtain (shared memory on plain tensors)
  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    import numpy as np
    a = int(actions.argmax())
    dr, dc = ((0, -1), (0, 1), (-1, 0), (1, 0), (0, 0))[a]
    view = self.curtain.numpy()
    (r,), (c,) = np.nonzero(view)
    if not all_things['#'].curtain.numpy()[r + dr, c + dc]:
      view[r, c] = 0
      view[r + dr, c + dc] = 1


def game(a, b=Still):
  def build():
    return ascii_art_to_game(ART, what_lies_beneath=' ',
                             drapes={'A': a, 'B': b, '#': things.FixedDrape, 'G': things.FixedDrape},
                             z_order='G#BA', update_schedule='AB#G')
  return build


# (class of 'A', class of 'B', what the auto walker's LAST_WALK must start with, or the refusal both give)
CASES = [(Where, Still, 'lanes: '), (Counter, Still, 'one frame per play (lanes: something besides the curtains'),
         (ViewWrite, Still, 'lanes: '), (PlotMem, Still, "one frame per play (lanes: the_plot['prev']"),
         (Rand, Still, 'REFUSED: draws random numbers in Rand.update (torch.rand)'), (Nonzero, Still, 'one frame per play (lanes: nonzero'),
         (Roll, Still, 'lanes: '), (Terminator, Still, 'lanes: '),
         (Grower, Still, 'REFUSED: cover several cells that come and go - 12 tracked cells'), (Indexer, Still, 'one frame per play (lanes: __getitem__'),
         (Float, Still, 'lanes: '), (ChangeZ, Still, 'one frame per play (lanes: the game changes the z-order'),
         (Pusher, Still, 'lanes: '), (BoardReader, Still, 'lanes: '), (Mover, Chaser, 'lanes: '),
         (NanReward, Still, 'lanes: '), (Discounter, Still, 'lanes: '), (IntIndex, Still, 'lanes: '),
         (TwoBranches, Follower, 'lanes: '), (Terminator, Follower, 'lanes: '),
         (LeavesAMark, Follower, 'one frame per play (lanes: something besides the curtains'),
         (NumpyReader, Still, 'lanes: '), (NumpyReader, Follower, 'lanes: '),
         (NumpyWriter, Still, 'one frame per play (lanes: ValueError')]


# ---- live game objects reached behind the engine's back: refused statically, by name
REGISTRY = {}


class Spy(things.Drape):
  """Rewards by where 'A' stands - read through a module global, not through all_things."""

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    a = REGISTRY['A'].curtain
    the_plot.add_reward((a * torch.arange(30, dtype=torch.uint8).reshape(5, 6)).sum().float())


def spy_through_a_global():
  eng = game(Mover, Spy)()
  REGISTRY['A'] = eng.things['A']
  return eng


def spy_through_a_closure():
  seen = []

  class ClosureSpy(things.Drape):
    def update(self, actions, board, layers, backdrop, all_things, the_plot):
      if actions is None:
        return
      the_plot.add_reward((seen[0].curtain * torch.arange(30, dtype=torch.uint8).reshape(5, 6)).sum().float())

  eng = game(Mover, ClosureSpy)()
  seen.append(eng.things['A'])
  return eng


def spy_through_a_default_argument():
  box = {}

  class DefaultSpy(things.Drape):
    def update(self, actions, board, layers, backdrop, all_things, the_plot, where=box):
      if actions is None:
        return
      a = where['engine'].things['A'].curtain
      the_plot.add_reward((a * torch.arange(30, dtype=torch.uint8).reshape(5, 6)).sum().float())

  eng = game(Mover, DefaultSpy)()
  box['engine'] = eng
  return eng


def _helper_reward():
  return (REGISTRY['A'].curtain * torch.arange(30, dtype=torch.uint8).reshape(5, 6)).sum().float()


class HelperSpy(things.Drape):
  """... through a helper function of the user's module."""

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    the_plot.add_reward(_helper_reward())


def spy_through_a_helper():
  eng = game(Mover, HelperSpy)()
  REGISTRY['A'] = eng.things['A']
  return eng


SPIES = [(spy_through_a_global, "a live Mover through the module global 'REGISTRY'"),
         (spy_through_a_closure, "a live Mover through the closure variable 'seen'"),
         (spy_through_a_default_argument, 'a live Engine through a default argument'),
         (spy_through_a_helper, "a live Mover through the module global 'REGISTRY' (named by _helper_reward())")]


class AttrSpy(things.Drape):
  """... through an attribute of its own instance: deep copies keep that consistent (the copy's
  partner is the copy's 'A'), so this one IS tabulated - and must predict live play."""
  partner = None

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    the_plot.add_reward((self.partner.curtain * torch.arange(30, dtype=torch.uint8).reshape(5, 6)).sum().float())


def spy_through_its_own_attribute():
  eng = game(Mover, AttrSpy)()
  eng.things['B'].partner = eng.things['A']
  return eng


class ClassAttrSpy(AttrSpy):
  pass


def spy_through_a_class_attribute():
  eng = game(Mover, ClassAttrSpy)()
  ClassAttrSpy.partner = eng.things['A']       # (classes are shared between deep copies)
  return eng


SPIES.append((spy_through_a_class_attribute, 'a live Mover through the class attribute ClassAttrSpy.partner'))


# ---- a Sprite agent, PyColab style: Python ints and branches; and a crate that reads where it stands
class Porter(things.Sprite):
  """Walks one cell; walls and a crate that cannot give way stop it (it looks a cell further)."""

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    dr, dc = ((0, -1), (0, 1), (-1, 0), (1, 0), (0, 0))[int(actions.argmax())]
    r, c = self.position.row + dr, self.position.col + dc
    if layers['#'][r, c]:
      return
    if all_things['X'].curtain[r, c]:
      if layers['#'][r + dr, c + dc]:
        return                                   # the crate is against a wall: nobody moves
    self._position = self.Position(r, c)
    the_plot.add_reward(-0.25)


class Crate(things.Drape):
  """Updated AFTER the porter: if he now stands on it, it moves on one cell the way he came."""

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    p = all_things['P'].position
    if self.curtain[p.row, p.col]:
      dr, dc = ((0, -1), (0, 1), (-1, 0), (1, 0), (0, 0))[int(actions.argmax())]
      self.curtain[p.row, p.col] = 0
      self.curtain[p.row + dr, p.col + dc] = 1
      if all_things['G'].curtain[p.row + dr, p.col + dc]:
        the_plot.add_reward(5.0)
        the_plot.terminate_episode()


PORTER_ART = ['#######', '#P    #', '# X   #', '#   G #', '#     #', '#######']


def porter(art=None):
  def build():
    return ascii_art_to_game(art or PORTER_ART, what_lies_beneath=' ', sprites={'P': Porter},
                             drapes={'X': Crate, '#': things.FixedDrape, 'G': things.FixedDrape},
                             z_order='G#XP', update_schedule='PX#G')
  return build


class Swallower(Base):     # wraps its branch in `except Exception`: must not swallow the frame's split
  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    new = self.step(actions.byte(), layers)
    try:
      if (new * layers['G']).sum() > 0:
        the_plot.add_reward(2.0)
    except Exception:       # noqa: BLE001
      the_plot.add_reward(-100.0)
    self.curtain.set_(new)


CASES.append((Swallower, Still, 'lanes: '))
