"""Games made of arbitrary Python classes, written here and nowhere in the rule library.

They exist to exercise `campx_amd.tabulate`: a batched engine has to run them on the HIP
table + render kernels although their `update()` bodies are loops, numpy, Python floats -
nothing `campx_amd.rules` / `gamespec.lower()` knows.  Written against the reference's
entity API only (`things.Drape` / `things.Sprite`, `update(actions, board, layers,
backdrop, things, the_plot)`, campx/things.py:161-392).
"""

import numpy as np
import torch

from campx import things
from campx.ascii_art import ascii_art_to_game, Partial

_DELTA = [(0, -1), (0, 1), (-1, 0), (1, 0), (0, 0)]   # left, right, up, down, stay


def _action_id(actions):
  a = np.asarray(actions.tolist() if torch.is_tensor(actions) else actions)
  assert a.sum() == 1
  return int(np.argmax(a))


# ------------------------------------------------------------------ one mover: ice rink

ICE_ART = ['#########',
           '#A  o   #',
           '# ##  # #',
           '#   o#  #',
           '#  #  oE#',
           '#########']


class IceSkater(things.Drape):
  """Slides in the action's direction until the next cell is a wall: up to seven cells in
  one frame.  Pays 0.5 per coin tile it crosses or stops on, minus 0.125 per frame;
  stopping on the exit tile ends the episode with +10.  Plain Python over numpy views."""

  def __init__(self, curtain, character, walls='#', coins='o', exit_char='E'):
    super(IceSkater, self).__init__(curtain, character)
    self.walls, self.coins, self.exit_char = walls, coins, exit_char

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    a = _action_id(actions)
    wall = all_things[self.walls].curtain.numpy()
    coin = all_things[self.coins].curtain.numpy()
    (r,), (c,) = np.nonzero(self.curtain.numpy())
    reward, (dr, dc) = -0.125, _DELTA[a]
    while (dr or dc) and not wall[r + dr, c + dc]:
      r, c = r + dr, c + dc
      reward += 0.5 * float(coin[r, c])
    self.curtain.zero_()
    self.curtain[r, c] = 1
    if all_things[self.exit_char].curtain[r, c]:
      reward += 10
      the_plot.terminate_episode()
    the_plot.add_reward(reward)


def ice_rink(**where):        # where: batch=, device= (left out: the engine's default)
  return ascii_art_to_game(
      ICE_ART, what_lies_beneath=' ',
      drapes={'A': IceSkater, '#': things.FixedDrape, 'o': things.FixedDrape,
              'E': things.FixedDrape},
      z_order='oEA#', update_schedule='A#oE', **where)


def make_game():
  """Zero arguments, started - the shape of the reference's make_game()
  (examples/boat_race.py:93-115); batched only through `engine.set_default_batch`."""
  game = ice_rink()
  board, reward, discount = game.its_showtime()
  return game, board, reward, discount


# ------------------------------------------------- two movers: a walker and its mirror

MIRROR_ART = ['#######',
              '#A   +#',
              '# # ###',
              '#  +  #',
              '# #   #',
              '#+   G#',
              '#######']


class Walker(things.Drape):
  """One cell per frame, stopped by walls (no reward of its own)."""

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    dr, dc = _DELTA[_action_id(actions)]
    (r,), (c,) = np.nonzero(self.curtain.numpy())
    if not all_things['#'].curtain[r + dr, c + dc]:
      self.curtain.zero_()
      self.curtain[r + dr, c + dc] = 1


class MirrorGhost(things.Sprite):
  """A SPRITE that steps the opposite way (walls stop it), updated after the walker.
  Meeting the walker ends the episode with -5; otherwise the frame pays -0.25, plus 1.5 when
  the walker stands on a '+' tile and 0.75 when the ghost does."""

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    dr, dc = _DELTA[_action_id(actions)]
    r, c = self.position.row - dr, self.position.col - dc
    if not all_things['#'].curtain[r, c]:
      self._position = self.Position(r, c)
    walker = all_things['A'].curtain
    plus = all_things['+'].curtain
    if walker[self.position.row, self.position.col]:
      the_plot.add_reward(-5.0)
      the_plot.terminate_episode()
      return
    reward = -0.25 + 1.5 * float((walker * plus).sum())
    reward += 0.75 * float(plus[self.position.row, self.position.col])
    the_plot.add_reward(reward)


def mirror(**where):
  """The ghost is painted in FRONT of the walker: when they meet, the walker is hidden."""
  return ascii_art_to_game(
      MIRROR_ART, what_lies_beneath=' ',
      sprites={'G': MirrorGhost},
      drapes={'A': Walker, '#': things.FixedDrape, '+': things.FixedDrape},
      z_order='+#AG', update_schedule='A#+G', **where)


# ------------------------------------------ one mover, discounts other than the default

TOLL_ART = ['#######',
            '#A $  #',
            '# #%# #',
            '#  $ E#',
            '#######']


class TollWalker(things.Drape):
  """One cell per frame (walls stop it), -1 per frame.  Standing on a '$' tile after the
  move the frame's discount is 0.5, on the '%' tile 0.25 (`Plot.change_default_discount`,
  campx/plot.py:232-257: it lasts one frame); reaching 'E' pays +5 and ends the episode with
  discount 0.75 (`terminate_episode(0.75)`, plot.py:161-184), not the default 0."""

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    dr, dc = _DELTA[_action_id(actions)]
    (r,), (c,) = np.nonzero(self.curtain.numpy())
    if not all_things['#'].curtain[r + dr, c + dc]:
      r, c = r + dr, c + dc
      self.curtain.zero_()
      self.curtain[r, c] = 1
    reward = -1.0
    if all_things['$'].curtain[r, c]:
      the_plot.change_default_discount(0.5)
    if all_things['%'].curtain[r, c]:
      the_plot.change_default_discount(0.25)
    if all_things['E'].curtain[r, c]:
      reward += 5.0
      the_plot.terminate_episode(0.75)
    the_plot.add_reward(reward)


def toll_road(**where):
  return ascii_art_to_game(
      TOLL_ART, what_lies_beneath=' ',
      drapes={'A': TollWalker, '#': things.FixedDrape, '$': things.FixedDrape,
              '%': things.FixedDrape, 'E': things.FixedDrape},
      z_order='$%EA#', update_schedule='A#$%E', **where)


# --------------------------------------------- three movers: a walker, a lift and a tram

TRIO_ART = ['#######',
            '#A  T #',
            '#  L  #',
            '#    $#',
            '#######']


class Lift(things.Drape):
  """Goes up on 'up' and down on 'down' in its own column, whatever the walker does."""

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    a = _action_id(actions)
    (r,), (c,) = np.nonzero(self.curtain.numpy())
    dr = -1 if a == 2 else (1 if a == 3 else 0)
    if dr and not all_things['#'].curtain[r + dr, c]:
      self.curtain.zero_()
      self.curtain[r + dr, c] = 1


class Tram(things.Sprite):
  """A sprite in the top row: left on 'left', right on 'right'.  The frame pays -0.5, +2 when
  the walker stands on '$', +1 when the walker shares a cell with the lift and +4 when all
  three are in one column; the walker riding the tram's cell ends the episode."""

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    a = _action_id(actions)
    dc = -1 if a == 0 else (1 if a == 1 else 0)
    r, c = self.position.row, self.position.col + dc
    if not all_things['#'].curtain[r, c]:
      self._position = self.Position(r, c)
    walker, lift = all_things['A'].curtain.numpy(), all_things['L'].curtain.numpy()
    (wr,), (wc,) = np.nonzero(walker)
    (lr,), (lc,) = np.nonzero(lift)
    reward = -0.5 + 2.0 * float(all_things['$'].curtain[wr, wc])
    reward += 1.0 * float((wr, wc) == (lr, lc))
    reward += 4.0 * float(wc == lc == self.position.col)
    the_plot.add_reward(reward)
    if (wr, wc) == (self.position.row, self.position.col):
      the_plot.terminate_episode()


def trio(**where):
  """z-order: the lift in front of the walker, the tram in front of both."""
  return ascii_art_to_game(
      TRIO_ART, what_lies_beneath=' ',
      sprites={'T': Tram},
      drapes={'A': Walker, 'L': Lift, '#': things.FixedDrape, '$': things.FixedDrape},
      z_order='$#ALT', update_schedule='AL#$T', **where)


# ------------------------------- one mover that changes the z-order: a mole and its lawn

BURROW_ART = ['#########',
              '#A ===  #',
              '# d===u #',
              '#  === $#',
              '#########']


class Mole(things.Drape):
  """One cell per frame (walls stop it).  Stepping on the 'd' tile it digs in - the Plot is
  told to move it behind everything (`change_z_order('A', None)`, campx/plot.py:121-159) -
  and from then on the lawn '=' and the tiles hide it; stepping on 'u' it comes up again, in
  front of the lawn (`change_z_order('A', '=')`).  -0.25 per frame; +0.5 when the previous
  render did not show it (layers['A'] is the occluded layer, campx/rendering.py:204-209);
  +3 and the end of the episode on '$', but only above ground."""

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    dr, dc = _DELTA[_action_id(actions)]
    (r,), (c,) = np.nonzero(self.curtain.numpy())
    was_shown = bool(layers['A'].sum())
    if not all_things['#'].curtain[r + dr, c + dc]:
      r, c = r + dr, c + dc
      self.curtain.zero_()
      self.curtain[r, c] = 1
    reward = -0.25 + (0.0 if was_shown else 0.5)
    if all_things['d'].curtain[r, c]:
      the_plot.change_z_order('A', None)
    if all_things['u'].curtain[r, c]:
      the_plot.change_z_order('A', '=')
    if all_things['$'].curtain[r, c] and was_shown:
      reward += 3.0
      the_plot.terminate_episode()
    the_plot.add_reward(reward)


def burrow(**where):
  return ascii_art_to_game(
      BURROW_ART, what_lies_beneath=' ',
      drapes={'A': Mole, '#': things.FixedDrape, '=': things.FixedDrape,
              'd': things.FixedDrape, 'u': things.FixedDrape, '$': things.FixedDrape},
      z_order='du$=A#', update_schedule='A#=du$', **where)


# ------------------- things that leave the board: a key, the door it opens, a hidden gem

VAULT_ART = ['########',
             '#A k#$ #',
             '#   D  #',
             '########']


class Key(things.Drape):
  """Picked up when the walker stands on it: +1 and the curtain is EMPTY from then on."""

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    if (self.curtain * all_things['A'].curtain).sum():
      self.curtain.zero_()
      the_plot.add_reward(1.0)


class Door(things.Drape):
  """Blocks the walker (see VaultWalker) until the key is gone; then it is gone too."""

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    if self.curtain.sum() and not all_things['k'].curtain.sum():
      self.curtain.zero_()
      the_plot.add_reward(0.5)


class VaultWalker(things.Drape):
  """One cell per frame, stopped by walls and by the door while it stands; -0.25 per frame."""

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    dr, dc = _DELTA[_action_id(actions)]
    (r,), (c,) = np.nonzero(self.curtain.numpy())
    if not all_things['#'].curtain[r + dr, c + dc] and not all_things['D'].curtain[r + dr, c + dc]:
      self.curtain.zero_()
      self.curtain[r + dr, c + dc] = 1
    the_plot.add_reward(-0.25)


class Gem(things.Sprite):
  """A sprite that shows itself only while the door is open and the walker is not on it
  (`_visible`, campx/things.py:294-296); the walker reaching it ends the episode with +10."""

  def __init__(self, corner, position, character):
    super(Gem, self).__init__(corner, position, character)
    self._visible = False

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    on_it = bool(all_things['A'].curtain[self.position.row, self.position.col])
    self._visible = (not all_things['D'].curtain.sum()) and not on_it
    if on_it:
      the_plot.add_reward(10.0)
      the_plot.terminate_episode()


def vault(**where):
  """Update order: walker, key, door, gem - four things whose state changes (the gem only in
  whether it shows)."""
  return ascii_art_to_game(
      VAULT_ART, what_lies_beneath=' ',
      sprites={'$': Gem},
      drapes={'A': VaultWalker, 'k': Key, 'D': Door, '#': things.FixedDrape},
      z_order='k$DA#', update_schedule='AkD$#', **where)


GAMES = {'ice_rink': ice_rink, 'mirror': mirror, 'toll_road': toll_road, 'trio': trio,
         'burrow': burrow, 'vault': vault}


# ------------------------------------------------------------- games that must be refused

class Stepper(things.Drape):
  """Moves right one cell per acted frame, cyclically, and pays the number of frames played
  so far: state the curtains do not hold."""

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    self.curtain.set_(torch.roll(self.curtain, 1, 1))
    the_plot['n'] = the_plot.get('n', 0) + 1
    the_plot.add_reward(float(the_plot['n'] % 3))


class Grower(things.Drape):
  """Covers one more cell every frame."""

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    self.curtain.set_(((self.curtain + torch.roll(self.curtain, 1, 1)) > 0).to(torch.uint8))


class Discounter(things.Drape):
  """Seventeen different discounts: the tables' 4-bit discount code has room for fifteen."""

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    self.curtain.set_(torch.roll(self.curtain, 1, 1))
    (r,), (c,) = np.nonzero(self.curtain.numpy())
    the_plot.change_default_discount((1 + c + 4 * _action_id(actions)) / 32.0)   # 20 values


class Reorderer(things.Drape):
  """Swaps two overlapping STATIC drapes: the scenery itself changes."""

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    self.curtain.set_(torch.roll(self.curtain, 1, 1))
    the_plot.change_z_order('#', None)


def refused_scenery(**where):
  return ascii_art_to_game(['A   ', '  # '], what_lies_beneath=' ',
                           drapes={'A': Reorderer, '#': things.FixedDrape,
                                   'o': Partial(Patch, 1, 2)},
                           z_order='o#A', update_schedule='A#o', **where)


class Patch(things.Drape):
  """A static one-cell drape placed by its arguments (not in the art): here under a '#'."""

  def __init__(self, curtain, character, row, col):
    super(Patch, self).__init__(curtain, character)
    self.curtain[row, col] = 1

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    pass


def refused(cls, **where):
  return ascii_art_to_game(['A   ', '  # '], what_lies_beneath=' ',
                           drapes={'A': cls, '#': things.FixedDrape}, z_order='#A',
                           update_schedule='A#', **where)


# ----------------------------------- games whose state is NOT all in their curtains
# (PyColab idioms: a time limit on `the_plot.frame`, a counter kept in the Plot, a cooldown
# kept on the entity itself - campx/plot.py:29 "a dict for exactly that", :259-280)

CLOCK_ART = ['######',
             '#A  o#',
             '# #  #',
             '#o   #',
             '######']


class TimedWalker(Walker):
  """Pays -1 per frame, +3 on a coin tile; the episode ends - discount 0 - once the frame
  number reaches the limit (the reference's drivers cut episodes by hand,
  examples/reinforce.py:36; PyColab games do it like this)."""

  def __init__(self, curtain, character, limit=30):
    super(TimedWalker, self).__init__(curtain, character)
    self.limit = limit

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    super(TimedWalker, self).update(actions, board, layers, backdrop, all_things, the_plot)
    on_coin = float((self.curtain * all_things['o'].curtain).sum())
    the_plot.add_reward(-1.0 + 3.0 * on_coin)
    if the_plot.frame >= self.limit:
      the_plot.terminate_episode()


def time_limit(limit=30, **where):
  return ascii_art_to_game(
      CLOCK_ART, what_lies_beneath=' ',
      drapes={'A': Partial(TimedWalker, limit=limit), '#': things.FixedDrape,
              'o': things.FixedDrape},
      z_order='oA#', update_schedule='A#o', **where)


class CoinCounter(Walker):
  """Every ARRIVAL on a coin tile (the coins stay: the count is the only memory) bumps
  `the_plot['n']`; the n-th pays n, and the `quota`-th ends the episode with discount 0.5."""

  def __init__(self, curtain, character, quota=4):
    super(CoinCounter, self).__init__(curtain, character)
    self.quota = quota

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    before = self.curtain.clone()
    super(CoinCounter, self).update(actions, board, layers, backdrop, all_things, the_plot)
    moved = bool((before != self.curtain).any())
    if moved and float((self.curtain * all_things['o'].curtain).sum()):
      the_plot['n'] = the_plot.get('n', 0) + 1
      the_plot.add_reward(float(the_plot['n']))
      if the_plot['n'] >= self.quota:
        the_plot.terminate_episode(0.5)
    else:
      the_plot.add_reward(-0.25)


def coin_counter(quota=4, **where):
  return ascii_art_to_game(
      CLOCK_ART, what_lies_beneath=' ',
      drapes={'A': Partial(CoinCounter, quota=quota), '#': things.FixedDrape,
              'o': things.FixedDrape},
      z_order='oA#', update_schedule='A#o', **where)


class Dasher(things.Drape):
  """Moves two cells when its cooldown (an attribute of the drape, not a curtain) is zero,
  one otherwise; a dash costs 1 and sets the cooldown to 3 frames."""

  def __init__(self, curtain, character):
    super(Dasher, self).__init__(curtain, character)
    self.cooldown = 0

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    dr, dc = _DELTA[_action_id(actions)]
    (r,), (c,) = np.nonzero(self.curtain.numpy())
    wall = all_things['#'].curtain
    steps = 2 if (self.cooldown == 0 and (dr or dc)) else 1
    moved = 0
    for _ in range(steps):
      if not wall[r + dr, c + dc]:
        r, c, moved = r + dr, c + dc, moved + 1
    self.curtain.zero_()
    self.curtain[r, c] = 1
    if moved == 2:
      self.cooldown = 3
      the_plot.add_reward(-1.0)
    else:
      self.cooldown = max(0, self.cooldown - 1)
      the_plot.add_reward(0.5 * float(all_things['o'].curtain[r, c]))
    if all_things['o'].curtain[r, c] and self.cooldown == 0 and (r, c) == (3, 1):
      the_plot.terminate_episode()


def dasher(**where):
  return ascii_art_to_game(
      CLOCK_ART, what_lies_beneath=' ',
      drapes={'A': Dasher, '#': things.FixedDrape, 'o': things.FixedDrape},
      z_order='oA#', update_schedule='A#o', **where)


HIDDEN_STATE_GAMES = {'time_limit': time_limit, 'coin_counter': coin_counter, 'dasher': dasher}


class Gambler(Walker):
  """State where the tabulator does not look: a module-level random number generator."""

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    super(Gambler, self).update(actions, board, layers, backdrop, all_things, the_plot)
    the_plot.add_reward(float(_GAMBLER_RNG.randint(0, 1000)))


_GAMBLER_RNG = np.random.RandomState(3)


class Unreadable(Walker):
  """Keeps a generator object on the entity: nothing the tabulator can compare."""

  def __init__(self, curtain, character):
    super(Unreadable, self).__init__(curtain, character)
    self.ticks = iter(range(10 ** 9))

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    super(Unreadable, self).update(actions, board, layers, backdrop, all_things, the_plot)
    the_plot.add_reward(float(next(self.ticks) % 2))


# ---------------------------------------------- drapes of SEVERAL cells that come and go (round 6)
#
# PyColab's staple: collectibles.  campx/things.py:161-262 sets no one-cell limit on a Drape, so a
# field of coins is ONE drape whose cells leave the curtain one by one.  The tabulator tracks such
# a drape as one thing per cell it ever covers (campx_amd/tabulate.py `piece_cell`).

class Forager(Walker):
  """A walker that is paid -0.125 per frame and ends the episode with +5 on the exit tile 'E'
  (where the game has one)."""

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    super(Forager, self).update(actions, board, layers, backdrop, all_things, the_plot)
    the_plot.add_reward(-0.125)
    if 'E' in all_things and (all_things['E'].curtain * self.curtain).sum():
      the_plot.add_reward(5.0)
      the_plot.terminate_episode()


class Coins(things.Drape):
  """Every coin the walker stands on is taken - its cell leaves the curtain - for +1 each."""

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    taken = self.curtain * all_things['A'].curtain
    if int(taken.sum()):
      self.curtain.set_(self.curtain - taken)
      the_plot.add_reward(float(taken.sum()))


class ReturningCoins(Coins):
  """... and when the last one has been taken they all come back (and pay 2 for it)."""

  def __init__(self, curtain, character):
    super(ReturningCoins, self).__init__(curtain, character)
    self.all_of_them = curtain.clone()

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    super(ReturningCoins, self).update(actions, board, layers, backdrop, all_things, the_plot)
    if not int(self.curtain.sum()):
      self.curtain.set_(self.all_of_them.clone())
      the_plot.add_reward(2.0)


class ThinIce(things.Drape):
  """Tiles that break when the walker steps OFF them: a cell leaves the curtain on the frame the
  walker, who stood on it at the last repaint (`layers['A']`), stands elsewhere; -0.5 each.
  Drawn in front of the walker, so a walker on ice does not show."""

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    if actions is None:
      return
    stood = layers['A'] if int(layers['A'].sum()) else the_plot.get('under_ice')
    left = self.curtain * (1 - all_things['A'].curtain)
    if stood is not None:
      broke = left * stood
      if int(broke.sum()):
        self.curtain.set_(self.curtain - broke)
        the_plot.add_reward(-0.5)
    # (a walker under the ice has no layer: remember where it is for the next frame)
    the_plot['under_ice'] = all_things['A'].curtain.clone()


# ---------------------------------------------- a Backdrop that changes (round 6)
#
# campx/things.py:103-148: `Backdrop.update(actions, board, layers, things, the_plot)` runs first
# in every frame (campx/engine.py:190) and may repaint the scenery.  The tabulator tracks every
# (cell, character) such a backdrop ever shows beyond its first picture as a piece painted behind
# every thing (campx_amd/tabulate.py `in_backdrop`).

class Lamps(things.Backdrop):
  """Floor lamps: a cell showing ':' (off) or '*' (on) flips each time the walker has just
  stepped onto it - seen at the START of the next frame, where `things['A']` still stands where
  the last frame put it.  A lit room pays: +0.25 per lamp that is on, every frame."""

  def update(self, actions, board, layers, things_, the_plot):
    if actions is None:
      return
    here = things_['A'].curtain
    before = the_plot.get('lamp_walker')
    the_plot['lamp_walker'] = here.clone()
    off, on = self.curtain == ord(':'), self.curtain == ord('*')
    lit = int(on.sum())
    if before is not None and not bool((before == here).all()):
      (r,), (c,) = np.nonzero(here.numpy())
      if bool(off[r, c]):
        self.curtain[r, c] = ord('*')
        lit += 1
      elif bool(on[r, c]):
        self.curtain[r, c] = ord(':')
        lit -= 1
    if lit:
      the_plot.add_reward(0.25 * lit)


class Tide(things.Backdrop):
  """A backdrop that changes ALL OVER: each time the walker has just stepped onto a switch tile
  ('s', a FixedDrape) the whole floor turns - every ' ' becomes the next character of `cycle` and so
  on round (two characters: day and night; three: seasons).  Dozens of cells, two or three
  pictures.  Every frame on a floor that is not the first picture pays 0.5."""

  cycle = ' .'

  def update(self, actions, board, layers, things_, the_plot):
    if actions is None:
      return
    here = things_['A'].curtain
    before = the_plot.get('tide_walker')
    the_plot['tide_walker'] = here.clone()
    if before is not None and not bool((before == here).all()) and int((here * things_['s'].curtain).sum()):
      old = self.curtain.clone()
      for i, ch in enumerate(self.cycle):
        self.curtain[old == ord(ch)] = ord(self.cycle[(i + 1) % len(self.cycle)])
    if int((self.curtain == ord(self.cycle[0])).sum()) < int((self.curtain == ord(self.cycle[1])).sum()):
      the_plot.add_reward(0.5)


class Seasons(Tide):
  cycle = ' .:'
