#!/usr/bin/env python3
"""tests/golden/random_tracks.npz: the seeded family of tests/random_tracks.py, every game built
from the REFERENCE's own classes (examples/boat_race.py AgentDrape, DirectionalHoverRewardDrape;
campx.things.FixedDrape) and run on the reference's engine / renderer / Plot, imported from
/root/reference where they lie (ref_harness).  Build container only:

    python tests/golden/make_random_golden.py [tracks] [hellos] [warehouses] [coins] [quests] [pickups]

(`hellos`: tests/golden/random_hellos.npz, the family of tests/random_hellos.py from the Hello World
notebook's own RollingDrape / SlidingSprite - see hellos() below.)

Per game k the arrays of make_golden.py's layout under `k<k>_<name>` (board, layered, reward,
discount, done, actions, chars), plus the definition itself - `k<k>_art` [H, W] uint8, `k<k>_meta`
a JSON string (tiles, reward vectors, z-order, update groups, blocking characters) - so that the
test notices a generator that no longer produces the games these frames belong to.  Each game is
also run with this repo's rule-library classes bound to the reference's `things` on the reference
engine and must give the same frames before anything is written.
"""

import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (loads the reference through ref_harness; repo appended after it)

sys.path.append(os.path.dirname(HERE))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import random_tracks  # noqa: E402

T, N = 48, 6


def main():
  torch.set_num_threads(1)
  ref = mg.ref
  out, rewards, blocked_by_tile, in_front = {}, 0, 0, 0
  for k, d in enumerate(random_tracks.definitions()):
    acts = mg.random_actions(5000 + k, T, N)
    acts[:, 0] = np.random.RandomState(6000 + k).choice(4, size=T)          # one that never stays

    def reference_game():
      return random_tracks.build(d, mg.to_game, mg.Partial, ref.boat_race.AgentDrape,
                                 ref.boat_race.DirectionalHoverRewardDrape, ref.things.FixedDrape,
                                 torch.FloatTensor)

    def library_game():
      return random_tracks.build(d, mg.to_game, mg.Partial, mg.R.AgentDrape,
                                 mg.R.DirectionalHoverRewardDrape, mg.R.FixedDrape, torch.FloatTensor)

    golden = mg.run(reference_game, acts)
    mg.assert_same(golden, mg.run(library_game, acts), 'random track {}: library rules'.format(k))
    for name, value in golden.items():
      out['k{}_{}'.format(k, name)] = value
    out['k{}_art'.format(k)] = np.array([[ord(c) for c in row] for row in d['art']], np.uint8)
    out['k{}_meta'.format(k)] = np.array(json.dumps(
        dict(tiles=d['tiles'], dctns=d['dctns'], z_order=d['z_order'], schedule=d['schedule'],
             blocking=d['blocking']), sort_keys=True))
    base = -0.25 * len(d['tiles'])
    rewards += int((golden['reward'] != np.float32(base)).sum())
    blocked_by_tile += int(len(d['blocking']) > 1)
    in_front += int(any(d['z_order'].index(ch) > d['z_order'].index('A') for ch in d['tiles']))
    print('track {:2d} {}x{} tiles {!r:6} z {!r:8} groups {} blocking {!r:4} return[mean] {:.2f}'.format(
        k, len(d['art']), len(d['art'][0]), d['tiles'], d['z_order'], len(d['schedule']), d['blocking'],
        float(golden['reward'].sum(0).mean())))
  assert rewards > 200 and blocked_by_tile >= 4 and in_front >= 8, (rewards, blocked_by_tile, in_front)
  path = os.path.join(HERE, 'random_tracks.npz')
  np.savez_compressed(path, **out)
  print('{} games, {} frames with a bonus, {} with a blocking tile, {} with a tile in front of the '
        'agent -> {} KiB; reference at {}'.format(random_tracks.N_GAMES + random_tracks.N_BIG, rewards, blocked_by_tile, in_front,
                                                   os.path.getsize(path) // 1024, ref.campx.__file__))


def hellos():
  """tests/golden/random_hellos.npz: tests/random_hellos.py's family from the notebook's own
  classes (Hello World Example.ipynb cell 3), integer actions 0..4 (4: quit)."""
  import ref_harness
  import random_hellos
  ns = ref_harness.notebook_namespace('Hello World Example.ipynb', [3])
  T, N = 40, 4
  out, trails, quits = {}, 0, 0
  for k, d in enumerate(random_hellos.definitions()):
    acts = mg.random_actions(7000 + k, T, N, n_actions=4)
    acts[:, 1] = mg.random_actions(7100 + k, T, 1, n_actions=5)[:, 0]          # with quits
    acts[:, 2] = int(k % 4)                                                    # a long straight trail

    def notebook_game():
      return random_hellos.build(d, mg.to_game, mg.Partial, ns['RollingDrape'], ns['SlidingSprite'])

    def library_game():
      return random_hellos.build(d, mg.to_game, mg.Partial, mg.R.RollingDrape, mg.R.SlidingSprite)

    golden = mg.run(notebook_game, acts, to_action=int)
    mg.assert_same(golden, mg.run(library_game, acts, to_action=int), 'random hello {}: library rules'.format(k))
    for name, value in golden.items():
      out['k{}_{}'.format(k, name)] = value
    out['k{}_art'.format(k)] = np.array([[ord(c) for c in row] for row in d['art']], np.uint8)
    out['k{}_meta'.format(k)] = np.array(json.dumps(
        dict(drapes=d['drapes'], sprites=d['sprites'], z_order=d['z_order'], schedule=d['schedule']),
        sort_keys=True))
    first_drape = min(d['z_order'].index(ch) for ch in d['drapes'])
    trails += int(first_drape > 0)
    quits += int(golden['done'].sum())
    print('hello {:2d} {}x{} drapes {!r:4} sprites {} z {!r:9} return[mean] {:.1f} done {}'.format(
        k, len(d['art']), len(d['art'][0]), d['drapes'], len(d['sprites']), d['z_order'],
        float(np.nansum(golden['reward'], 0).mean()), int(golden['done'].sum())))
  assert trails >= 8 and quits >= 40, (trails, quits)
  path = os.path.join(HERE, 'random_hellos.npz')
  np.savez_compressed(path, **out)
  print('{} games, {} with trails, {} episode ends -> {} KiB'.format(
      random_hellos.N_GAMES, trails, quits, os.path.getsize(path) // 1024))


def warehouses():
  """tests/golden/random_warehouses.npz: tests/random_warehouses.py's sokoban levels - the
  reference's AgentDrape + this repo's Box / Goal rules bound to the reference's `things`, on
  the reference's engine."""
  import random_warehouses
  ref = mg.ref
  T, N = 60, 8
  out, pushes, ends = {}, 0, 0
  for k, d in enumerate(random_warehouses.definitions()):
    acts = mg.random_actions(8000 + k, T, N)
    acts[:, 0] = np.random.RandomState(8100 + k).choice(4, size=T)            # one that never stays
    runs = np.random.RandomState(8200 + k)
    for n in (1, 2, 3):                                                        # three that keep going: pushes
      t = 0
      while t < T:
        length = int(runs.randint(2, 6))
        acts[t:t + length, n] = int(runs.randint(4))
        t += length

    def game(agent_cls, fixed_cls):
      return lambda: random_warehouses.build(d, mg.to_game, mg.Partial, agent_cls, mg.R.BoxDrape,
                                             mg.R.GoalDrape, fixed_cls)

    golden = mg.run(game(ref.boat_race.AgentDrape, ref.things.FixedDrape), acts)
    mg.assert_same(golden, mg.run(game(mg.R.AgentDrape, mg.R.FixedDrape), acts),
                   'random warehouse {}: library agent'.format(k))
    for name, value in golden.items():
      out['k{}_{}'.format(k, name)] = value
    out['k{}_art'.format(k)] = np.array([[ord(c) for c in row] for row in d['art']], np.uint8)
    out['k{}_meta'.format(k)] = np.array(json.dumps(
        dict(boxes=d['boxes'], z_order=d['z_order'], schedule=d['schedule']), sort_keys=True))
    moved = 0
    for ch in d['boxes']:
      where = (golden['board'] == ord(ch)).reshape(T + 1, N, -1).argmax(-1)
      moved += int((where[1:] != where[:-1]).sum())
    pushes += moved
    ends += int(golden['done'].sum())
    print('warehouse {:2d} {}x{} boxes {!r:4} z {!r:7} groups {} pushes {} done {}'.format(
        k, len(d['art']), len(d['art'][0]), d['boxes'], d['z_order'], d['schedule'], moved,
        int(golden['done'].sum())))
  assert pushes > 150 and ends >= 5, (pushes, ends)
  path = os.path.join(HERE, 'random_warehouses.npz')
  np.savez_compressed(path, **out)
  print('{} levels, {} box moves, {} episode ends -> {} KiB'.format(
      random_warehouses.N_GAMES, pushes, ends, os.path.getsize(path) // 1024))


def coins():
  """tests/golden/random_coins.npz: tests/random_coins.py's coin fields from Demo 3's own AgentDrape
  (notebook cell 3), one-hot LISTS as actions, as the notebook's cells pass them."""
  import ref_harness
  import random_coins
  ref = mg.ref
  ns = ref_harness.notebook_namespace('Demo 3: Hover Reward Example.ipynb', [3])
  T, N = 48, 6
  out, paid, in_front = {}, 0, 0
  for k, d in enumerate(random_coins.definitions()):
    acts = mg.random_actions(9000 + k, T, N)
    acts[:, 0] = np.random.RandomState(9100 + k).choice(4, size=T)

    def notebook_game():
      return random_coins.build(d, mg.to_game, mg.Partial(ns['AgentDrape'], blocking_chars=d['blocking'],
                                                          reward_chars=d['rewarding']), ref.things.FixedDrape)

    def library_game():
      return random_coins.build(d, mg.to_game, mg.Partial(mg.R.AgentDrape, blocking_chars=d['blocking'],
                                                          step_reward=0, reward_chars=d['rewarding']),
                                mg.R.FixedDrape)

    golden = mg.run(notebook_game, acts, to_action=mg.one_hot_list)
    mg.assert_same(golden, mg.run(library_game, acts, to_action=mg.one_hot_list),
                   'random coins {}: library rules'.format(k))
    for name, value in golden.items():
      out['k{}_{}'.format(k, name)] = value
    out['k{}_art'.format(k)] = np.array([[ord(c) for c in row] for row in d['art']], np.uint8)
    out['k{}_meta'.format(k)] = np.array(json.dumps(
        dict(tiles=d['tiles'], z_order=d['z_order'], schedule=d['schedule'], blocking=d['blocking'],
             rewarding=d['rewarding']), sort_keys=True))
    paid += int((golden['reward'] > 0).sum())
    in_front += int(any(d['z_order'].index(ch) > d['z_order'].index('A') for ch in d['tiles']))
    print('coins {:2d} {}x{} tiles {!r:4} z {!r:6} schedule {!r:6} blocking {!r:4} rewarding {!r:4} paid {}'.format(
        k, len(d['art']), len(d['art'][0]), d['tiles'], d['z_order'], d['schedule'], d['blocking'], d['rewarding'],
        int((golden['reward'] > 0).sum())))
  assert paid > 150 and in_front >= 4, (paid, in_front)
  path = os.path.join(HERE, 'random_coins.npz')
  np.savez_compressed(path, **out)
  print('{} games, {} paid frames, {} with a tile in front of the agent -> {} KiB'.format(
      random_coins.N_GAMES, paid, in_front, os.path.getsize(path) // 1024))


def quests():
  """tests/golden/random_quests.npz: tests/random_quests.py's games of arbitrary Python classes
  (tests/traced_games.py, which in this process imports the REFERENCE's campx) on the reference's
  engine; an environment whose episode ended gets a fresh game before its next action."""
  import random_quests
  import traced_games
  assert traced_games.things is mg.ref.things
  T, N = 60, 6
  out, ends, discounts = {}, 0, set()
  for k, d in enumerate(random_quests.definitions()):
    acts = mg.random_actions(9500 + k, T, N)
    runs = np.random.RandomState(9600 + k)
    for n in (1, 2):                                   # two that keep going
      t = 0
      while t < T:
        length = int(runs.randint(2, 6))
        acts[t:t + length, n] = int(runs.randint(4))
        t += length
    golden = mg.run(random_quests.builder(d), acts)
    for name, value in golden.items():
      out['k{}_{}'.format(k, name)] = value
    out['k{}_art'.format(k)] = np.array([[ord(c) for c in row] for row in d['art']], np.uint8)
    out['k{}_meta'.format(k)] = np.array(json.dumps(dict(kind=d['kind']), sort_keys=True))
    ends += int(golden['done'].sum())
    discounts |= set(np.unique(golden['discount']).tolist())
    print('quest {:2d} {:6s} {}x{} return[mean] {:.2f} done {} discounts {}'.format(
        k, d['kind'], len(d['art']), len(d['art'][0]), float(np.nansum(golden['reward'], 0).mean()),
        int(golden['done'].sum()), sorted(set(np.unique(golden['discount']).tolist()))))
  assert ends >= 10 and discounts >= {0.0, 0.25, 0.5, 0.75, 1.0}, (ends, discounts)
  path = os.path.join(HERE, 'random_quests.npz')
  np.savez_compressed(path, **out)
  print('{} games, {} episode ends -> {} KiB'.format(random_quests.N_GAMES, ends, os.path.getsize(path) // 1024))


def pickups():
  """tests/golden/random_pickups.npz: tests/random_pickups.py's games - drapes of several cells
  that come and go and a Backdrop that changes (tests/traced_games.py Coins / ReturningCoins /
  ThinIce / Lamps / Tide / Seasons, which in this process
  imports the REFERENCE's campx) - on the reference's engine; an environment whose episode ended
  gets a fresh game before its next action."""
  import random_pickups
  import traced_games
  assert traced_games.things is mg.ref.things
  T, N = 120, 6
  out, ends, taken, back, broke, flips, turns = {}, 0, 0, 0, 0, 0, 0
  for k, d in enumerate(random_pickups.definitions()):
    acts = mg.random_actions(9700 + k, T, N)
    runs = np.random.RandomState(9800 + k)
    for n in (1, 2, 3, 4):                             # four that keep going: they get around
      t = 0
      while t < T:
        length = int(runs.randint(2, 7))
        acts[t:t + length, n] = int(runs.randint(4))
        t += length
    golden = mg.run(random_pickups.builder(d), acts)
    for name, value in golden.items():
      out['k{}_{}'.format(k, name)] = value
    out['k{}_art'.format(k)] = np.array([[ord(c) for c in row] for row in d['art']], np.uint8)
    out['k{}_meta'.format(k)] = np.array(json.dumps(dict(kind=d['kind']), sort_keys=True))
    ends += int(golden['done'].sum())
    ch = {'ice': ord('~'), 'lamps': ord('*'), 'tide': ord('.'), 'seasons': ord('.')}.get(d['kind'], ord('o'))
    cells = (golden['board'] == ch).sum(axis=(2, 3)).astype(np.int64)        # [T + 1, N]
    if d['kind'] == 'lamps':
      flips += int((np.diff(cells, axis=0) != 0).sum())
    if d['kind'] in ('tide', 'seasons'):               # (a whole floor turns; an episode's end turns it back)
      turns += int((np.abs(np.diff(cells, axis=0)) > 3).sum())
    gone = int((np.diff(cells, axis=0) < 0).sum())
    taken += gone if d['kind'] in ('coins', 'returning') else 0
    broke += gone if d['kind'] == 'ice' else 0
    back += int((np.diff(cells, axis=0) > 0).sum()) if d['kind'] == 'returning' else 0
    print('pickup {:2d} {:9s} {}x{} return[mean] {:.2f} done {} cells left at the end {}'.format(
        k, d['kind'], len(d['art']), len(d['art'][0]), float(np.nansum(golden['reward'], 0).mean()),
        int(golden['done'].sum()), cells[-1].tolist()))
  assert ends >= 4 and taken >= 40 and back >= 4 and broke >= 15 and flips >= 20 and turns >= 15, (
      ends, taken, back, broke, flips, turns)
  path = os.path.join(HERE, 'random_pickups.npz')
  np.savez_compressed(path, **out)
  print('{} games, {} episode ends, {} coins taken, {} times they came back, {} tiles of ice broke, '
        '{} lamps flipped, {} times a whole floor turned -> {} KiB'.format(
            random_pickups.N_GAMES, ends, taken, back, broke, flips, turns, os.path.getsize(path) // 1024))


if __name__ == '__main__':
  which = sys.argv[1:] or ['tracks', 'hellos', 'warehouses', 'coins', 'quests', 'pickups']
  if 'tracks' in which:
    main()
  if 'hellos' in which:
    hellos()
  if 'warehouses' in which:
    warehouses()
  if 'coins' in which:
    coins()
  if 'quests' in which:
    quests()
  if 'pickups' in which:
    pickups()
