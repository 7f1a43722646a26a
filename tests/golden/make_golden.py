#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by EXECUTING the reference.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

For every game the reference's own `Engine` / renderer / `Plot` (imported from
/root/reference through `ref_harness`) is stepped with committed random action
streams, one run per environment (the reference has no batch axis), and the
per-step flat board, per-character layers, reward, discount and game-over flag
are stored as `<game>.npz`.  Recorded notebook outputs that still agree with the
reference's current code are stored in `notebook_kats.json`.

Game sources:
  boat_race   the reference's examples/boat_race.py, unchanged.
  demo1..4    notebook cells exec'd from examples/Demo N*.ipynb, unchanged.
  hello_world notebook cell 3/4 of examples/Hello World Example.ipynb.
  wall_world  Demo 2's AgentDrape (exec'd from the notebook) + things.FixedDrape
              on the build-authored 10x10 board.
  sokoban     reference AgentDrape (boat_race.py) + the build's Box/Goal rules
              bound to the reference's `things`.
  big_rows    tests/big_rows_game.py: sokoban's rules on a 10x12 board with 15 characters
              (rows of 1800 bytes), reference AgentDrape + the build's Box/Goal rules.
  maze_RxC    campx_amd/games/maze.py: boards above 128 cells (the wide tier), rule classes
              bound to the reference's `things` on the reference engine.
  shape_zoo*  tests/shape_zoo.py: the build's RollingDrape / SlidingSprite (pinned to the
              notebook's classes by hello_world) in other arrangements, bound to the
              reference's `things`, on the reference's engine / renderer / Plot.

Each game is ALSO run with the build's library rule classes
(`campx_amd.rules.bind(<reference things>)`) on the reference engine and the two
trajectories are asserted identical before anything is written: that pins the
rule library's Python `update()` bodies to the reference example classes.

Array layout in every npz (T steps, N environments, L characters ascending):
  chars    [L]            uint8 character codes, ascending
  actions  [T, N]         int8 action ids (0..4 = left,right,up,down,stay;
                          hello_world: the game's own 0..4)
  board    [T+1, N, H, W] int8; index 0 is the its_showtime() observation
  layered  [T+1, N, L, H, W] uint8, from `Observation.layers[ch]`
  reward   [T, N]         float32, NaN where the reference returned None
  discount [T, N]         float32
  done     [T, N]         uint8 game-over flag after the step.  After a game-over
                          step the next step is played on a fresh make_game().
  perf     [T, N]         int8, boat_race only: the reference's step_perf() of
                          layers['A'] before/after each play().
"""

import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)

import ref_harness  # noqa: E402

ref = ref_harness.load()
sys.path.append(REPO)            # after the reference: `campx` stays the reference

import numpy as np  # noqa: E402
import torch  # noqa: E402

from campx_amd import rules as _rules  # noqa: E402
from campx_amd.games import (boat_race as g_br, wall_world as g_ww,  # noqa: E402
                             sokoban as g_sk, demos as g_demos)

R = _rules.bind(ref.things)      # library rules on the reference's base classes
Partial = ref.ascii_art.Partial
to_game = ref.ascii_art.ascii_art_to_game


def one_hot(a):
  v = torch.zeros(5)
  v[int(a)] = 1
  return v


def one_hot_list(a):
  """Demo 1-3 pass plain lists (`game.play([1,0,0,0,0])`, Demo 1 cell 6)."""
  return [int(i == int(a)) for i in range(5)]


def run(make_game, actions, to_action=one_hot, perf_fn=None):
  """Step N independent reference games; see module docstring for the layout.

  `perf_fn(pre, post)`, if given, is evaluated on `layers['A']` before and after
  every `play()` as the reference's driver does (examples/reinforce.py:138-156)
  and stored as `perf [T, N]` int8.
  """
  T, N = actions.shape
  out = None
  for n in range(N):
    game = make_game()
    obs, reward, discount = game.its_showtime()
    assert reward is None and discount == 1.0
    chars = sorted(obs.layers.keys())
    if out is None:
      H, W = obs.board.shape
      out = dict(
          chars=np.array([ord(c) for c in chars], dtype=np.uint8),
          actions=actions.astype(np.int8),
          board=np.zeros((T + 1, N, H, W), np.int8),
          layered=np.zeros((T + 1, N, len(chars), H, W), np.uint8),
          reward=np.zeros((T, N), np.float32),
          discount=np.zeros((T, N), np.float32),
          done=np.zeros((T, N), np.uint8))
      if perf_fn is not None:
        out['perf'] = np.zeros((T, N), np.int8)

    def record(i, obs):
      out['board'][i, n] = obs.board.numpy()
      for k, ch in enumerate(chars):
        out['layered'][i, n, k] = obs.layers[ch].numpy()
      # every plane of the reference's layered_board is one of the layers
      lb = obs.layered_board.numpy()
      assert lb.shape[0] == len(chars)
      for plane in lb:
        assert any((plane == out['layered'][i, n, k]).all()
                   for k in range(len(chars)))

    record(0, obs)
    for t in range(T):
      if game._game_over:
        game = make_game()
        obs, _, _ = game.its_showtime()
      if perf_fn is not None:
        pre = obs.layers['A'] + 0          # a copy, as reinforce.py:141 takes it
      obs, reward, discount = game.play(to_action(actions[t, n]))
      if perf_fn is not None:
        out['perf'][t, n] = int(perf_fn(pre, obs.layers['A']))
      record(t + 1, obs)
      out['reward'][t, n] = np.nan if reward is None else float(reward)
      out['discount'][t, n] = float(discount)
      out['done'][t, n] = int(game._game_over)
  return out


def assert_same(a, b, what):
  for key in a:
    x, y = a[key], b[key]
    same = np.array_equal(x, y, equal_nan=True) if x.dtype.kind == 'f' \
        else np.array_equal(x, y)
    assert same, '{}: {} differs'.format(what, key)


def save(name, data, **meta):
  path = os.path.join(HERE, name + '.npz')
  np.savez_compressed(path, **data)
  print('{:12s} T={:4d} N={:3d} chars={!r} return[mean]={:.3f} done={} -> {} KiB'
        .format(name, data['actions'].shape[0], data['actions'].shape[1],
                ''.join(chr(c) for c in data['chars']),
                float(np.nansum(data['reward'], 0).mean()),
                int(data['done'].sum()), os.path.getsize(path) // 1024))


def random_actions(seed, T, N, n_actions=5):
  return np.random.RandomState(seed).randint(0, n_actions, size=(T, N))


# --------------------------------------------------------------------- games

def gen_boat_race():
  acts = random_actions(101, 100, 64)
  acts[:20, 0] = [g_br.select_action_preset(t) for t in range(20)]
  acts[:, 1] = 4                                   # stays put for 100 frames
  acts[:, 2] = np.tile([1, 1, 3, 3, 0, 0, 2, 2], 13)[:100]   # clockwise laps

  def reference_game():
    # examples/boat_race.py:93-113 without the its_showtime() call
    b = ref.boat_race
    return to_game(
        b.GAME_ART, what_lies_beneath=' ',
        drapes={'A': b.AgentDrape, '#': ref.things.FixedDrape,
                '^': Partial(b.DirectionalHoverRewardDrape,
                             dctns=torch.FloatTensor([0, 0, b.CW_reward, b.CCW_reward, 0])),
                '>': Partial(b.DirectionalHoverRewardDrape,
                             dctns=torch.FloatTensor([b.CCW_reward, b.CW_reward, 0, 0, 0])),
                'v': Partial(b.DirectionalHoverRewardDrape,
                             dctns=torch.FloatTensor([0, 0, b.CCW_reward, b.CW_reward, 0])),
                '<': Partial(b.DirectionalHoverRewardDrape,
                             dctns=torch.FloatTensor([b.CW_reward, b.CCW_reward, 0, 0, 0]))},
        z_order='^>v<A#', update_schedule='A^>v<#')

  # hidden performance: the reference's own step_perf with the driver's views
  # a, b, c, d (examples/reinforce.py:242-258); its debugging prints are discarded.
  import contextlib
  import io
  a = torch.zeros(5, 5).long()
  a[1, 2] = 1
  a[3, 2] = 1
  b = torch.zeros(5, 5).long()
  b[1, 3] = 1
  b[3, 1] = 1
  c = a.t()
  d = torch.zeros(5, 5).long()
  d[1, 1] = 1
  d[3, 3] = 1

  def reference_perf(pre, post):
    with contextlib.redirect_stdout(io.StringIO()):
      return ref.boat_race.step_perf(a, b, c, d, pre.long(), post.long())

  # make_game() itself (calls its_showtime inside) must give the same first frame
  game, board, reward, discount = ref.boat_race.make_game()
  golden = run(reference_game, acts, perf_fn=reference_perf)
  # two clockwise laps are +1 per frame, the scripted lap-and-back sums to zero
  assert golden['perf'][:16, 2].tolist() == [1] * 16
  assert golden['perf'][:8, 0].tolist() == [1] * 8
  assert golden['perf'][8:16, 0].tolist() == [-1] * 8
  assert np.array_equal(golden['board'][0, 0], board.board.numpy())
  # scripted lap rewards
  assert golden['reward'][:20, 0].tolist() == [
      2, -1, 2, -1, 2, -1, 2, -1, 0, -1, 0, -1, 0, -1, 0, -1, -1, -1, -1, -1]

  def library_game():
    drapes = {'A': R.AgentDrape, '#': R.FixedDrape}
    for ch, d in g_br.ARROW_DCTNS.items():
      drapes[ch] = Partial(R.DirectionalHoverRewardDrape,
                           dctns=torch.FloatTensor(d), base_reward=-0.25)
    return to_game(g_br.GAME_ART, what_lies_beneath=' ', drapes=drapes,
                   z_order='^>v<A#', update_schedule='A^>v<#')

  from campx_amd.games.boat_race import step_perf as build_step_perf, performance_masks
  views = performance_masks()
  lib = run(library_game, acts,
            perf_fn=lambda pre, post: build_step_perf(*views, pre.long(), post.long()))
  assert_same(golden, lib, 'boat_race library rules')
  save('boat_race', golden)


def gen_demos():
  kats = {}
  # ---- Demo 1
  ns = ref_harness.notebook_namespace('Demo 1: Simple Agent Example.ipynb', [2, 3])
  acts = random_actions(201, 60, 16)
  acts[0, 0] = 0     # the notebook's recorded step: left
  golden = run(lambda: to_game(ns['GAME_ART'], what_lies_beneath=' ',
                               drapes={'A': ns['AgentDrape']}, z_order='A'), acts,
               to_action=one_hot_list)
  lib = run(lambda: to_game(g_demos.STAR_ART, what_lies_beneath=' ',
                            drapes={'A': Partial(R.AgentDrape, blocking_chars='',
                                                 step_reward=1)}, z_order='A'), acts,
            to_action=one_hot_list)
  assert_same(golden, lib, 'demo1 library rules')
  save('demo1', golden)
  kats['demo1_board_after_left'] = golden['board'][1, 0].tolist()

  # ---- Demo 2
  ns = ref_harness.notebook_namespace('Demo 2: Simple Wall Example.ipynb', [2, 3])
  demo2_agent = ns['AgentDrape']
  acts = random_actions(202, 60, 16)
  acts[:3, 0] = 1    # the notebook's recorded steps: right x3 (third is blocked)
  golden = run(lambda: to_game(ns['GAME_ART'], what_lies_beneath=' ',
                               drapes={'A': demo2_agent, '#': ref.things.FixedDrape},
                               z_order='A#'), acts, to_action=one_hot_list)
  lib = run(lambda: to_game(g_demos.STAR_ART, what_lies_beneath=' ',
                            drapes={'A': Partial(R.AgentDrape, blocking_chars='#',
                                                 step_reward=1),
                                    '#': R.FixedDrape}, z_order='A#'), acts, to_action=one_hot_list)
  assert_same(golden, lib, 'demo2 library rules')
  save('demo2', golden)
  kats['demo2_board_after_3_right'] = golden['board'][3, 0].tolist()

  # ---- Demo 3 (cell 1 needs real PySyft; cells 2-3 do not)
  ns = ref_harness.notebook_namespace('Demo 3: Hover Reward Example.ipynb', [2, 3])
  acts = random_actions(203, 60, 16)
  acts[:7, 0] = [1, 1, 0, 4, 0, 3, 3]
  golden = run(lambda: to_game(ns['GAME_ART'], what_lies_beneath=' ',
                               drapes={'A': ns['AgentDrape'],
                                       '#': ref.things.FixedDrape,
                                       '*': ref.things.FixedDrape},
                               z_order='*A#'), acts, to_action=one_hot_list)
  assert golden['reward'][:7, 0].tolist() == [1, 0, 1, 0, 0, 1, 0]
  lib = run(lambda: to_game(g_demos.STAR_ART, what_lies_beneath=' ',
                            drapes={'A': Partial(R.AgentDrape, blocking_chars='#',
                                                 step_reward=0, reward_chars='*'),
                                    '#': R.FixedDrape, '*': R.FixedDrape},
                            z_order='*A#'), acts, to_action=one_hot_list)
  assert_same(golden, lib, 'demo3 library rules')
  save('demo3', golden)

  # ---- Demo 4
  ns = ref_harness.notebook_namespace(
      'Demo 4: Directional Hover Reward Example.ipynb', [2, 3])
  acts = random_actions(204, 60, 16)
  acts[:4, 0] = [1, 3, 0, 2]               # the notebook's cells 6-9
  unit = {'^': [0, 0, 1, 0, 0], '>': [0, 1, 0, 0, 0],
          'v': [0, 0, 0, 1, 0], '<': [1, 0, 0, 0, 0]}

  def demo4_reference():
    drapes = {'A': ns['AgentDrape'], '#': ref.things.FixedDrape}
    for ch, d in unit.items():
      drapes[ch] = Partial(ns['DirectionalHoverRewardDrape'],
                           dctns=torch.FloatTensor(d))
    return to_game(ns['GAME_ART'], what_lies_beneath=' ', drapes=drapes,
                   z_order='^>v<A#', update_schedule='A^>v<#')

  def demo4_library():
    drapes = {'A': R.AgentDrape, '#': R.FixedDrape}
    for ch, d in unit.items():
      drapes[ch] = Partial(R.DirectionalHoverRewardDrape,
                           dctns=torch.FloatTensor(d), base_reward=0)
    return to_game(g_demos.ARROW_ART, what_lies_beneath=' ', drapes=drapes,
                   z_order='^>v<A#', update_schedule='A^>v<#')

  golden = run(demo4_reference, acts)
  assert golden['reward'][:4, 0].tolist() == [1, 0, 0, 0]   # recorded: 1.0 0.0 0.0 0.0
  assert_same(golden, run(demo4_library, acts), 'demo4 library rules')
  save('demo4', golden)

  # ---- Demo 5 (the boat-race training notebook): cell 1 defines Demo 4's game again, with classes
  # of its own; the rest is the RL driver.  Same frames, so tests hold it to demo4.npz.
  ns5 = ref_harness.notebook_namespace('Demo 5: Boat Race Example.ipynb', [1])
  assert ns5['GAME_ART'] == ns['GAME_ART'] and ns5['AgentDrape'] is not ns['AgentDrape']

  def demo5_reference():
    drapes = {'A': ns5['AgentDrape'], '#': ref.things.FixedDrape}
    for ch, d in unit.items():
      drapes[ch] = Partial(ns5['DirectionalHoverRewardDrape'], dctns=torch.FloatTensor(d))
    return to_game(ns5['GAME_ART'], what_lies_beneath=' ', drapes=drapes,
                   z_order='^>v<A#', update_schedule='A^>v<#')
  assert_same(golden, run(demo5_reference, acts), "Demo 5's classes give Demo 4's frames")

  # ---- recorded notebook outputs that the current code still reproduces
  demo1_recorded = [[35, 35, 35, 35, 35], [65, 32, 42, 32, 35], [35, 42, 35, 42, 35],
                    [35, 32, 42, 32, 35], [35, 35, 35, 35, 35]]       # Demo 1 cell 6
  demo2_recorded = [[35, 35, 35, 35, 35], [35, 32, 42, 65, 35], [35, 42, 35, 42, 35],
                    [35, 32, 42, 32, 35], [35, 35, 35, 35, 35]]       # Demo 2 cell 6
  assert kats['demo1_board_after_left'] == demo1_recorded
  assert kats['demo2_board_after_3_right'] == demo2_recorded
  with open(os.path.join(HERE, 'notebook_kats.json'), 'w') as f:
    json.dump(kats, f, indent=1, sort_keys=True)
  return demo2_agent


def gen_wall_world(demo2_agent):
  acts = random_actions(301, 200, 32)
  golden = run(lambda: to_game(
      g_ww.GAME_ART, what_lies_beneath=' ',
      drapes={'A': demo2_agent, '#': ref.things.FixedDrape,
              '*': ref.things.FixedDrape, 'o': ref.things.FixedDrape},
      z_order='*oA#', update_schedule='A*o#'), acts, to_action=one_hot_list)
  lib = run(lambda: to_game(
      g_ww.GAME_ART, what_lies_beneath=' ',
      drapes={'A': Partial(R.AgentDrape, blocking_chars='#', step_reward=1),
              '#': R.FixedDrape, '*': R.FixedDrape, 'o': R.FixedDrape},
      z_order='*oA#', update_schedule='A*o#'), acts, to_action=one_hot_list)
  assert_same(golden, lib, 'wall_world library rules')
  save('wall_world', golden)


def gen_sokoban():
  acts = random_actions(401, 100, 64)
  acts[:5, 0] = [3, 1, 1, 3, 3]          # push box down, walk to the goal
  acts[:2, 1] = [3, 3]                   # second push is blocked by the wall
  acts[:4, 2] = [0, 3, 1, 1]             # walk round and push the box right twice

  def game(agent_cls):
    return lambda: to_game(
        g_sk.GAME_ART, what_lies_beneath=' ',
        drapes={'#': ref.things.FixedDrape,
                'X': Partial(R.BoxDrape, agent_char='A', blocking_chars='#'),
                'A': Partial(agent_cls, blocking_chars='#X'),
                'G': Partial(R.GoalDrape, agent_char='A', step_reward=-1,
                             goal_reward=50)},
        update_schedule=[['X'], ['A', 'G', '#']], z_order='GXA#')

  golden = run(game(ref.boat_race.AgentDrape), acts)
  assert golden['reward'][:5, 0].tolist() == [-1, -1, -1, -1, 49]
  assert golden['done'][:5, 0].tolist() == [0, 0, 0, 0, 1]
  assert golden['discount'][4, 0] == 0.0
  assert_same(golden, run(game(R.AgentDrape), acts), 'sokoban library agent')
  save('sokoban', golden)


def gen_sokoban_levels():
  """Two and three boxes (K = 3, 4 moving things), reference AgentDrape + the
  build's Box/Goal rules on the reference engine."""
  for level in (1, 2, 3):
    art = g_sk.LEVELS[level]
    boxes = [ch for ch in 'XYZ' if any(ch in row for row in art)]
    if level == 3:
      # the 16x16 level (device-enumerated state table): pushes of both boxes, a walk to the goal
      acts = random_actions(413, 100, 12)
      acts[:5, 0] = [1, 1, 1, 1, 1]                         # push X right (against the wall)
      walk = [3] * 11 + [1] * 11                            # down the left side, along the bottom
      acts[:len(walk), 1] = walk
      acts[:40, 2] = ([3] * 3 + [1] * 9 + [3] * 2) * 2 + [1] * 12   # towards Y, push it
    else:
      acts = random_actions(410 + level, 80, 32)
      acts[:6, 0] = [3, 0, 3, 3, 1, 1]       # walk round a box and to the goal row

    def game(agent_cls, box_cls, goal_cls, fixed_cls):
      def make():
        drapes = {'#': fixed_cls,
                  'A': Partial(agent_cls, blocking_chars='#' + ''.join(boxes)),
                  'G': Partial(goal_cls, agent_char='A', step_reward=-1, goal_reward=50)}
        for ch in boxes:
          drapes[ch] = Partial(box_cls, agent_char='A',
                               blocking_chars='#' + ''.join(b for b in boxes if b != ch))
        return to_game(art, what_lies_beneath=' ', drapes=drapes,
                       update_schedule=[boxes, ['A', 'G', '#']],
                       z_order='G' + ''.join(boxes) + 'A#')
      return make

    golden = run(game(ref.boat_race.AgentDrape, R.BoxDrape, R.GoalDrape,
                      ref.things.FixedDrape), acts)
    lib = run(game(R.AgentDrape, R.BoxDrape, R.GoalDrape, R.FixedDrape), acts)
    assert_same(golden, lib, 'sokoban level {} library agent'.format(level))
    if level == 3:
      moved = [(golden['board'][:, n] == ord(ch)).reshape(golden['board'].shape[0], -1).argmax(1)
               for n in range(3) for ch in 'XY']
      assert sum(len(set(m.tolist())) > 1 for m in moved) >= 2, 'no box was pushed'
    save('sokoban_l{}'.format(level), golden)


def gen_hello_world():
  ns = ref_harness.notebook_namespace('Hello World Example.ipynb', [3, 4])
  acts = random_actions(501, 40, 4, n_actions=4)
  acts[25, 0] = 4                          # quit -> terminate_episode()
  acts[:, 1] = 0
  golden = run(lambda: ns['make_game'](), acts, to_action=int)
  assert golden['done'][25, 0] == 1 and golden['discount'][25, 0] == 0.0
  # the library's RollingDrape / SlidingSprite, on the reference's engine and things
  from campx_amd.games import hello_world as g_hw

  def lib_game():
    return to_game(
        g_hw.HELLO_ART, what_lies_beneath=' ',
        sprites={'1': Partial(R.SlidingSprite, 0), '2': Partial(R.SlidingSprite, 1),
                 '3': Partial(R.SlidingSprite, 2), '4': Partial(R.SlidingSprite, 3)},
        drapes={'@': R.RollingDrape}, z_order='12@34')
  lib = run(lib_game, acts, to_action=int)
  assert_same(golden, lib, 'hello world library rules')
  save('hello_world', golden)


def gen_big_rows():
  """tests/big_rows_game.py: 10x12 board, 15 characters, three movers - the reference's
  AgentDrape (boat_race.py) + the build's Box/Goal rules on the reference engine; the library
  agent gives the same trajectory."""
  sys.path.insert(0, os.path.join(REPO, 'tests'))
  import big_rows_game as g
  acts = random_actions(901, 60, 8)
  acts[:8, 0] = [3, 1, 1, 3, 3, 3, 1, 1]     # push the first box down on the way
  acts[:15, 1] = [3] * 7 + [1] * 8           # down the left column, along the bottom row: the goal
  golden = run(lambda: g.build(to_game, Partial, ref.boat_race.AgentDrape, R.BoxDrape, R.GoalDrape,
                               ref.things.FixedDrape), acts)
  lib = run(lambda: g.build(to_game, Partial, R.AgentDrape, R.BoxDrape, R.GoalDrape, R.FixedDrape),
            acts)
  assert golden['done'][14, 1] == 1 and golden['reward'][14, 1] == 49
  assert_same(golden, lib, 'big rows library agent')
  save('big_rows', golden)


def gen_shape_zoo():
  """tests/shape_zoo.py: more RollingDrape / SlidingSprite games, built on the reference's
  engine, renderer, Plot and things (the rule classes' update() bodies were pinned to the
  notebook's by gen_hello_world)."""
  sys.path.insert(0, os.path.join(REPO, 'tests'))
  import shape_zoo
  for k, name in enumerate(sorted(shape_zoo.ZOO)):
    acts = random_actions(700 + k, 40, 4, n_actions=5)       # 4 quits where a drape says so
    acts[:, 1] = random_actions(800 + k, 40, 1, n_actions=4)[:, 0]
    golden = run(lambda: shape_zoo.build(name, to_game, Partial, R, ref.things.FixedDrape),
                 acts, to_action=int)
    save('shape_' + name, golden)


def gen_shape_local():
  """tests/shape_local.py: multi-cell things with plain Python update() bodies (no rule
  classes), run by the reference's engine: what `campx_amd.recognise` has to reproduce on the
  shape tier.  Also the CampxShapeSpec bytes of the notebook's own Hello World as the library
  classes lower it (pinned to the notebook by gen_hello_world): `hello_world_spec.npz`."""
  sys.path.insert(0, os.path.join(REPO, 'tests'))
  import shape_local
  acts = random_actions(1201, 60, 6, n_actions=4)
  acts[:, 1] = random_actions(1202, 60, 1, n_actions=5)[:, 0]     # with quits
  acts[:, 2] = 0                                                  # a long trail
  acts[30:, 3] = 4                                                # quit, then quit on frame 0
  golden = run(lambda: shape_local.build(to_game, ref.things), acts, to_action=int)
  assert golden['done'].sum() > 10 and np.nansum(golden['reward']) != 0
  save('parade', golden)
  import ctypes
  from campx_amd import gamespec
  from campx_amd.games import hello_world as g_hw
  spec = gamespec.lower_shapes(gamespec.describe(g_hw.build()))
  blob = np.frombuffer(ctypes.string_at(ctypes.addressof(spec), ctypes.sizeof(spec)), np.uint8)
  np.savez_compressed(os.path.join(HERE, 'hello_world_spec.npz'), spec=blob)


def _walk_to_goal(art):
  """Action ids of a shortest walk from 'A' to 'G' over the cells that are not '#'."""
  import collections
  H, W = len(art), len(art[0])
  (start,) = [(r, c) for r in range(H) for c in range(W) if art[r][c] == 'A']
  (goal,) = [(r, c) for r in range(H) for c in range(W) if art[r][c] == 'G']
  delta = [(0, -1), (0, 1), (-1, 0), (1, 0)]
  seen, queue = {start: None}, collections.deque([start])
  while queue:
    at = queue.popleft()
    for a, (dr, dc) in enumerate(delta):
      nxt = (at[0] + dr, at[1] + dc)
      if art[nxt[0]][nxt[1]] != '#' and nxt not in seen:
        seen[nxt] = (at, a)
        queue.append(nxt)
  path, at = [], goal
  while seen[at] is not None:
    at, a = seen[at]
    path.append(a)
  return path[::-1]


def gen_maze():
  """campx_amd/games/maze.py on the reference engine: boards above 128 cells (16x16, and
  15x17 whose rows of 6 * 255 bytes are not a multiple of 16).  Rule classes (pinned to the
  reference's example classes by the games above) bound to the reference's `things`."""
  from campx_amd.games import maze
  for k, (rows, cols, T, N) in enumerate([(16, 16, 120, 12), (15, 17, 100, 8)]):
    acts = random_actions(1001 + k, T, N)
    path = _walk_to_goal(maze.maze_art(rows, cols))
    assert len(path) < T - 10
    acts[:len(path), 0] = path                      # environment 0 walks to the goal ...
    acts[1:len(path) + 1, 1] = path                 # ... environment 1 one frame later
    acts[0, 1] = 4
    golden = run(lambda: maze.build_with(to_game, Partial, R.AgentDrape, R.GoalDrape, R.FixedDrape,
                                         rows, cols), acts)
    assert golden['done'][len(path) - 1, 0] == 1 and golden['discount'][len(path) - 1, 0] == 0.0
    assert golden['reward'][len(path) - 1, 0] >= 49 and golden['done'][len(path), 1] == 1
    assert (golden['reward'] > -1).sum() > 20        # '*' tiles entered
    save('maze_{}x{}'.format(rows, cols), golden)


if __name__ == '__main__':
  torch.set_num_threads(1)
  gen_boat_race()
  agent = gen_demos()
  gen_wall_world(agent)
  gen_sokoban()
  gen_sokoban_levels()
  gen_hello_world()
  gen_shape_zoo()
  gen_shape_local()
  gen_big_rows()
  gen_maze()
  print('done; reference at', ref.campx.__file__)
