"""`Engine.capture_play()`: n frames of `play()` - and the policy in front of each - in one HIP graph
(campx_amd/play_graph.py), bit for bit what frame-by-frame `play()` gives.

VERDICT r5 item 6: the graph form of `play()` was a leg of bench.py, 20-30 % faster than a call per
frame, and not an API.  Here: an open-loop graph against `play()` on a twin engine for every tier
(one-cell with one and three movers, wide, shape), replays carrying state into each other and
into ordinary `play()` / `rollout()` calls; a graph that holds a policy network, its sampling and
`play()` against the same loop run eagerly with the same generator state; the batched REINFORCE
example's graph form.
"""
import os
import sys

import numpy as np
import pytest
import torch

from campx_amd.games import boat_race, sokoban
from conftest import REPO
import random_pickups
from games_under_test import SHAPE_GAMES, WIDE_GAMES

pytestmark = pytest.mark.gpu


def _pair(build, B, **kw):
  a, b = build(batch=B, device='cuda', **kw), build(batch=B, device='cuda', **kw)
  a.its_showtime()
  b.its_showtime()
  return a, b


def _same(x, y):
  if x is None or y is None:
    return x is None and y is None
  if x.dtype.is_floating_point:
    return torch.equal(x.view(torch.int32) if x.dtype == torch.float32 else x.view(torch.int16),
                       y.view(torch.int32) if y.dtype == torch.float32 else y.view(torch.int16))
  return torch.equal(x, y)


@pytest.mark.parametrize('name,build,kw,B', [
    ('boat_race', boat_race.build, {}, 4097),
    ('sokoban_l1', sokoban.build, {'level': 1}, 2048),
    ('maze_16x16', WIDE_GAMES['maze_16x16'], {}, 1000),
    ('hello_world', SHAPE_GAMES['hello_world'], {}, 333),
    # a scenery that changes (tests/random_pickups.py): seven coins as pieces in a mask, a floor in
    # three variants - their play() is the update + render pair, two kernel nodes a frame
    ('pickup3_coins', random_pickups.builder(random_pickups.definitions()[3]), {}, 777),
    ('pickup13_seasons', random_pickups.builder(random_pickups.definitions()[13]), {}, 1025)])
def test_open_loop_graph_equals_frame_by_frame_play(name, build, kw, B):
  n = 12
  a, b = _pair(build, B, **kw)
  graph = a.capture_play(n)
  assert a.fused.frame == 0                      # warm-up and capture left the game where it was
  gen = torch.Generator(device='cuda').manual_seed(5)
  top = 4 if name == 'hello_world' else 5        # (Hello World's action 4 ends the episode: keep some play)
  for replay in range(3):
    acts = torch.randint(0, top, (n, B), generator=gen, device='cuda', dtype=torch.int64)
    graph.replay(acts if replay else acts.to(torch.int8))
    for t in range(n):
      obs, reward, discount = b.play(acts[t].to(torch.int8))
      assert _same(graph.reward[t] if graph.reward is not None else None, reward), (name, replay, t)
      assert _same(graph.discount[t], discount), (name, replay, t)
      assert torch.equal(graph.done[t], b.fused._step_done), (name, replay, t)
      if graph.perf is not None:
        assert torch.equal(graph.perf[t], b.fused.perf), (name, replay, t)
    assert torch.equal(graph.observation.layered_board, obs.layered_board), (name, replay)
    assert torch.equal(graph.observation.board, obs.board), (name, replay)
    assert a.fused.frame == b.fused.frame == n * (replay + 1) + replay
    # ... and an ordinary call between two replays carries on from where the graph left the game
    one = torch.randint(0, top, (B,), generator=gen, device='cuda', dtype=torch.int8)
    oa, ra, _ = a.play(one)
    ob, rb, _ = b.play(one)
    assert torch.equal(oa.layered_board, ob.layered_board) and _same(ra, rb)


def test_a_graph_that_holds_the_policy_its_sampling_and_play():
  """The loop of examples/reinforce.py:136-149 - forward, sample, play - 16 frames per launch of
  the host's, against the same loop run op by op.  The policy reads the bf16 observation straight
  from the engine's buffer; sampling is `torch.multinomial` on the default generator, whose state
  a graph replay advances exactly as the eager calls do."""
  B, n = 2048, 16
  a, b = _pair(boat_race.build, B)
  for g in (a, b):
    g.fused.set_play_obs_dtype(torch.bfloat16)
    g.fused.validate_actions = False
  n_in = a.fused.n_layers * a.fused.rows * a.fused.cols
  torch.manual_seed(3)
  net = torch.nn.Sequential(torch.nn.Linear(n_in, 32), torch.nn.ReLU(), torch.nn.Linear(32, 5)).to(
      device='cuda', dtype=torch.bfloat16)

  def policy(obs, t):
    logits = net(obs.layered_board.view(B, n_in))
    return torch.multinomial(torch.softmax(logits.float(), dim=-1), 1).squeeze(1)

  graph = a.capture_play(n, policy=policy, record_obs=True)
  assert graph.obs.dtype == torch.bfloat16 and tuple(graph.obs.shape) == (n, B) + tuple(a.fused._obs.shape[1:])
  for replay in range(3):
    torch.cuda.manual_seed(100 + replay)
    graph.replay()
    torch.cuda.synchronize()
    torch.cuda.manual_seed(100 + replay)
    with torch.no_grad():
      for t in range(n):
        seen = b.fused._obs.clone()
        ids = policy(b.fused._observation_cache, t)
        _, reward, discount = b.play(ids.to(torch.int8))
        assert torch.equal(graph.actions[t].long(), ids), (replay, t)
        assert torch.equal(graph.obs[t], seen), (replay, t)
        assert _same(graph.reward[t], reward) and _same(graph.discount[t], discount)
        assert torch.equal(graph.perf[t], b.fused.perf)
    assert torch.equal(a.fused._obs, b.fused._obs) and torch.equal(a.fused.pos, b.fused.pos)
  assert 0 < float((graph.actions == 1).float().mean()) < 1      # the policy did choose


def test_what_a_graph_cannot_hold_is_refused():
  game = boat_race.build(batch=64, device='cuda')
  with pytest.raises(RuntimeError, match='its_showtime'):
    game.capture_play(4)
  game.its_showtime()
  with pytest.raises(ValueError, match='at least one frame'):
    game.capture_play(0)
  game.fused.validate_actions = 'sync'
  with pytest.raises(ValueError, match='sync'):
    game.capture_play(4)
  game.fused.validate_actions = True
  graph = game.capture_play(4)
  with pytest.raises(ValueError, match='shape'):
    graph.replay(torch.zeros((3, 64), dtype=torch.int8, device='cuda'))
  # bad ids are counted by the kernels inside the graph and reported like anywhere else
  bad = torch.full((4, 64), 9, dtype=torch.int64, device='cuda')
  with pytest.raises(ValueError, match='outside 0..4'):     # (at the replay already, if the flag is up by then)
    graph.replay(bad)
    game.fused.check_actions()
  # a graph holds addresses: another observation buffer (a change of play()'s dtype) means another graph
  stale = game.capture_play(2)
  game.fused.set_play_obs_dtype(torch.bfloat16)
  with pytest.raises(RuntimeError, match='capture again'):
    stale.replay(torch.zeros((2, 64), dtype=torch.int8, device='cuda'))
  game.fused.set_play_obs_dtype(torch.int8)
  with_policy = game.capture_play(2, policy=lambda obs, t: torch.zeros(64, dtype=torch.int8, device='cuda'))
  with pytest.raises(ValueError, match='policy'):
    with_policy.replay(torch.zeros((2, 64), dtype=torch.int8, device='cuda'))
  with pytest.raises(ValueError, match=r'shape \[B\]'):
    game.capture_play(2, policy=lambda obs, t: torch.zeros((64, 1), dtype=torch.int8, device='cuda'))


def test_the_batched_reinforce_example_learns_through_the_graph():
  sys.path.insert(0, os.path.join(REPO, 'examples'))
  import reinforce_batched
  eager = reinforce_batched.run(batch=512, episodes=3, frames=20, seed=1)
  graphed = reinforce_batched.run(batch=512, episodes=3, frames=20, seed=1, graph=True)
  assert len(eager) == len(graphed) == 3
  for loss, ret, perf in graphed:
    assert np.isfinite(loss) and -60.0 <= ret <= 60.0 and -20.0 <= perf <= 20.0
