#!/usr/bin/env python3
"""Generate `traced_<game>.npz`: what the REFERENCE engine does with the test-local games of
tests/traced_games.py (arbitrary Python classes: a skater that slides many cells, a sprite
that mirrors a walker, a walker whose tiles change the frame's discount, a mole that
changes its place in the z-order, a key, a door and a gem that leave and enter the board).

Run in the build container only (needs /root/reference):

    python tests/golden/make_traced_golden.py

tests/traced_games.py imports `campx.things` / `campx.ascii_art` and nothing else of an
engine, so in THIS process - /root/reference first on sys.path, through `ref_harness` - the
very same file builds its games on the reference's base classes, engine, renderer and Plot.
One reference run per environment (the reference has no batch axis); an environment whose
episode ended gets a fresh game before its next action (examples/reinforce.py:122).

Arrays (T steps, N environments, L characters ascending), as in make_golden.py:
  chars [L], actions [T, N] int8, board [T+1, N, H, W] int8 (index 0: its_showtime()),
  layered [T+1, N, L, H, W] uint8, reward [T, N] float32 (NaN = None), discount [T, N]
  float32, done [T, N] uint8.

tests/test_tabulate.py replays them on this repo's generic tier, through the tabulated
table, and - on the GPU - through the table and render kernels.
"""

import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import ref_harness  # noqa: E402

ref = ref_harness.load()
sys.path.append(os.path.dirname(HERE))      # tests/: traced_games.py (after the reference)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import traced_games  # noqa: E402

assert traced_games.things is ref.things    # the reference's base classes


def one_hot(a):
  v = torch.zeros(5)
  v[int(a)] = 1
  return v


def run(build, actions):
  T, N = actions.shape
  out = None
  for n in range(N):
    game = build()
    obs, reward, discount = game.its_showtime()
    assert reward is None and discount == 1.0
    chars = sorted(obs.layers.keys())
    if out is None:
      H, W = obs.board.shape
      out = dict(chars=np.array([ord(c) for c in chars], np.uint8),
                 actions=actions.astype(np.int8),
                 board=np.zeros((T + 1, N, H, W), np.int8),
                 layered=np.zeros((T + 1, N, len(chars), H, W), np.uint8),
                 reward=np.zeros((T, N), np.float32), discount=np.zeros((T, N), np.float32),
                 done=np.zeros((T, N), np.uint8))

    def record(i, obs):
      out['board'][i, n] = obs.board.numpy()
      for k, ch in enumerate(chars):
        out['layered'][i, n, k] = obs.layers[ch].numpy()

    record(0, obs)
    for t in range(T):
      if game._game_over:
        game = build()
        game.its_showtime()
      obs, reward, discount = game.play(one_hot(actions[t, n]))
      record(t + 1, obs)
      out['reward'][t, n] = np.nan if reward is None else float(reward)
      out['discount'][t, n] = float(discount)
      out['done'][t, n] = int(game._game_over)
  return out


def main():
  seeds = dict(ice_rink=300, mirror=301, toll_road=302, trio=303, burrow=304, vault=305)
  for name in sorted(traced_games.GAMES):
    actions = np.random.RandomState(seeds[name]).randint(0, 5, size=(80, 24))
    data = run(traced_games.GAMES[name], actions)
    path = os.path.join(HERE, 'traced_' + name + '.npz')
    np.savez_compressed(path, **data)
    print('{:10s} T={} N={} chars={!r} episodes ended={} discounts={} -> {} KiB'.format(
        name, actions.shape[0], actions.shape[1], ''.join(chr(c) for c in data['chars']),
        int(data['done'].sum()), sorted(set(data['discount'].ravel().tolist())),
        os.path.getsize(path) // 1024))


if __name__ == '__main__':
  main()
