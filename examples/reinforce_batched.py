#!/usr/bin/env python3
"""REINFORCE on the boat race, B environments at once, the whole loop on the GPU.

The batched counterpart of the reference's driver (examples/reinforce.py): same policy
network shape (one hidden layer, reinforce.py:53-67), same episode length (100), same
log columns (`id,step,t(s),ep,L,R,R_av_5,P,P_av`, reinforce.py:270-284) - but one
`Engine` holds B environments, `play()` hands the policy its input directly in bf16
(`set_play_obs_dtype`, no `.float()` pass), actions are sampled on the device and go
back into `play()` as int8 ids, and the hidden performance is scored in the kernel.

    python examples/reinforce_batched.py --batch 4096 --episodes 20 --csv /tmp/log.csv
    python examples/reinforce_batched.py --batch 4096 --episodes 20 --graph

`--graph`: the acting loop - policy forward, sampling, `play()`, 100 frames of them - is captured
once in a HIP graph (`Engine.capture_play`, campx_amd/play_graph.py) and replayed per episode: one
launch of the host's instead of ~1 000 op dispatches; the learner then recomputes the log-
probabilities of the actions taken, with gradients, in ONE batched forward pass over the recorded
observations [T * B, n_in] (the usual act-then-learn split).

A consumer of the engine, not part of it (RL drivers are out of
scope); it exists to show the hand-off and is smoke-tested in tests/test_example.py.
"""

import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from campx_amd.episode_log import EpisodeCsvLog  # noqa: E402
from campx_amd.games import boat_race  # noqa: E402


class Policy(torch.nn.Module):
  def __init__(self, n_in, hidden=32, n_out=5):
    super().__init__()
    self.affine1 = torch.nn.Linear(n_in, hidden)
    self.affine2 = torch.nn.Linear(hidden, n_out)

  def forward(self, x):
    return torch.log_softmax(self.affine2(torch.relu(self.affine1(x))), dim=-1)


def run(batch=4096, episodes=10, frames=100, gamma=0.99, lr=1e-2, csv=None, seed=0,
        device='cuda', graph=False):
  torch.manual_seed(seed)
  game, obs, _, _ = boat_race.make_game(batch=batch, device=device)
  fused = game.fused
  fused.set_play_obs_dtype(torch.bfloat16)
  fused.validate_actions = False                 # ids come from multinomial: always 0..4
  n_in = fused.n_layers * fused.rows * fused.cols
  policy = Policy(n_in).to(device=device, dtype=torch.bfloat16)
  optim = torch.optim.Adam(policy.parameters(), lr=lr)
  log = EpisodeCsvLog(csv, frames_per_episode=frames) if csv else None
  history = []
  acting = None
  if graph:
    def act(observation, t):
      logp = policy(observation.layered_board.view(batch, n_in))
      return torch.multinomial(logp.float().exp(), 1).squeeze(1)
    acting = game.capture_play(frames, policy=act, record_obs=True)
  for episode in range(episodes):
    obs, _, _ = fused.reset()                           # new episode: rebuild from the art
    log_probs, rewards, perf = [], [], torch.zeros(batch, device=device)
    if graph:
      acting.replay()                                   # `frames` x (forward, sample, play)
      seen = acting.obs.view(frames * batch, n_in)      # what the policy saw, bf16, as recorded
      logp = policy(seen).view(frames, batch, -1)       # ... once more, this time with gradients
      log_probs = list(logp.gather(2, acting.actions.long()[:, :, None]).squeeze(2).float())
      rewards = list(acting.reward)
      perf = acting.perf.float().sum(0)
    for t in range(0 if graph else frames):
      # bf16 straight from the kernel, no conversion - but a COPY: play() overwrites the
      # engine's frame buffer in place, and autograd keeps the policy's input for backward
      # (the campx:: ops bump the buffer's version counter, so feeding the view itself makes
      # backward() raise instead of differentiating the last frame T times)
      logp = policy(obs.layered_board.view(batch, n_in).clone())
      ids = torch.multinomial(logp.float().exp(), 1).squeeze(1)
      log_probs.append(logp.gather(1, ids[:, None]).squeeze(1).float())
      obs, reward, _ = game.play(ids.to(torch.int8))
      rewards.append(reward.clone())
      perf += fused.perf.float()
    returns, running = [], torch.zeros(batch, device=device)
    for r in reversed(rewards):
      running = r + gamma * running
      returns.append(running)
    returns = torch.stack(returns[::-1])
    returns = (returns - returns.mean()) / (returns.std() + 1e-6)
    loss = -(torch.stack(log_probs) * returns).sum(0).mean()
    optim.zero_grad()
    loss.backward()
    optim.step()
    episode_return = torch.stack(rewards).sum(0)
    history.append((float(loss.detach()), float(episode_return.mean()), float(perf.mean())))
    if log:
      log.episode(episode_return, perf, loss=float(loss.detach()))
  if log:
    log.close()
  return history


if __name__ == '__main__':
  p = argparse.ArgumentParser()
  p.add_argument('--batch', type=int, default=4096)
  p.add_argument('--episodes', type=int, default=10)
  p.add_argument('--frames', type=int, default=100)
  p.add_argument('--csv', default=None)
  p.add_argument('--graph', action='store_true', help='the acting loop as one HIP graph per episode')
  args = p.parse_args()
  for i, (loss, ret, perf) in enumerate(run(args.batch, args.episodes, args.frames, csv=args.csv,
                                            graph=args.graph)):
    print('ep: {}, L: {:.3f}, R: {:.2f}, P: {:.2f}'.format(i, loss, ret, perf))
