"""GameSpec lowering (host logic) and the C ABI's load/export/validate surface.
No kernel is launched here."""

import ctypes
import os
import re

import numpy as np
import pytest

from campx_amd import gamespec, rules, things, _hip
from campx_amd.ascii_art import ascii_art_to_game, Partial
from campx_amd.games import boat_race, sokoban, wall_world
from conftest import REPO
from games_under_test import FUSED_GAMES


def header_functions():
  text = open(os.path.join(REPO, 'include', 'campx_hip.h')).read()
  text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
  return sorted(set(re.findall(r'\b(campx_[a-z_]+)\s*\(', text)))


def test_library_exports_every_symbol_the_header_declares():
  declared = header_functions()
  assert 'campx_rollout_launch' in declared and 'campx_spec_compile' in declared
  for name in declared:
    assert hasattr(_hip.lib, name), name
  assert sorted(_hip.EXPORTS) == declared


def test_struct_layouts_match_the_library():
  assert _hip.lib.campx_spec_size() == ctypes.sizeof(gamespec.CampxSpec)
  assert ctypes.sizeof(gamespec.CampxRule) == 64
  assert ctypes.sizeof(gamespec.CampxTransition) == 8
  assert gamespec.CampxSpec.rot_obs.offset % 16 == 0
  assert gamespec.CampxSpec.rot_board.offset % 16 == 0
  assert _hip.lib.campx_strerror(0) == b'ok'
  assert b'GameSpec' in _hip.lib.campx_strerror(-2)


@pytest.mark.parametrize('name', sorted(FUSED_GAMES))
def test_every_game_lowers_to_a_valid_spec(name):
  spec = gamespec.lower(gamespec.describe(FUSED_GAMES[name]()))
  assert _hip.lib.campx_spec_validate(ctypes.byref(spec)) == 0
  HW = spec.rows * spec.cols
  tmpl = np.ctypeslib.as_array(spec.obs_template)[:spec.n_layers * HW].reshape(spec.n_layers, HW)
  assert (tmpl.sum(0) == 1).all()           # the scenery shows one character per cell


def test_boat_race_spec_contents():
  desc = gamespec.describe(boat_race.build())
  assert desc.chars == [' ', '#', '<', '>', 'A', '^', 'v'] and desc.z_order == list('^>v<A#')
  assert [e.char for e in desc.entities] == list('A^>v<#')
  spec = gamespec.lower(desc)
  assert (spec.rows, spec.cols, spec.n_layers, spec.n_dyn, spec.n_static) == (5, 5, 7, 1, 5)
  assert (spec.dyn_row0[0], spec.dyn_col0[0], spec.dyn_layer[0], spec.dyn_z[0]) == (1, 1, 4, 5)
  agent = spec.rules[0]
  assert agent.op == gamespec.OP_AGENT and agent.block_layers == 1 << 1 and not agent.has_reward
  hover = spec.rules[1]                      # '^': dctns [0, 0, 3, 1, 0], base -0.25
  assert hover.op == gamespec.OP_DIR_HOVER and hover.aux == 5 and hover.base == -0.25
  assert list(hover.bonus) == [0, 0, 3, 1, 0]
  assert [spec.rules[i].end_group for i in range(5)] == [0, 0, 0, 0, 1]
  # where the agent starts the scenery shows the backdrop's ' '
  assert spec.static_top_layer[6] == 0 and spec.static_top_z[6] == 0
  assert spec.static_top_layer[0] == 1 and spec.static_top_z[0] == 6     # '#', front-most


def test_sokoban_spec_has_two_update_groups():
  spec = gamespec.lower(gamespec.describe(sokoban.build()))
  ops = [spec.rules[i].op for i in range(spec.n_rules)]
  assert ops == [gamespec.OP_BOX, gamespec.OP_AGENT, gamespec.OP_GOAL]
  assert [spec.rules[i].end_group for i in range(3)] == [1, 0, 1]
  assert spec.n_dyn == 2 and spec.rules[0].aux == 1       # the box is pushed by dyn 1 (agent)


def test_validate_rejects_corrupt_specs():
  good = gamespec.lower(gamespec.describe(wall_world.build()))

  def corrupt(**kw):
    spec = gamespec.CampxSpec.from_buffer_copy(gamespec.spec_bytes(good))
    for k, v in kw.items():
      setattr(spec, k, v)
    return _hip.lib.campx_spec_validate(ctypes.byref(spec))

  assert corrupt() == 0
  assert corrupt(magic=1) == -2
  assert corrupt(rows=0) == -2
  assert corrupt(rows=100) == -2               # 100 x 10 cells > CAMPX_MAX_CELLS
  assert corrupt(n_layers=17) == -2
  assert corrupt(n_dyn=0) == -2
  assert corrupt(n_rules=17) == -2
  spec = gamespec.CampxSpec.from_buffer_copy(gamespec.spec_bytes(good))
  spec.rules[0].op = 9
  assert _hip.lib.campx_spec_validate(ctypes.byref(spec)) == -2
  spec = gamespec.CampxSpec.from_buffer_copy(gamespec.spec_bytes(good))
  spec.rules[0].dyn = 3
  assert _hip.lib.campx_spec_validate(ctypes.byref(spec)) == -2
  assert _hip.lib.campx_spec_validate(None) == -1


def test_launch_argument_checks_without_a_gpu():
  """NULL buffers are rejected before anything touches the device."""
  spec = gamespec.lower(gamespec.describe(boat_race.build()))
  st = _hip.CampxState(None, None, None, None)
  out = _hip.CampxOutputs()
  assert _hip.lib.campx_rollout_launch(ctypes.byref(spec), None, st, None, out, 64, 1, 0, None) == -1
  assert _hip.lib.campx_check_actions_launch(None, 4, None, None) == -1
  assert _hip.lib.campx_onehot_to_ids_launch(None, None, 4, None, None) == -1


def test_state_table_sizes():
  """campx_pair_table_bytes(): host arithmetic, no GPU.  1 KiB reward list + one entry per
  (cell, ..., cell, action): uint32 for two movers (<= 1 MiB), uint64 for three and four
  (<= 512 MiB); nothing for one mover (its table lives in the spec) or oversized boards."""
  from campx_amd.games import sokoban, wall_world

  def table_bytes(game):
    spec = gamespec.lower(gamespec.describe(game))
    return int(_hip.lib.campx_pair_table_bytes(ctypes.byref(spec))), spec

  assert table_bytes(boat_race.build())[0] == 0
  assert table_bytes(wall_world.build())[0] == 0
  assert table_bytes(sokoban.build())[0] == 1024 + 36 * 36 * 5 * 4
  assert table_bytes(sokoban.build(level=1))[0] == 1024 + 48 ** 3 * 5 * 8
  n4, spec4 = table_bytes(sokoban.build(level=2))
  assert n4 == 1024 + 48 ** 4 * 5 * 8
  spec4.rows, spec4.cols = 8, 16            # 128 cells, four movers: 10 GiB -> not tabulated
  for d in range(4):
    spec4.dyn_row0[d], spec4.dyn_col0[d] = 1, 1 + d
  assert _hip.lib.campx_pair_table_bytes(ctypes.byref(spec4)) in (0,)
  assert _hip.lib.campx_pair_table_bytes(None) == 0
  assert _hip.lib.campx_pair_table_build(None, None, None, None) == -1


def test_lowering_refuses_what_the_cell_model_cannot_express():
  def lower(*a, **k):
    return gamespec.lower(gamespec.describe(ascii_art_to_game(*a, **k)))

  # two agent cells
  with pytest.raises(ValueError, match='exactly one cell'):
    lower(['AA', '  '], ' ', drapes={'A': rules.AgentDrape})
  # a thing painted in front of the agent that does not block it
  with pytest.raises(ValueError, match='painted in front of agent'):
    lower(['A*'], ' ', drapes={'A': Partial(rules.AgentDrape, blocking_chars=''),
                               '*': things.FixedDrape}, z_order='A*')
  # blocking character that is not in the game
  with pytest.raises(ValueError, match='not in this game'):
    lower(['A '], ' ', drapes={'A': Partial(rules.AgentDrape, blocking_chars='#')})
  # no moving thing at all
  with pytest.raises(ValueError, match='moving things'):
    lower(['# '], ' ', drapes={'#': things.FixedDrape})
  # hover reward watching something static
  with pytest.raises(ValueError, match='not a moving thing'):
    lower(['A>#'], ' ', drapes={
        'A': rules.AgentDrape, '#': things.FixedDrape,
        '>': Partial(rules.DirectionalHoverRewardDrape, agent_chars='#',
                     dctns=[0, 1, 0, 0, 0])}, z_order='>A#')
  # a sprite
  class S(things.Sprite):
    def update(self, *a):
      pass
  with pytest.raises(ValueError, match="sprite 'S' is a S, which cannot be lowered"):
    lower(['AS'], ' ', sprites={'S': S}, drapes={'A': Partial(rules.AgentDrape, blocking_chars='')},
          z_order='SA')
  # shape rules (Hello World) do not mix with interacting one-cell rules
  with pytest.raises(ValueError, match='moving things|cannot be mixed'):
    g = ascii_art_to_game(['A@'], ' ', drapes={'A': Partial(rules.AgentDrape, blocking_chars=''),
                                                '@': rules.RollingDrape}, z_order='A@')
    d = gamespec.describe(g)
    assert d.is_shape_game
    gamespec.lower_shapes(d)
  # board too large for the bit/cell tables
  with pytest.raises(ValueError, match='more than 128 cells'):
    lower(['A' + ' ' * 12] + [' ' * 13] * 9, ' ',
          drapes={'A': Partial(rules.AgentDrape, blocking_chars='')})


def test_hello_world_lowers_to_a_valid_shape_spec():
  from campx_amd.games import hello_world
  desc = gamespec.describe(hello_world.build())
  assert desc.is_shape_game and desc.z_order == list('12@34')
  spec = gamespec.lower_shapes(desc)
  assert _hip.lib.campx_shape_spec_validate(ctypes.byref(spec)) == 0
  assert _hip.lib.campx_shape_spec_size() == ctypes.sizeof(gamespec.CampxShapeSpec)
  assert (spec.rows, spec.cols, spec.n_layers, spec.n_things, spec.first_drape) == (13, 36, 7, 5, 2)
  drape = spec.things[2]
  assert (drape.is_sprite, drape.n_cells, drape.terminate_mask, drape.has_reward_mask) == (0, 59, 16, 15)
  assert list(drape.drow)[:5] == [12, 1, 0, 0, 0] and list(drape.dcol)[:5] == [0, 0, 35, 1, 0]
  # sprites 1 and 2 sit behind the drape: their art cells are already in the backdrop
  # (the reference renderer's aliasing); sprites 3 and 4 are not
  layer = {chr(spec.layer_char[i]): i for i in range(spec.n_layers)}
  assert spec.backdrop[7 * 36 + 34] == layer['1'] and spec.backdrop[8 * 36 + 34] == layer['2']
  assert spec.backdrop[9 * 36 + 34] == layer[' '] and spec.backdrop[11 * 36 + 34] == layer[' ']
  # corrupting it is caught
  spec.first_drape = 0          # thing 0 is a sprite: not a drape
  assert _hip.lib.campx_shape_spec_validate(ctypes.byref(spec)) == -2


def test_a_game_of_sprites_only_is_refused():
  g = ascii_art_to_game(['1 '], ' ', sprites={'1': Partial(rules.SlidingSprite, 0)})
  with pytest.raises(ValueError, match='no drape at all'):
    gamespec.lower_shapes(gamespec.describe(g))
