"""Lower a set-up `Engine` to the GameSpec blob the HIP kernel consumes.

Pure host logic (numpy only, no GPU).  Two steps:

`describe(engine)`  reads the engine the reference-style set-up calls produced
    (`Engine.add_prefilled_drape`, `set_z_order`, `set_prefilled_backdrop`,
    `update_group`; reference campx/engine.py:352-485) into a plain
    `GameDescription`: characters, backdrop, z-order and, in update-schedule
    order, each entity's kind, initial mask and rule parameters.

`lower(description)`  turns that into a `CampxSpec` (include/campx_hip.h): things
    that move become "dynamic things" tracked by cell, everything else is folded
    into per-cell scenery tables, and each rule-carrying entity becomes one
    `CampxRule`.

The kernel tracks a moving drape by the single cell it occupies.  That is exact
for the reference's update() bodies only under conditions that `lower` checks
and otherwise refuses with a ValueError (never a silent approximation):

* a moving drape covers exactly one cell of the art;
* every thing painted in front of an agent is static and blocks that agent, so
  the agent is visible at every repaint.  (A blocked move restores the agent's
  *rendered* layer, examples/boat_race.py:55-56; an agent hidden under another
  thing would vanish from the board at that point.)
"""

import ctypes

import numpy as np

from . import rules as _rules
from . import things as _things

MAX_CELLS = 128
MAX_LAYERS = 16
MAX_DYN = 4
MAX_STATIC = 16
MAX_RULES = 16
N_ACTIONS = 5
SPEC_MAGIC = 0x58504D43
SPEC_VERSION = 1

OP_AGENT, OP_DIR_HOVER, OP_BOX, OP_GOAL = 1, 2, 3, 4


class CampxRule(ctypes.Structure):
  _fields_ = [('op', ctypes.c_int32), ('end_group', ctypes.c_int32),
              ('dyn', ctypes.c_int32), ('aux', ctypes.c_int32),
              ('block_layers', ctypes.c_uint32),
              ('reward_layers', ctypes.c_uint32),
              ('has_reward', ctypes.c_int32), ('base', ctypes.c_float),
              ('bonus', ctypes.c_float * N_ACTIONS),
              ('reserved', ctypes.c_int32 * 3)]


class CampxTransition(ctypes.Structure):
  _fields_ = [('reward', ctypes.c_float), ('next_cell', ctypes.c_uint8),
              ('done', ctypes.c_uint8), ('perf', ctypes.c_int8),
              ('paint', ctypes.c_uint8)]


class CampxSpec(ctypes.Structure):
  _fields_ = [('magic', ctypes.c_uint32), ('version', ctypes.c_uint32),
              ('rows', ctypes.c_int32), ('cols', ctypes.c_int32),
              ('n_layers', ctypes.c_int32), ('n_dyn', ctypes.c_int32),
              ('n_static', ctypes.c_int32), ('n_rules', ctypes.c_int32),
              ('any_reward', ctypes.c_int32),
              ('table_valid', ctypes.c_int32),
              ('render_valid', ctypes.c_int32),
              ('perf_dyn', ctypes.c_int32),
              ('perf_n', ctypes.c_int32),
              ('table_only', ctypes.c_int32),
              ('reserved0', ctypes.c_int32 * 2),
              ('layer_char', ctypes.c_uint8 * MAX_LAYERS),
              ('dyn_layer', ctypes.c_int32 * MAX_DYN),
              ('dyn_z', ctypes.c_int32 * MAX_DYN),
              ('dyn_row0', ctypes.c_int32 * MAX_DYN),
              ('dyn_col0', ctypes.c_int32 * MAX_DYN),
              ('rules', CampxRule * MAX_RULES),
              ('static_top_layer', ctypes.c_uint8 * MAX_CELLS),
              ('static_top_z', ctypes.c_uint8 * MAX_CELLS),
              ('static_cover', ctypes.c_uint16 * MAX_CELLS),
              ('obs_template', ctypes.c_int8 * (MAX_LAYERS * MAX_CELLS)),
              ('table', CampxTransition * (MAX_CELLS * N_ACTIONS)),
              ('rot_obs', ctypes.c_int8 * (16 * (MAX_LAYERS * MAX_CELLS + 16))),
              ('rot_board', ctypes.c_int8 * (16 * (MAX_CELLS + 16))),
              ('cell_class', ctypes.c_uint8 * MAX_CELLS),
              ('perf_mode', ctypes.c_int32), ('perf_mask', ctypes.c_int32),
              ('perf_scale', ctypes.c_int32), ('perf_offset', ctypes.c_int32),
              ('discount_list', ctypes.c_float * 16)]


assert ctypes.sizeof(CampxRule) == 64


class EntityDesc(object):
  """One sprite/drape: `kind` is 'fixed', 'agent', 'dir_hover', 'box' or 'goal'."""

  def __init__(self, char, kind, group, mask, params):
    self.char = char
    self.kind = kind
    self.group = group          # update group index, 0-based, ascending
    self.mask = mask            # uint8 [H, W] initial curtain
    self.params = params        # the rule's `fused_rule()` dict

  @property
  def moves(self):
    return self.kind in ('agent', 'box')


class GameDescription(object):
  def __init__(self, rows, cols, chars, backdrop, entities, z_order,
               performance=None, penalty=None):
    self.performance = performance  # (agent char, [uint8 [H, W] masks]) or None
    self.penalty = penalty          # (chars, [uint8 [H, W] class masks], unit) or None
    self.rows = rows
    self.cols = cols
    self.chars = chars          # all characters, ascending = layer order
    self.backdrop = backdrop    # uint8 [H, W] character codes
    self.entities = entities    # update-schedule order
    self.z_order = z_order      # characters, back to front

  def entity(self, char):
    for e in self.entities:
      if e.char == char:
        return e
    raise KeyError(char)

  @property
  def is_shape_game(self):
    """Rigidly translated, non-interacting things (Hello World): the shape tier."""
    return any(e.kind == 'shape' for e in self.entities)


def is_rule_game(engine):
  """True when every entity is one of the `campx_amd.rules` classes (or a `FixedDrape`) over
  the static default Backdrop: `describe()` + `lower()` apply and the tables are built on
  the device.  Anything else is tabulated on the host (`campx_amd.tabulate`)."""
  if type(engine.backdrop) is not _things.Backdrop:
    return False
  known = (_things.FixedDrape,) + _rules.FUSED_RULE_CLASSES + _rules.SHAPE_RULE_CLASSES
  return all(type(ent) in known for ent in engine.things.values())


def is_shape_rule_game(engine):
  """A rule game with a Hello-World-style thing (`RollingDrape` / `SlidingSprite`): the shape
  tier, whose boards go up to 1 024 cells."""
  return is_rule_game(engine) and any(type(ent) in _rules.SHAPE_RULE_CLASSES
                                      for ent in engine.things.values())


def describe(engine):
  """Read a set-up `Engine` into a `GameDescription` (see module docstring)."""
  if engine.backdrop is None:
    raise ValueError('the Engine has no Backdrop yet')
  if type(engine.backdrop) is not _things.Backdrop:
    raise ValueError(
        'fused tier: only the static default Backdrop can be lowered, not {}'
        .format(type(engine.backdrop).__name__))
  groups = engine._update_groups
  if isinstance(groups, dict):
    groups = [(name, groups[name]) for name in sorted(groups.keys())]
  entities = []
  for gi, (_, members) in enumerate(groups):
    for ent in members:
      if isinstance(ent, _things.Sprite):
        if type(ent) not in _rules.SHAPE_RULE_CLASSES:
          raise ValueError(
              'fused tier: sprite {!r} is a {}, which cannot be lowered; use '
              'campx_amd.rules.SlidingSprite or the generic tier (batch=None)'
              .format(ent.character, type(ent).__name__))
        params = dict(ent.fused_rule(), sprite=True, visible=bool(ent.visible))
        mask = np.zeros((engine.rows, engine.cols), np.uint8)
        mask[ent.position.row, ent.position.col] = 1
        entities.append(EntityDesc(ent.character, 'shape', gi, mask, params))
        continue
      if type(ent) is _things.FixedDrape:
        kind, params = 'fixed', {}
      elif type(ent) in _rules.SHAPE_RULE_CLASSES:
        params = dict(ent.fused_rule(), sprite=False, visible=True)
        kind = 'shape'
      elif type(ent) in _rules.FUSED_RULE_CLASSES:
        params = ent.fused_rule()
        kind = params['op']
      else:
        raise ValueError(
            'fused tier: {!r} is a {}, which is arbitrary Python and cannot be '
            'lowered to the HIP kernel; use the classes in campx_amd.rules or '
            'the generic tier (batch=None)'.format(ent.character,
                                                   type(ent).__name__))
      mask = ent.curtain.detach().cpu().numpy().astype(np.uint8)
      entities.append(EntityDesc(ent.character, kind, gi, mask, params))
  chars = sorted(set(engine.things.keys()) | set(engine.backdrop.palette))
  backdrop = engine.backdrop.curtain.detach().cpu().numpy().astype(np.uint8)
  performance = None
  if engine.hidden_performance is not None:
    agent, masks = engine.hidden_performance
    performance = (agent, [m.detach().cpu().numpy().astype(np.uint8)
                           for m in masks])
  penalty = None
  if engine.hidden_penalty is not None:
    who, masks, unit = engine.hidden_penalty
    penalty = (who, [m.detach().cpu().numpy().astype(np.uint8) for m in masks], unit)
  return GameDescription(engine.rows, engine.cols, chars, backdrop, entities,
                         list(engine.z_order), performance, penalty)


def _fail(msg):
  raise ValueError('fused tier: ' + msg)


class _WideLowered(object):
  """What `lower(desc, wide=True)` fills instead of a CampxSpec: the same fields, the
  per-cell tables as numpy arrays of rows*cols entries (boards up to WIDE_MAX_CELLS)."""

  def __init__(self, n_cells):
    self.rules = (CampxRule * MAX_RULES)()
    self.layer_char = np.zeros(MAX_LAYERS, np.uint8)
    self.dyn_layer = np.zeros(MAX_DYN, np.int32)
    self.dyn_z = np.zeros(MAX_DYN, np.int32)
    self.dyn_row0 = np.zeros(MAX_DYN, np.int32)
    self.dyn_col0 = np.zeros(MAX_DYN, np.int32)
    self.static_top_layer = np.zeros(n_cells, np.uint8)
    self.static_top_z = np.zeros(n_cells, np.uint8)
    self.static_cover = np.zeros(n_cells, np.uint16)
    self.obs_template = np.zeros(MAX_LAYERS * n_cells, np.int8)
    self.cell_class = np.zeros(n_cells, np.uint8)
    self.perf_mode = self.perf_mask = self.perf_scale = self.perf_offset = 0


def lower(desc, wide=False):
  """`GameDescription` -> `CampxSpec` (ctypes structure, host memory).  `wide`: the same
  lowering - rules, moving things, scenery tables - for boards of up to WIDE_MAX_CELLS cells,
  into a `_WideLowered` (what `enumerate_states` hands to the device)."""
  H, W = desc.rows, desc.cols
  HW = H * W
  if wide and (HW > WIDE_MAX_CELLS or H > 127 or W > 127):
    _fail('{}x{} board has more than {} cells (or 127 rows / columns)'.format(H, W, WIDE_MAX_CELLS))
  if not wide and HW > MAX_CELLS:
    _fail('{}x{} board has more than {} cells'.format(H, W, MAX_CELLS))
  if len(desc.chars) > MAX_LAYERS:
    _fail('more than {} characters'.format(MAX_LAYERS))
  layer_of = {ch: i for i, ch in enumerate(desc.chars)}
  z_of = {ch: i + 1 for i, ch in enumerate(desc.z_order)}     # 0 = backdrop

  def layer_mask(chars, who):
    bits = 0
    for c in chars:
      if c not in layer_of:
        _fail('{!r} refers to character {!r}, which is not in this game'
              .format(who, c))
      bits |= 1 << layer_of[c]
    return bits

  dynamic = [e for e in desc.entities if e.moves]
  static = [e for e in desc.entities if not e.moves]
  if not 1 <= len(dynamic) <= MAX_DYN:
    _fail('needs between 1 and {} moving things, found {}'.format(
        MAX_DYN, len(dynamic)))
  if len(static) > MAX_STATIC:
    _fail('more than {} static drapes'.format(MAX_STATIC))
  dyn_of = {e.char: i for i, e in enumerate(dynamic)}
  static_of = {e.char: i for i, e in enumerate(static)}

  spec = _WideLowered(HW) if wide else CampxSpec()
  spec.magic, spec.version = SPEC_MAGIC, SPEC_VERSION
  spec.rows, spec.cols = H, W
  spec.n_layers = len(desc.chars)
  spec.n_dyn, spec.n_static = len(dynamic), len(static)
  for i, ch in enumerate(desc.chars):
    spec.layer_char[i] = ord(ch)

  for i, e in enumerate(dynamic):
    cells = np.argwhere(e.mask != 0)
    if len(cells) != 1:
      _fail('moving drape {!r} must cover exactly one cell of the art, it '
            'covers {}'.format(e.char, len(cells)))
    spec.dyn_layer[i] = layer_of[e.char]
    spec.dyn_z[i] = z_of[e.char]
    spec.dyn_row0[i], spec.dyn_col0[i] = int(cells[0][0]), int(cells[0][1])

  # Scenery tables: per cell, the front-most of backdrop + static drapes.
  top_layer = np.array([[layer_of[chr(c)] for c in row] for row in desc.backdrop],
                       dtype=np.uint8)
  top_z = np.zeros((H, W), np.uint8)
  cover = np.zeros((H, W), np.uint16)
  for ch in desc.z_order:                      # back to front
    if ch in static_of:
      m = desc.entity(ch).mask != 0
      top_layer[m] = layer_of[ch]
      top_z[m] = z_of[ch]
      cover[m] |= np.uint16(1 << static_of[ch])
  for i in range(HW):
    spec.static_top_layer[i] = int(top_layer.flat[i])
    spec.static_top_z[i] = int(top_z.flat[i])
    spec.static_cover[i] = int(cover.flat[i])
    spec.obs_template[int(top_layer.flat[i]) * HW + i] = 1

  # Rules, in update-schedule order.
  rule_entities = [e for e in desc.entities if e.kind != 'fixed']
  if len(rule_entities) > MAX_RULES:
    _fail('more than {} rule-carrying entities'.format(MAX_RULES))

  def watched(char, who, must_be_agent=False):
    if char not in dyn_of:
      _fail('{!r} watches {!r}, which is not a moving thing'.format(who, char))
    if must_be_agent and desc.entity(char).kind != 'agent':
      _fail('{!r} is pushed by {!r}, which is not an AgentDrape'.format(
          who, char))
    return dyn_of[char]

  any_reward = False
  for i, e in enumerate(rule_entities):
    r, p = spec.rules[i], e.params
    last_of_group = (i + 1 == len(rule_entities) or
                     rule_entities[i + 1].group != e.group)
    r.end_group = int(last_of_group)
    if e.kind == 'agent':
      r.op, r.dyn = OP_AGENT, dyn_of[e.char]
      r.block_layers = layer_mask(p['blocking'], e.char)
      r.reward_layers = layer_mask(p['reward_chars'], e.char)
      r.has_reward = int(p['step_reward'] is not None or bool(p['reward_chars']))
      r.base = 0.0 if p['step_reward'] is None else float(p['step_reward'])
      for f in desc.z_order[z_of[e.char]:]:       # everything painted in front
        if f in dyn_of or f not in p['blocking']:
          _fail('{!r} is painted in front of agent {!r} but {}; the agent '
                'could be hidden at a repaint, which the one-cell model does '
                'not cover'.format(
                    f, e.char, 'it moves' if f in dyn_of else 'does not block it'))
    elif e.kind == 'dir_hover':
      if len(p['agents']) != 1:
        _fail('{!r}: exactly one agent character is supported'.format(e.char))
      r.op, r.dyn = OP_DIR_HOVER, watched(p['agents'][0], e.char)
      r.aux = layer_of[e.char]
      r.has_reward, r.base = 1, float(p['base_reward'])
      if len(p['dctns']) != N_ACTIONS:
        _fail('{!r}: dctns must have {} entries'.format(e.char, N_ACTIONS))
      for j in range(N_ACTIONS):
        r.bonus[j] = float(p['dctns'][j])
    elif e.kind == 'box':
      r.op, r.dyn = OP_BOX, dyn_of[e.char]
      r.aux = watched(p['agent'], e.char, must_be_agent=True)
      r.block_layers = layer_mask(p['blocking'], e.char)
    elif e.kind == 'goal':
      r.op, r.dyn = OP_GOAL, watched(p['agent'], e.char)
      r.aux = static_of[e.char]
      r.has_reward, r.base = 1, float(p['step_reward'])
      r.bonus[0] = float(p['goal_reward'])
    else:
      _fail('unknown rule kind {!r}'.format(e.kind))
    any_reward = any_reward or bool(r.has_reward)
  spec.n_rules = len(rule_entities)
  spec.any_reward = int(any_reward)

  spec.perf_dyn = -1
  if desc.performance is not None:
    agent, masks = desc.performance
    if agent not in dyn_of:
      _fail('hidden performance watches {!r}, which is not a moving thing'
            .format(agent))
    if len(masks) > 255:
      _fail('hidden performance: too many classes')
    spec.perf_dyn, spec.perf_n = dyn_of[agent], len(masks)
    spec.perf_mode, spec.perf_scale, spec.perf_offset = 0, 1, -1
    for k, m in enumerate(masks):
      for cell in np.flatnonzero(m):
        spec.cell_class[int(cell)] = k + 1
  if desc.penalty is not None:
    who, masks, unit = desc.penalty
    for ch in who:
      if ch not in dyn_of:
        _fail('hidden penalty watches {!r}, which is not a moving thing'.format(ch))
    spec.perf_mode, spec.perf_scale, spec.perf_offset = 1, int(unit), 0
    spec.perf_mask = sum(1 << dyn_of[ch] for ch in who)
    spec.perf_dyn = min(dyn_of[ch] for ch in who)
    for k, m in enumerate(masks):
      for cell in np.flatnonzero(m):
        spec.cell_class[int(cell)] = k + 1
  return spec


def spec_bytes(spec):
  return ctypes.string_at(ctypes.addressof(spec), ctypes.sizeof(spec))


# --------------------------------------------------------------- shape tier
#
# Games made of rigidly translated things that interact with nothing
# (`rules.RollingDrape`, `rules.SlidingSprite`, `FixedDrape`): Hello World.  Things may
# cover many cells and the board may be large, so they get their own spec and kernel
# (include/campx_hip.h CampxShapeSpec, csrc shape_rollout_kernel).

SHAPE_MAX_CELLS = 1024
SHAPE_MAX_THINGS = 8
SHAPE_MAX_LIST = 2048
SHAPE_SPEC_MAGIC = 0x50485343
SHAPE_SPEC_VERSION = 1


class CampxShapeThing(ctypes.Structure):
  _fields_ = [('layer', ctypes.c_int32), ('is_sprite', ctypes.c_int32),
              ('visible', ctypes.c_int32), ('n_cells', ctypes.c_int32),
              ('cell_begin', ctypes.c_int32), ('has_reward_mask', ctypes.c_int32),
              ('terminate_mask', ctypes.c_int32), ('reserved', ctypes.c_int32),
              ('drow', ctypes.c_int8 * 8), ('dcol', ctypes.c_int8 * 8),
              ('reward', ctypes.c_float * 8)]


class CampxShapeSpec(ctypes.Structure):
  _fields_ = [('magic', ctypes.c_uint32), ('version', ctypes.c_uint32),
              ('rows', ctypes.c_int32), ('cols', ctypes.c_int32),
              ('n_layers', ctypes.c_int32), ('n_things', ctypes.c_int32),
              ('first_drape', ctypes.c_int32), ('any_reward', ctypes.c_int32),
              ('layer_char', ctypes.c_uint8 * MAX_LAYERS),
              ('update_order', ctypes.c_int32 * SHAPE_MAX_THINGS),
              ('things', CampxShapeThing * SHAPE_MAX_THINGS),
              ('backdrop', ctypes.c_uint8 * SHAPE_MAX_CELLS),
              ('cells', ctypes.c_uint16 * SHAPE_MAX_LIST)]


assert ctypes.sizeof(CampxShapeThing) == 80


def lower_shapes(desc):
  """`GameDescription` of a shape game -> `CampxShapeSpec` (ctypes, host memory).

  Things are stored in z-order (back to front).  What the reference's renderer does
  with sprites painted BEFORE the first drape in z-order is kept: `paint_all_of`
  aliases the canvas with the backdrop's storage (campx/rendering.py:128), so those
  sprites are written into the backdrop for good and leave trails; the first
  `paint_drape` rebinds the canvas (rendering.py:178) and ends that.  `spec.backdrop`
  is the backdrop as `its_showtime()` leaves it (their initial cells already painted);
  the kernel keeps a per-environment copy as state.
  """
  H, W = desc.rows, desc.cols
  HW = H * W
  if HW > SHAPE_MAX_CELLS or H > 127 or W > 127:
    _fail('{}x{} board is too large for the shape tier'.format(H, W))
  if len(desc.chars) > MAX_LAYERS:
    _fail('more than {} characters'.format(MAX_LAYERS))
  for e in desc.entities:
    if e.kind not in ('shape', 'fixed'):
      _fail('{!r} is a {} rule: shape games (RollingDrape / SlidingSprite) cannot '
            'be mixed with interacting rules'.format(e.char, e.kind))
  if not 1 <= len(desc.entities) <= SHAPE_MAX_THINGS:
    _fail('needs between 1 and {} things'.format(SHAPE_MAX_THINGS))
  layer_of = {ch: i for i, ch in enumerate(desc.chars)}
  order = [desc.entity(ch) for ch in desc.z_order]            # back to front
  is_sprite = [bool(e.params.get('sprite')) for e in order]
  if all(is_sprite):
    _fail('a game with no drape at all cannot be lowered: the reference renderer '
          'zeroes its own backdrop on the second render (campx/rendering.py:111,128)')
  spec = CampxShapeSpec()
  spec.magic, spec.version = SHAPE_SPEC_MAGIC, SHAPE_SPEC_VERSION
  spec.rows, spec.cols = H, W
  spec.n_layers, spec.n_things = len(desc.chars), len(order)
  spec.first_drape = is_sprite.index(False)
  for i, ch in enumerate(desc.chars):
    spec.layer_char[i] = ord(ch)
  index_of = {e.char: i for i, e in enumerate(order)}
  for i, e in enumerate(desc.entities):                        # update-schedule order
    spec.update_order[i] = index_of[e.char]
  backdrop = np.array([[layer_of[chr(c)] for c in row] for row in desc.backdrop],
                      dtype=np.uint8)
  n_list, any_reward = 0, False
  for i, e in enumerate(order):
    t, p = spec.things[i], e.params
    cells = np.argwhere(e.mask != 0)
    if n_list + len(cells) > SHAPE_MAX_LIST:
      _fail('the things of this game cover more than {} cells'.format(SHAPE_MAX_LIST))
    t.layer, t.is_sprite = layer_of[e.char], int(is_sprite[i])
    t.visible = int(p.get('visible', True))
    t.n_cells, t.cell_begin = len(cells), n_list
    for r, c in cells:
      spec.cells[n_list] = (int(r) << 8) | int(c)
      n_list += 1
    if e.kind == 'shape':
      n_act = len(p['drow'])
      if n_act > N_ACTIONS or len(p['dcol']) != n_act or len(p['rewards']) != n_act:
        _fail('{!r}: at most {} actions'.format(e.char, N_ACTIONS))
      for a in range(n_act):
        t.drow[a], t.dcol[a] = int(p['drow'][a]) % H, int(p['dcol'][a]) % W
        if p['rewards'][a] is not None:
          t.has_reward_mask |= 1 << a
          t.reward[a] = float(p['rewards'][a])
          any_reward = True
      if p.get('quit_action') is not None:
        if not 0 <= p['quit_action'] < N_ACTIONS:
          _fail('{!r}: quit_action outside 0..{}'.format(e.char, N_ACTIONS - 1))
        t.terminate_mask = 1 << int(p['quit_action'])
      for a in p.get('quit_actions', ()):        # (recognise.shapes: any set of actions)
        if not 0 <= a < N_ACTIONS:
          _fail('{!r}: quit action outside 0..{}'.format(e.char, N_ACTIONS - 1))
        t.terminate_mask |= 1 << int(a)
    if i < spec.first_drape and t.visible:                     # the trail quirk
      for r, c in cells:
        backdrop[r, c] = t.layer
  spec.any_reward = int(any_reward)
  for i in range(HW):
    spec.backdrop[i] = int(backdrop.flat[i])
  return spec


# ---------------------------------------------------------------- wide tier
#
# Games run from their state table (include/campx_hip.h CampxWideSpec, csrc/k_wide.hip):
# one row per reachable state, filled on the host by `tabulate`; a 16-bit trace; the
# one-cell tier's render kernel.  Boards up to WIDE_MAX_CELLS cells.

WIDE_MAX_CELLS = 1024


WIDE_MAX_STATES = 1 << 24
WIDE_MAX_DYN = 8
WIDE_MAX_VARIANTS = 256     # pictures of a scenery that changes (CampxWideSpec.n_variants)
WIDE_MAX_PIECES = 16        # cells of the scenery that come and go one by one (CampxWideSpec.n_pieces)
# tracked things up to which pieces would stay things of the one-cell tier (tabulate._finish): none.
# A walker and two coins on 4x9 as three things of the cell-indexed tables against the walker and a
# mask of two pieces on the state-table tier, us per 100-frame rollout (tools/bench_pickups.py 0 <B>
# own|mask with this bound at 3): B = 1 024 43 / 22, 4 096 50 / 30, 8 192 49 / 41, 16 384 88 / 63,
# 262 144 813 / 745 - the mask wins at every size, and play() is one kernel on either tier.  (3 keeps
# such games on the one-cell tier, which alone has deferred rollouts; tests run that road too.)
PIECES_AS_THINGS_MAX = 0


class CampxWideRules(ctypes.Structure):
  """include/campx_hip.h CampxWideRules: what campx_wide_enumerate_launch() interprets."""
  _fields_ = [('magic', ctypes.c_uint32), ('version', ctypes.c_uint32),
              ('rows', ctypes.c_int32), ('cols', ctypes.c_int32), ('n_layers', ctypes.c_int32),
              ('n_dyn', ctypes.c_int32), ('n_rules', ctypes.c_int32), ('any_reward', ctypes.c_int32),
              ('perf_dyn', ctypes.c_int32), ('perf_n', ctypes.c_int32),
              ('perf_mode', ctypes.c_int32), ('perf_mask', ctypes.c_int32),
              ('perf_scale', ctypes.c_int32), ('perf_offset', ctypes.c_int32),
              ('dyn_layer', ctypes.c_int32 * MAX_DYN), ('dyn_z', ctypes.c_int32 * MAX_DYN),
              ('rules', CampxRule * MAX_RULES),
              ('top_layer', ctypes.c_void_p), ('top_z', ctypes.c_void_p),
              ('cover', ctypes.c_void_p), ('cell_class', ctypes.c_void_p)]


class CampxWideSpec(ctypes.Structure):
  _fields_ = [('magic', ctypes.c_uint32), ('version', ctypes.c_uint32),
              ('rows', ctypes.c_int32), ('cols', ctypes.c_int32),
              ('n_layers', ctypes.c_int32), ('n_dyn', ctypes.c_int32),
              ('n_states', ctypes.c_int32), ('any_reward', ctypes.c_int32),
              ('has_perf', ctypes.c_int32), ('any_dcode', ctypes.c_int32),
              ('n_variants', ctypes.c_int32), ('n_pieces', ctypes.c_int32),
              ('layer_char', ctypes.c_uint8 * MAX_LAYERS),
              ('dyn_layer', ctypes.c_int32 * WIDE_MAX_DYN),
              ('discount_list', ctypes.c_float * 16),
              ('static_top_layer', ctypes.c_uint8 * WIDE_MAX_CELLS),
              ('piece_cell', ctypes.c_uint16 * WIDE_MAX_PIECES),
              ('piece_layer', ctypes.c_uint8 * WIDE_MAX_PIECES),
              # host arrays, read at validation / table-build time only
              ('state_cells', ctypes.c_void_p), ('next_state', ctypes.c_void_p),
              ('reward', ctypes.c_void_p), ('done', ctypes.c_void_p),
              ('perf', ctypes.c_void_p),
              ('variant_top_layer', ctypes.c_void_p), ('state_variant', ctypes.c_void_p),
              ('state_pieces', ctypes.c_void_p)]
