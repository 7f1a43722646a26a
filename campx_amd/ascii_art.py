"""Build an `Engine` from an ASCII-art diagram.

Mirror of the reference's construction API (`campx/ascii_art.py:29-340`):
`ascii_art_to_game(art, what_lies_beneath, sprites, drapes, backdrop,
update_schedule, z_order, occlusion_in_layers)` and `Partial`.  Two keyword
arguments are added at the end, `batch` and `device`; leaving them out gives the
reference's single-environment engine (or the process-wide default of
`engine.set_default_batch`), passing `batch=B` gives the fused HIP tier (see
`engine.Engine`).

Behaviour kept from the reference:

* every class is wrapped in a `Partial` (ascii_art.py:167-172);
* a string `update_schedule` is split into characters and a flat schedule is one
  update group (ascii_art.py:182-186); groups are named '00000', '00001', ...
  (ascii_art.py:255-258);
* default `z_order` is the flattened schedule (ascii_art.py:204); the default schedule itself
  is the entity characters in ascending order (the reference: in `set` order, which varies
  from process to process);
* entities are added in schedule order, each character is then replaced in the
  art by `what_lies_beneath`, and what remains is the backdrop, whose palette is
  the set of characters left (ascii_art.py:266-307);
* a sprite missing from the art starts at (0, 0); a drape missing from the art
  starts empty.
"""

import itertools

import numpy as np
import torch

from . import things
from .engine import Engine, _UNSET

_ART_ERROR = (
    'the argument to ascii_art_to_uint8_nparray must be a list (or tuple) '
    'of strings containing the same number of strictly-ASCII characters.')


def ascii_art_to_long_tensor(art):
  """[H] strings of W ASCII characters -> int64 tensor [H, W] of their codes."""
  if not isinstance(art, (list, tuple)) or not all(
      isinstance(row, str) for row in art):
    hint = ''
    if isinstance(art, (list, tuple)):
      hint = ' Did you pass a list of list of single characters?'
    raise TypeError(_ART_ERROR + hint)
  if not art or len({len(row) for row in art}) != 1:
    raise ValueError(_ART_ERROR)
  try:
    rows = [np.frombuffer(row.encode('ascii'), dtype=np.uint8) for row in art]
  except UnicodeEncodeError as e:
    raise ValueError('{} (original error: {})'.format(_ART_ERROR, e))
  return torch.from_numpy(np.stack(rows).astype(np.int64))


class Partial(object):
  """An entity class plus the extra constructor arguments to build it with."""

  def __init__(self, pycolab_thing, *args, **kwargs):
    if not issubclass(pycolab_thing,
                      (things.Backdrop, things.Sprite, things.Drape)):
      raise TypeError('the pycolab_thing argument to ascii_art.Partial must be '
                      'a Backdrop, Sprite, or Drape subclass.')
    self.pycolab_thing = pycolab_thing
    self.args = args
    self.kwargs = kwargs


def _as_partial(thing):
  return thing if isinstance(thing, Partial) else Partial(thing)


def ascii_art_to_game(art,
                      what_lies_beneath,
                      sprites={},
                      drapes={},
                      backdrop=things.Backdrop,
                      update_schedule=None,
                      z_order=None,
                      occlusion_in_layers=True,
                      batch=_UNSET,
                      device=None):
  """Turn an ASCII-art board plus entity classes into an initialised `Engine`.

  The caller still has to call `its_showtime()`.  Raises `TypeError` /
  `ValueError` for the malformed inputs the reference rejects
  (ascii_art.py:191-247).
  """
  sprites = {ch: _as_partial(cls) for ch, cls in sprites.items()}
  drapes = {ch: _as_partial(cls) for ch, cls in drapes.items()}
  backdrop = _as_partial(backdrop)
  entity_chars = set(sprites) | set(drapes)

  # Normalise the schedule to a list of groups, each a list of characters.
  if update_schedule is None:
    # (the reference takes `list(set)`, ascii_art.py:178, whose order changes with
    # PYTHONHASHSEED; ascending characters is one of those orders, and always the same one)
    update_schedule = sorted(entity_chars)
  if isinstance(update_schedule, str):
    update_schedule = list(update_schedule)
  if all(isinstance(item, str) for item in update_schedule):
    update_schedule = [update_schedule]
  try:
    flat_schedule = list(itertools.chain.from_iterable(update_schedule))
  except TypeError:
    raise TypeError('if any element in update_schedule is an iterable (like a '
                    'list), all elements in update_schedule must be')
  if set(flat_schedule) != entity_chars:
    raise ValueError('if specified, update_schedule must list each sprite and '
                     'drape exactly once.')

  if z_order is None:
    z_order = flat_schedule
  if set(z_order) != entity_chars:
    raise ValueError('if specified, z_order must list each sprite and drape '
                     'exactly once.')

  if isinstance(what_lies_beneath, str) and len(what_lies_beneath) != 1:
    raise ValueError(
        'what_lies_beneath may either be a single-character ASCII string or '
        'a list of ASCII-character strings')
  if entity_chars.intersection(''.join(what_lies_beneath)):
    raise ValueError(
        'any character specified in what_lies_beneath must not be one of the '
        'characters used as keys in the sprites or drapes arguments.')

  art = ascii_art_to_long_tensor(art)
  if isinstance(what_lies_beneath, str):
    beneath = torch.full_like(art, ord(what_lies_beneath))
  else:
    beneath = ascii_art_to_long_tensor(what_lies_beneath)
    if art.shape != beneath.shape:
      raise ValueError(
          'if not a single ASCII character, what_lies_beneath must be ASCII '
          'art whose shape is the same as that of the ASCII art in art.')

  game = Engine(*art.shape, occlusion_in_layers=occlusion_in_layers,
                batch=batch, device=device)

  group_of = {}
  for i, group in enumerate(update_schedule):
    for ch in group:
      group_of[ch] = '{:05d}'.format(i)

  for ch in flat_schedule:
    game.update_group(group_of[ch])
    where = art == ord(ch)
    if ch in drapes:
      p = drapes[ch]
      game.add_prefilled_drape(ch, where.to(torch.uint8), p.pycolab_thing,
                               *p.args, **p.kwargs)
    if ch in sprites:
      cells = where.nonzero()
      if len(cells) > 1:
        raise ValueError('sprite character {} can appear in at most one place '
                         'in art.'.format(ch))
      row, col = (int(cells[0, 0]), int(cells[0, 1])) if len(cells) else (0, 0)
      p = sprites[ch]
      game.add_sprite(ch, (row, col), p.pycolab_thing, *p.args, **p.kwargs)
    art[where] = beneath[where]

  game.set_z_order(z_order)
  game.set_prefilled_backdrop(
      characters=''.join(chr(c) for c in torch.unique(art).tolist()),
      prefill=art,
      backdrop_class=backdrop.pycolab_thing,
      *backdrop.args, **backdrop.kwargs)
  return game
