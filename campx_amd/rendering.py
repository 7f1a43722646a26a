"""Single-environment (generic tier) observation renderer.

Host-side mirror of the reference's occluded renderer
(`campx/rendering.py:29-224`).  The batched path does not use this module: there
the render is fused into the HIP step kernel (`csrc/campx_hip.hip`).

Semantics kept from the reference, because games observe them:

* The `layers[ch]` tensors are created once and *rebound* (`Tensor.set_`) on every
  render (rendering.py:209).  Games keep references to them
  (`the_plot['prev_pos_A'] = layers['A']`, examples/boat_race.py:59), so what they
  hold is always "layer as of the latest render".
* Layers are derived from the flat board, so they are always occluded
  (rendering.py:204-205).
* `paint_all_of` makes the canvas share storage with the backdrop curtain
  (rendering.py:128) and `paint_drape` rebinds the canvas to a fresh tensor
  (rendering.py:178).  Consequently a sprite painted before the first drape in
  z-order is written into the backdrop itself, and a game with no drape at all
  has its backdrop wiped by the next `clear()`.  That is what the reference does,
  so it is what this tier does; the fused tier refuses such games.
* The `board` of every returned `Observation` is the renderer's own canvas
  (rendering.py:217: `.long()` of an int64 tensor is the tensor itself): callers
  who want to keep a frame must copy it.

Deviation: the channel order of `layered_board` is ascending character code.
The reference takes `list(set(keys))` (rendering.py:198), which changes with
PYTHONHASHSEED; compare per character through `layers[ch]`.
"""

import collections

import torch

Observation = collections.namedtuple(
    'Observation', ['board', 'layers', 'layered_board'])


class BaseObservationRenderer(object):
  """Canvas used as clear() -> paint_*() back to front -> render()."""

  def __init__(self, rows, cols, characters):
    self.rows = rows
    self.cols = cols
    self._board = torch.zeros((rows, cols), dtype=torch.int64)
    self._layers = {ch: torch.zeros((rows, cols), dtype=torch.uint8)
                    for ch in characters}
    self._channel_order = sorted(self._layers)
    self._layered_board = torch.zeros(
        (len(self._channel_order), rows, cols), dtype=torch.int64)

  @property
  def channel_order(self):
    """Characters in `layered_board` channel order (build addition)."""
    return list(self._channel_order)

  @property
  def shape(self):
    return self._board.shape

  def _require_known(self, character):
    if character not in self._layers:
      raise ValueError('character {} does not seem to be a valid character for '
                       'this game'.format(str(character)))

  def clear(self):
    # In place on purpose: see the module docstring about backdrop aliasing.
    self._board.mul_(0)

  def paint_all_of(self, curtain):
    self._board.set_(curtain)

  def paint_sprite(self, character, position):
    self._require_known(character)
    self._board[tuple(position)] = ord(character)

  def paint_drape(self, character, curtain):
    self._require_known(character)
    mask = curtain if curtain.dtype == torch.int64 else curtain.long()
    # board - mask*board + mask*code (rendering.py:176-178), regrouped.
    self._board.set_(self._board + mask * (ord(character) - self._board))

  def render(self):
    planes = []
    for ch in self._channel_order:
      self._layers[ch].set_((self._board == ord(ch)).to(torch.uint8))
      planes.append(self._layers[ch])
    self._layered_board = torch.stack(planes).long()
    return Observation(board=self._board,
                       layers=self._layers,
                       layered_board=self._layered_board)
