"""Entity base classes: what paints on a board and carries the game rules.

Mirrors the reference's entity plugin surface (`campx/things.py:58-398`): a game
is one `Backdrop` plus any number of `Sprite`s (one cell each) and `Drape`s (a
0/1 mask each).  The engine calls `update()` on every entity once per frame and
then paints them back-to-front.

Constructor argument order, property names (`curtain`, `palette`, `character`,
`corner`, `position`, `visible`) and the `update()` signatures are the contract
user games are written against, so they are kept exactly:

* Backdrop.update(actions, board, layers, things, the_plot)     (things.py:103)
* Drape.update(actions, board, layers, backdrop, things, the_plot)  (things.py:202)
* Sprite.update(actions, board, layers, backdrop, things, the_plot) (things.py:323)

Curtains handed to drapes are `torch.uint8` (not bool) so that user rules written
for torch 0.3.1 (`1 - layers[c]`, `curtain.set_(b)`) keep working.
"""

import abc
import collections


class Backdrop(object):
  """Background scenery: an [H, W] tensor of character codes painted first.

  The default backdrop never changes (reference `things.py:147-148`).
  """

  def __init__(self, curtain, palette):
    self._curtain = curtain
    self._palette = palette

  def update(self, actions, board, layers, things, the_plot):
    """Static scenery: nothing to do."""

  @property
  def curtain(self):
    return self._curtain

  @property
  def palette(self):
    return self._palette


class Drape(abc.ABC):
  """A 0/1 mask that paints one character wherever the mask is set."""

  def __init__(self, curtain, character):
    self._curtain = curtain
    self._character = character

  @abc.abstractmethod
  def update(self, actions, board, layers, backdrop, things, the_plot):
    """Change `self.curtain` (in place / via `set_`) in response to `actions`.

    `board` and `layers` are the rendering from the most recent repaint;
    `things` holds the *current* state of every sprite and drape, including
    those already updated earlier in this frame (reference `engine.py:200-204`).
    """

  @property
  def character(self):
    return self._character

  @property
  def curtain(self):
    return self._curtain


class Sprite(abc.ABC):
  """A single cell that paints one character at `self.position`."""

  Position = collections.namedtuple('Position', ['row', 'col'])

  def __init__(self, corner, position, character):
    self._corner = corner          # Position(rows, cols) of the board
    self._character = character
    self._position = position      # subclasses replace this by value
    self._visible = True           # subclasses may flip this

  @abc.abstractmethod
  def update(self, actions, board, layers, backdrop, things, the_plot):
    """Replace `self._position` in response to `actions`."""

  @property
  def character(self):
    return self._character

  @property
  def corner(self):
    return self._corner

  @property
  def position(self):
    return self._position

  @property
  def visible(self):
    return self._visible


class FixedDrape(Drape):
  """A drape whose mask never changes (reference `things.py:395-398`)."""

  def update(self, actions, board, layers, backdrop, all_things, the_plot):
    return None
