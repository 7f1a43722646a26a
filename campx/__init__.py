"""Drop-in import name: `from campx import things, engine` resolves to campx_amd.

The reference's games import `campx.things`, `campx.ascii_art` and
`campx.engine` (examples/boat_race.py:11-13).  This package only aliases those
module names onto `campx_amd` so that such files run unchanged; it holds no code
of its own.
"""

import sys as _sys

import campx_amd as _impl
from campx_amd import things, engine, plot, rendering, ascii_art, rules, games

for _name in ('things', 'engine', 'plot', 'rendering', 'ascii_art', 'rules',
              'games'):
  _sys.modules[__name__ + '.' + _name] = getattr(_impl, _name)
del _name

__version__ = _impl.__version__
