"""Ready-made games written with the library rule classes (`campx_amd.rules`).

Each `make_game(batch=None, device=None)` returns what the reference's
`examples/boat_race.py:93-115` returns: `(game, board, reward, discount)` after
`its_showtime()`.  `batch=None` builds the single-environment generic tier;
`batch=B` builds the fused HIP tier.
"""

from . import boat_race, wall_world, sokoban, demos, hello_world

__all__ = ['boat_race', 'wall_world', 'sokoban', 'demos', 'hello_world']
