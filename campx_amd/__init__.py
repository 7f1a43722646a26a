"""campx_amd: MI355X-native batched grid-world engine with the CampX / PyColab API.

`things`, `engine`, `plot`, `rendering`, `ascii_art` mirror the reference's
modules of the same names (reference `campx/__init__.py:1-2` imports `things`
and `engine` eagerly; so does this package).  `rules` holds the declarative rule
classes the fused HIP tier can lower, `fused` the GameSpec compiler and the
ctypes binding to `csrc/libcampx_hip.so` (imported lazily: it needs the built
library).
"""

from . import things
from . import engine

__all__ = ['things', 'engine', 'plot', 'rendering', 'ascii_art', 'rules',
           'games']
__version__ = '0.1.0'
