"""Namespace shim: modules present here shadow the reference's, everything else falls through to the
reference's own package of the same name further down sys.path."""
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_name = os.path.basename(_here)
for _p in sys.path:
    _cand = os.path.join(_p, _name)
    if _p and os.path.isdir(_cand) and os.path.abspath(_cand) != _here and _cand not in __path__:
        __path__.append(_cand)
