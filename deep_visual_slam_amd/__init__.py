"""Import shim: the product package lives in ``deep-visual-slam_amd/`` (a hyphen is not a
legal Python identifier), so this importable name forwards to it.  Nothing else lives here."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "deep-visual-slam_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
