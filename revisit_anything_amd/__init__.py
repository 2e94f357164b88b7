"""Import alias.  The package directory required by the project layout is ``revisit-anything_amd/``
(not a valid Python identifier); this shim makes it importable as ``revisit_anything_amd``."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "revisit-anything_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
