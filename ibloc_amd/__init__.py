"""Importable alias for the package directory `instance-based-loc_amd/` (a hyphenated directory
name cannot be imported directly).  `import ibloc_amd` resolves sub-modules from that directory."""
import os as _os

_pkg_dir = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "instance-based-loc_amd")
__path__ = [_pkg_dir]
with open(_os.path.join(_pkg_dir, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_pkg_dir, "__init__.py"), "exec"))
