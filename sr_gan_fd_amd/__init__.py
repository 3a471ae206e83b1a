"""Import shim: the package directory is ``sr-gan-fd_amd/`` (hyphenated, as the project layout
names it); Python imports it as ``sr_gan_fd_amd``."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "sr-gan-fd_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
