"""Import alias: the package lives in the directory `pc-gan_amd/` (the name the build
contract asks for), which is not a valid Python identifier.  This shim makes it
importable as `pcgan_amd` by pointing the package search path at that directory.
"""
import os as _os

_ROOT = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
__path__ = [_os.path.join(_ROOT, 'pc-gan_amd')]
with open(_os.path.join(__path__[0], '__init__.py')) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], '__init__.py'), 'exec'))
del _f
