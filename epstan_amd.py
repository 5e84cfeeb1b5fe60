"""Import shim: the package directory is `ep-stan_amd/` (not a valid Python
identifier), so `import epstan_amd` maps onto it."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), 'ep-stan_amd')]
with open(_os.path.join(__path__[0], '__init__.py')) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], '__init__.py'), 'exec'))
del _f, _os
