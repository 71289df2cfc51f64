"""`pytorch3d` as far as the Next3D generator forward needs it (next3d_amd/shims/__init__.py)."""
from . import io, renderer, structures      # noqa: F401

__version__ = '0.0-n3d-shim'
