"""Overlay of the stock `leann_backend_diskann` package: only the embedding-server module is replaced (see ../README.md).
`extend_path` keeps every other submodule resolving to the stock installation, whose own __init__ (backend registration:
`from . import diskann_backend`) is re-done here when it is importable."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
try:  # same side effect as the stock __init__: registering the backend with leann's registry
    from . import diskann_backend  # noqa: F401
except ImportError:  # stock package not installed: the shim alone is still launchable
    pass
