"""leann_amd -- MI355X (gfx950) native selective-recompute beam search for LEANN.

Importing the package registers the ``"mi355x"`` backend in LEANN's ``BACKEND_REGISTRY``
(packages/leann-core/src/leann/registry.py:16-27).  Compute lives in libleann_mi355x.so (HIP);
there is no CPU fallback.
"""

from .backend import (  # noqa: F401  (registers the backends)
    Mi355xBackend,
    Mi355xBuilder,
    Mi355xDiskannBackend,
    Mi355xDiskannBuilder,
    Mi355xDiskannSearcher,
    Mi355xSearcher,
)

__version__ = "0.2.0"
__all__ = ["Mi355xBackend", "Mi355xBuilder", "Mi355xSearcher", "Mi355xDiskannBackend", "Mi355xDiskannBuilder", "Mi355xDiskannSearcher"]
