"""Discovery shim: LEANN's ``autodiscover_backends()`` imports every installed distribution named
``leann-backend-*`` as module ``leann_backend_*`` (packages/leann-core/src/leann/registry.py:30-47).
The distribution ``leann-backend-mi355x`` therefore ships this module, which pulls in ``leann_amd``
and with it the ``@register_backend("mi355x")`` class."""

from leann_amd import Mi355xBackend, Mi355xBuilder, Mi355xSearcher  # noqa: F401
