"""Binding to LEANN's plugin boundary.

When ``leann-core`` is importable (the normal deployment: ``pip install leann-core
leann-backend-mi355x``) the backend registers itself in ``leann.registry.BACKEND_REGISTRY`` and
derives from ``leann.interface``'s ABCs (packages/leann-core/src/leann/interface.py:7-107,
registry.py:16-27).  On a box without leann-core (the GPU test box) structurally identical local
ABCs and a local registry are used, so the backend can be driven and tested through the very same
methods.
"""

from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Any, Literal, Optional

import numpy as np

try:  # pragma: no cover - depends on the environment
    from leann.interface import (  # type: ignore
        LeannBackendBuilderInterface,
        LeannBackendFactoryInterface,
        LeannBackendSearcherInterface,
    )
    from leann.registry import BACKEND_REGISTRY, register_backend  # type: ignore

    HAVE_LEANN_CORE = True
except Exception:  # noqa: BLE001
    HAVE_LEANN_CORE = False
    BACKEND_REGISTRY: dict = {}

    def register_backend(name: str):
        def decorator(cls):
            BACKEND_REGISTRY[name] = cls
            return cls

        return decorator

    class LeannBackendBuilderInterface(ABC):
        @abstractmethod
        def build(self, data: np.ndarray, ids: list, index_path: str, **kwargs) -> None: ...

    class LeannBackendSearcherInterface(ABC):
        @abstractmethod
        def __init__(self, index_path: str, **kwargs): ...

        @abstractmethod
        def _ensure_server_running(self, passages_source_file: str, port: Optional[int], **kwargs) -> int: ...

        @abstractmethod
        def search(self, query: np.ndarray, top_k: int, complexity: int = 64, beam_width: int = 1,
                   prune_ratio: float = 0.0, recompute_embeddings: bool = False,
                   pruning_strategy: Literal["global", "local", "proportional"] = "global",
                   zmq_port: Optional[int] = None, **kwargs) -> dict[str, Any]: ...

        @abstractmethod
        def compute_query_embedding(self, query: str, use_server_if_available: bool = True,
                                    zmq_port: Optional[int] = None) -> np.ndarray: ...

    class LeannBackendFactoryInterface(ABC):
        @staticmethod
        @abstractmethod
        def builder(**kwargs) -> LeannBackendBuilderInterface: ...

        @staticmethod
        @abstractmethod
        def searcher(index_path: str, **kwargs) -> LeannBackendSearcherInterface: ...
