"""endodav_amd — MI355X-native (gfx950) implementation of EndoDAV's per-clip forward.

Drop-in for ``import models.endodav as endodav`` (reference ``models/endodav/__init__.py:1-2``):
``endodav_amd.endodav(**kwargs)`` and ``endodav_amd.mark_only_part_as_trainable``.
"""
from .trainable import mark_only_part_as_trainable  # noqa: F401
from .endodav import endodav  # noqa: F401

__all__ = ["endodav", "mark_only_part_as_trainable"]
