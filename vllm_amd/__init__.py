"""vllm_amd -- MI355X-native exact similarity search + gathered InfoNCE.

Host-side mirror of the reference's Python entry points over libabcsearch.so.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
