"""MI355X-native caption-decoder hot path of sonniki/image-captioning-with-external-knowledge.

Import as `ick_amd` (see /ick_amd.py).  Sub-modules:
  synth      seeded synthetic workloads (CPU tensors)
  build      hipcc recipe for csrc/ -> libick_amd.so (gfx950)
  lib        ctypes binding of the C ABI declared in include/ick_amd.h
  ops        per-op Python wrappers (device pointers from torch tensors)
  decoder    DecoderTransformer / Encoder engine shared by the three variants
  geo_aware/models.py, knowledge_aware/models.py, news_knowledge_aware/models.py
             drop-in replacements for the reference's per-variant `models` module
"""
__version__ = "0.1.0"


def load_models(variant):
    """Return the drop-in `models` module of a variant ("geo" | "knowledge" | "news")."""
    import importlib
    name = {"geo": "geo_aware", "knowledge": "knowledge_aware", "news": "news_knowledge_aware"}[variant]
    return importlib.import_module("ick_amd.%s.models" % name)
