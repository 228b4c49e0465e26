"""streamvoiceanon_amd -- MI355X-native engine for StreamVoiceAnon's chunk-by-chunk
``infer_arvc`` hot path (content encoder -> dual-AR transformer -> Firefly vocoder).

The product path is the HIP library under ``csrc/`` reached through the C ABI declared in
``include/sva.h``; this package holds the host-side mirror of the reference's Python
interface (``InferenceWrapper`` / ``ARVCWrapper``) and the weight packer.
"""

__version__ = "0.1.0"
