"""Model dimension table of the reference (olmoasr/config/model_dims.py:4-89): field names and variant keys are part
of the checkpoint format (``ckpt["dims"]``, olmoasr/__init__.py:156) and of the CLI (``--model_variant``)."""
from dataclasses import dataclass


@dataclass
class ModelDimensions:
    n_mels: int
    n_audio_ctx: int
    n_audio_state: int
    n_audio_head: int
    n_audio_layer: int
    n_vocab: int
    n_text_ctx: int
    n_text_state: int
    n_text_head: int
    n_text_layer: int


def _variant(width: int, heads: int, layers: int) -> ModelDimensions:
    return ModelDimensions(n_mels=80, n_audio_ctx=1500, n_audio_state=width, n_audio_head=heads, n_audio_layer=layers,
                           n_vocab=51864, n_text_ctx=448, n_text_state=width, n_text_head=heads, n_text_layer=layers)


VARIANT_TO_DIMS = {
    "tiny": _variant(384, 6, 4),
    "base": _variant(512, 8, 6),
    "small": _variant(768, 12, 12),
    "medium": _variant(1024, 16, 24),
    "large": _variant(1280, 20, 32),
}
