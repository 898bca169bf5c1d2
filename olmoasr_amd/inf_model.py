"""``olmoasr.inf_model`` for the MI355X-native engine (reference olmoasr/inf_model.py:405-457).

The reference keeps a second copy of the model code whose only differences on this path are the token embedding without
the pad row (``n_vocab`` rows, inf_model.py:302 -- what scripts/eval/gen_inf_ckpt.py:4-11 writes) and the manual
``qkv_attention`` in place of SDPA (inf_model.py:172-196, same mathematics).  Here it is the same engine in its inference
layout: ``OLMoASR(dims)`` == ``olmoasr_amd.model.OLMoASR(dims, inference=True)``.
"""
from .config.model_dims import ModelDimensions  # noqa: F401
from .model import (AudioEncoder, Conv1d, LayerNorm, Linear, MultiHeadAttention, ResidualAttentionBlock, TextDecoder,  # noqa: F401
                    sinusoids)
from .model import OLMoASR as _OLMoASR


class OLMoASR(_OLMoASR):
    def __init__(self, dims: ModelDimensions, device=None, seed=None, compute_dtype="bfloat16"):
        super().__init__(dims, device=device, seed=seed, inference=True, compute_dtype=compute_dtype)
