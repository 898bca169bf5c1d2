"""Checkpoint entry points: ``load_model`` (reference olmoasr/__init__.py:97-166) and the inference-checkpoint
conversion of scripts/eval/gen_inf_ckpt.py:4-11.  The checkpoint format is the reference's:
``{"model_state_dict": {...}, "dims": ModelDimensions | dict, ...}``; DDP checkpoints carry a ``module.`` prefix
(train_timestamps.py:935).  There is no network in this environment, so names resolve to the reference's cache location
(``~/.cache/olmoasr/OLMoASR-{name}.pt``) and must already be on disk."""
import os
from pathlib import Path
from typing import Optional, Union

import torch

from .config.model_dims import ModelDimensions

MODEL2LINK = {  # olmoasr/__init__.py:23-30
    "tiny": "https://huggingface.co/allenai/OLMoASR/resolve/main/models/OLMoASR-tiny.en.pt",
    "base": "https://huggingface.co/allenai/OLMoASR/resolve/main/models/OLMoASR-base.en.pt",
    "small": "https://huggingface.co/allenai/OLMoASR/resolve/main/models/OLMoASR-small.en.pt",
    "medium": "https://huggingface.co/allenai/OLMoASR/resolve/main/models/OLMoASR-medium-v2.en.pt",
    "large": "https://huggingface.co/allenai/OLMoASR/resolve/main/models/OLMoASR-large.en.pt",
    "large-v2": "https://huggingface.co/allenai/OLMoASR/resolve/main/models/OLMoASR-large.en-v2.pt",
}


def gen_inf_ckpt(checkpoint: dict) -> dict:
    """Training checkpoint -> inference checkpoint: strip the pad row of the token embedding, dict-ify dims."""
    sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in checkpoint["model_state_dict"].items()}
    sd["decoder.token_embedding.weight"] = sd["decoder.token_embedding.weight"][:-1, :]
    dims = checkpoint["dims"]
    return {"model_state_dict": sd, "dims": dims if isinstance(dims, dict) else dims.__dict__}


def load_model(name: str, device: Optional[Union[str, torch.device]] = None, download_root: Optional[str] = None,
               inference: bool = False, in_memory: bool = False):
    from .model import OLMoASR
    if device is None:
        device = "cuda"
    if name in MODEL2LINK:
        root = Path(download_root).expanduser() if download_root else Path.home() / ".cache" / "olmoasr"
        path = root / f"OLMoASR-{name}.pt"
        if not path.is_file():
            raise RuntimeError(f"{path} not found and this environment has no network to fetch {MODEL2LINK[name]}")
    elif os.path.isfile(name):
        path = Path(name)
    else:
        raise RuntimeError(f"Model {name} not found; available models = {list(MODEL2LINK)}")
    checkpoint = torch.load(path, map_location="cpu", weights_only=False)
    dims = checkpoint["dims"]
    dims = ModelDimensions(**dims) if isinstance(dims, dict) else ModelDimensions(**dims.__dict__)
    sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in checkpoint["model_state_dict"].items()}
    rows = sd["decoder.token_embedding.weight"].shape[0]
    if inference and rows == dims.n_vocab + 1:  # a training checkpoint handed to the inference loader
        sd["decoder.token_embedding.weight"] = sd["decoder.token_embedding.weight"][:-1, :]
    model = OLMoASR(dims, device=device, inference=inference or rows == dims.n_vocab)
    model.load_state_dict(sd)
    return model
