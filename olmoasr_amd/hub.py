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


def dims_of(obj) -> ModelDimensions:
    """``ckpt["dims"]`` in any of its forms: a dict (inference checkpoints, gen_inf_ckpt.py), the reference's dataclass
    instance, or the SimpleNamespace this implementation writes (scripts/training/train_timestamps.py::build_checkpoint)."""
    fields = obj if isinstance(obj, dict) else obj.__dict__
    return ModelDimensions(**{k: int(fields[k]) for k in ModelDimensions.__dataclass_fields__})


def load_checkpoint(path, map_location="cpu") -> dict:
    """``torch.load`` of a checkpoint written by either side.  The reference pickles ``dims`` as an instance of
    ``olmoasr.config.model_dims.ModelDimensions`` (train_timestamps.py:944): when the reference package is not importable,
    that module path is aliased to this package's identical dataclass for the duration of the load."""
    import importlib
    import sys
    import types
    names = ("olmoasr", "olmoasr.config", "olmoasr.config.model_dims")
    added = []
    try:
        try:
            importlib.import_module("olmoasr.config.model_dims")
        except Exception:
            from .config import model_dims as md
            for nm in names:
                if nm not in sys.modules:
                    m = types.ModuleType(nm)
                    m.__path__ = []
                    sys.modules[nm] = m
                    added.append(nm)
            sys.modules["olmoasr.config.model_dims"].ModelDimensions = md.ModelDimensions
        return torch.load(path, map_location=map_location, weights_only=False)
    finally:
        for nm in added:
            sys.modules.pop(nm, None)


def gen_inf_ckpt(checkpoint: dict) -> dict:
    """Training checkpoint -> inference checkpoint (scripts/eval/gen_inf_ckpt.py:4-11): strip the pad row of the token
    embedding, dict-ify dims."""
    sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in checkpoint["model_state_dict"].items()}
    sd["decoder.token_embedding.weight"] = sd["decoder.token_embedding.weight"][:-1, :]
    return {"model_state_dict": sd, "dims": dict(dims_of(checkpoint["dims"]).__dict__)}


def load_model(name: str, device: Optional[Union[str, torch.device]] = None, download_root: Optional[str] = None,
               inference: bool = False, in_memory: bool = False):
    from .model import OLMoASR
    if device is None:  # olmoasr/__init__.py:127-128 (the native model itself refuses a CPU device, loudly: there is no CPU fallback)
        device = "cuda" if torch.cuda.is_available() else "cpu"
    if name in MODEL2LINK:
        root = Path(download_root).expanduser() if download_root else Path.home() / ".cache" / "olmoasr"
        path = root / f"OLMoASR-{name}.pt"
        if not path.is_file():  # olmoasr/__init__.py:44-94: fetch into the cache directory, drop a partial file on failure
            import urllib.request
            root.mkdir(parents=True, exist_ok=True)
            try:
                urllib.request.urlretrieve(MODEL2LINK[name], path)
            except Exception as e:
                if path.exists():
                    path.unlink()
                raise RuntimeError(f"{path} not found and downloading {MODEL2LINK[name]} failed: {e}") from e
    elif os.path.isfile(name):
        path = Path(name)
    else:  # olmoasr/__init__.py:135-138
        raise ValueError(f"Model '{name}' not found. Available models: {list(MODEL2LINK.keys())}")
    if in_memory:  # olmoasr/__init__.py:141-151: the file is read into host memory first, torch.load parses the bytes
        import io
        with open(path, "rb") as f:
            blob = f.read()
        checkpoint = load_checkpoint(io.BytesIO(blob))
        del blob
    else:
        checkpoint = load_checkpoint(path)
    dims = dims_of(checkpoint["dims"])
    sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in checkpoint["model_state_dict"].items()}
    rows = sd["decoder.token_embedding.weight"].shape[0]
    if inference and rows == dims.n_vocab + 1:  # a training checkpoint handed to the inference loader
        sd["decoder.token_embedding.weight"] = sd["decoder.token_embedding.weight"][:-1, :]
    model = OLMoASR(dims, device=device, inference=inference or rows == dims.n_vocab)
    model.load_state_dict(sd)
    return model
