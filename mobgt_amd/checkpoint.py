"""Weights of a reference-trained model.

The reference's `Graphormer` IS its LightningModule, so a Lightning `.ckpt` is a pickle whose `"state_dict"`
holds exactly the parameter names this repo keeps (`entry.py:71-93` restores it with
`Graphormer.load_from_checkpoint(path, strict=False, **hparams)`).  Loading copies INTO the existing
parameters, so the fused QKV storage of `MultiHeadAttention.fuse_qkv_storage` (the reference-named
`linear_q/k/v` parameters are views of one `[3C, C]` buffer) stays intact.
"""
import torch


def lightning_state_dict(ckpt):
    """The model state_dict inside a Lightning checkpoint dict (or the dict itself if it already is one)."""
    if isinstance(ckpt, dict) and "state_dict" in ckpt and isinstance(ckpt["state_dict"], dict):
        return ckpt["state_dict"]
    return ckpt


def load_lightning_checkpoint(model, path_or_dict, strict=False):
    """`Graphormer.load_from_checkpoint(path, strict=False, ...)` of entry.py:71-93 for an already constructed
    model: returns torch's (missing_keys, unexpected_keys).  Buffers the reference does not persist
    (adjacency products, bin tables) are never expected."""
    ckpt = path_or_dict
    if not isinstance(ckpt, dict):
        ckpt = torch.load(path_or_dict, map_location="cpu", weights_only=False)
    sd = {k: v for k, v in lightning_state_dict(ckpt).items() if torch.is_tensor(v)}
    own = model.state_dict()
    bad = [k for k, v in sd.items() if k in own and tuple(own[k].shape) != tuple(v.shape)]
    if bad:
        raise ValueError("checkpoint tensors with a different shape than the model's: " + ", ".join(bad[:8]))
    res = model.load_state_dict(sd, strict=strict)
    # stand-alone models re-derive their bf16 shadow weights every forward; shadows owned by a train.TrainStep built
    # BEFORE this load are refreshed here (its optimizer kernel only rewrites them at the next step)
    from .model import sync_external_shadows
    sync_external_shadows(model)
    return res


def save_lightning_checkpoint(model, path, **extra):
    """Write `{"state_dict": model.state_dict(), **extra}` -- the part of a Lightning checkpoint the reference reads."""
    torch.save({"state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()}, **extra}, path)
