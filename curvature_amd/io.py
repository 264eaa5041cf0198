"""Name-keyed (de)serialisation of estimator state.

The reference pickles ``est.state`` as it is: a dict keyed by ``nn.Module`` *objects*
(``scripts/factors.py:122-129``), which only loads back into a process that rebuilds identical module
objects and silently stops matching the model it is assigned to (``scripts/evaluate.py:348-370`` indexes
the loaded dict with the NEW model's modules).  Here a state is stored under the layers' qualified names in
``model.named_modules()`` order - the order ``add[i]`` / ``multiply[i]`` and every per-layer list of the API
refer to - and loaded back by name; module-keyed dicts (the reference's format, or ``est.state`` itself) are
accepted too and matched by position.  SURVEY.md section 8(f), rank 2.
"""
from typing import Any, Dict, Union

import torch

FORMAT = "curvature_amd.state.v1"


def _selected(est) -> Dict[torch.nn.Module, str]:
    names = {}
    for name, mod in est.model.named_modules():
        if mod.__class__.__name__ in est.layer_types:
            if mod.__class__.__name__ == 'MultiheadAttention' and getattr(est, "_mha_as_projections", False):
                from .curvatures import AttentionProjection       # KFAC / EFB / INF: the two projections are the layers
                for proj in AttentionProjection.of(mod):
                    names[proj] = (name + "." if name else "") + proj.kind
            else:
                names[mod] = name
    return names


def _to_cpu(v: Any) -> Any:
    if isinstance(v, torch.Tensor):
        return v.detach().cpu()
    if isinstance(v, (list, tuple)):
        return type(v)(_to_cpu(x) for x in v)
    return v


def _to_device(v: Any, device) -> Any:
    if isinstance(v, torch.Tensor):
        return v.to(device)
    if isinstance(v, (list, tuple)):
        return type(v)(_to_device(x, device) for x in v)
    return v


def named_state(est, attr: str = "state") -> Dict[str, Any]:
    """``getattr(est, attr)`` (a module-keyed dict) re-keyed by qualified layer name, tensors on the CPU."""
    names = _selected(est)
    src = getattr(est, attr)
    # string keys (Diagonal's 'attn_in' / 'attn_out' of MultiheadAttention modules) are names already
    return {(layer if isinstance(layer, str) else names[layer]): _to_cpu(value) for layer, value in src.items()}


def save_state(est, path: str, attrs=("state",)) -> None:
    """Write the given dict attributes of an estimator (``state``, ``inv_state``, EFB ``diags``, ...)."""
    payload = {"format": FORMAT, "estimator": est.__class__.__name__,
               "layers": [n for n in _selected(est).values()],
               "attrs": {a: named_state(est, a) for a in attrs}}
    torch.save(payload, path)


def load_state(est, source: Union[str, Dict], attr: str = "state", device=None):
    """Assign ``est.<attr>`` from a file written by `save_state`, from a name-keyed dict, or from a
    module-keyed dict of another model instance (the reference's format: matched by position in
    ``modules()`` order).  Returns the estimator."""
    if isinstance(source, str):
        source = torch.load(source, map_location="cpu")
    if isinstance(source, dict) and source.get("format") == FORMAT:
        source = source["attrs"][attr]
    names = _selected(est)
    if device is None:
        device = next(est.model.parameters()).device
    out = {}
    if all(isinstance(k, str) for k in source.keys()):
        attn = ('attn_in', 'attn_out')
        missing = [n for n in source if n not in set(names.values()) and n not in attn]
        if missing:
            raise KeyError(f"state holds layers the model does not have: {missing[:3]}")
        by_name = {name: layer for layer, name in names.items()}
        for name, value in source.items():          # stored order = modules() order of the writer
            out[name if name in attn and name not in by_name else by_name[name]] = _to_device(value, device)
    else:
        layers = list(names.keys())
        if len(source) != len(layers):
            raise ValueError(f"module-keyed state has {len(source)} entries, the model selects {len(layers)} layers")
        for layer, value in zip(layers, source.values()):
            out[layer] = _to_device(value, device)
    setattr(est, attr, out)
    return est
