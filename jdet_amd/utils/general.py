"""Helpers the heads use.  Mirrors python/jdet/utils/general.py: multi_apply L50-53, unmap L55-65,
parse_losses L67-79, sync L30-48 (mpi_all_reduce -> torch.distributed all_reduce)."""
from functools import partial

import torch


def multi_apply(func, *args, **kwargs):
    pfunc = partial(func, **kwargs) if kwargs else func
    map_results = map(pfunc, *args)
    return tuple(map(list, zip(*map_results)))


def unmap(data, count, inds, fill=0):
    """Unmap a subset of items (data) back to the original set of items (of size count)."""
    if data.dim() == 1:
        ret = torch.full((count,), fill, dtype=data.dtype, device=data.device)
        ret[inds] = data
    else:
        ret = torch.full((count,) + tuple(data.shape[1:]), fill, dtype=data.dtype, device=data.device)
        ret[inds, :] = data
    return ret


def parse_losses(losses):
    _losses = dict()
    for loss_name, loss_value in losses.items():
        if isinstance(loss_value, torch.Tensor):
            _losses[loss_name] = loss_value.mean()
        elif isinstance(loss_value, list):
            if all(_loss.dim() == 0 for _loss in loss_value):   # per-level scalars: one stack + one sum
                _losses[loss_name] = torch.stack(loss_value).sum()
            else:
                _losses[loss_name] = sum(_loss.mean() for _loss in loss_value)
        else:
            raise TypeError("{} is not a tensor or list of tensors".format(loss_name))
    total_loss = sum(_value for _key, _value in _losses.items() if "loss" in _key)
    return total_loss, _losses


def sync(data, reduce_mode="mean", to_numpy=True):
    """all-reduce (when running under torch.distributed) and convert to numpy, as general.py:L30-48"""
    import numpy as np
    import torch.distributed as dist

    def _sync(d):
        if isinstance(d, (list, tuple)):
            return [_sync(x) for x in d]
        if isinstance(d, dict):
            return {k: _sync(v) for k, v in d.items()}
        if isinstance(d, torch.Tensor):
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                d = d.detach().clone()
                dist.all_reduce(d, op=dist.ReduceOp.SUM)
                if reduce_mode == "mean":
                    d = d / dist.get_world_size()
            return d.detach().cpu().numpy() if to_numpy else d
        if not isinstance(d, (int, float, str, np.ndarray)):
            raise ValueError(f"{type(d)} is not supported")
        return d

    return _sync(data)


_CONST_CACHE = {}


def const_like(values, like):
    """small constant tensor on `like`'s device / dtype, uploaded once and cached: `like.new_tensor([...])` in a step
    is a synchronous host -> device copy (and cannot be captured in a HIP graph)"""
    key = (tuple(float(v) for v in values), like.device, like.dtype)
    t = _CONST_CACHE.get(key)
    if t is None:
        t = torch.tensor(key[0], dtype=like.dtype, device=like.device)
        _CONST_CACHE[key] = t
    return t
