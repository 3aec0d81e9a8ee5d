"""Debug aid: make every torch.empty / empty_like / new_empty of a floating dtype start as NaN (and integer ones as a large garbage value), so that a
kernel reading a global-memory byte nobody wrote shows up as a non-finite result in ONE process.  import tools.nan_fill_empty; it patches torch."""
import torch

_empty, _empty_like, _new_empty = torch.empty, torch.empty_like, torch.Tensor.new_empty


def _poison(t):
    if t.is_cuda and t.numel():
        if t.is_floating_point():
            t.fill_(float("nan"))
        elif t.dtype in (torch.int32, torch.int64, torch.uint8, torch.int16):
            t.fill_(113 if t.dtype == torch.uint8 else 12345)
    return t


def empty(*a, **kw):
    return _poison(_empty(*a, **kw))


def empty_like(*a, **kw):
    return _poison(_empty_like(*a, **kw))


def new_empty(self, *a, **kw):
    return _poison(_new_empty(self, *a, **kw))


torch.empty, torch.empty_like, torch.Tensor.new_empty = empty, empty_like, new_empty
