"""Small helpers the reference takes from timm 0.4.5 (timm.models.layers), restated so the package has no timm dependency."""
import torch.nn as nn


def to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


class DropPath(nn.Module):
    """Stochastic depth.  On the HIP path the mask is drawn by the owning block and applied inside a GEMM epilogue
    (row_scale), so this module only carries `drop_prob`; calling it directly applies the usual per-sample mask."""

    def __init__(self, drop_prob=0.):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0. or not self.training:
            return x
        keep = 1 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep).div_(keep)
        return x * mask

    def extra_repr(self):
        return f"drop_prob={self.drop_prob}"
