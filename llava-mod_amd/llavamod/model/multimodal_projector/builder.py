"""mm_projector (reference multimodal_projector/builder.py:26-66,125-149): the `mlpNx_gelu` image projector
the qwen shells select (`--image_projector_type mlp2x_gelu`).  Parameter path
`mm_projector.image_spatial_proj.{0,2}.{weight,bias}` as in the reference.  qformer / pool / simple /
video projectors are alternative types never selected on the distillation path."""
import re
from types import SimpleNamespace

import torch
import torch.nn as nn

from ... import ops
from ...ops import FusedWeight

BF16 = torch.bfloat16


class _Lin(nn.Module):
    def __init__(self, i, o, device):
        super().__init__()
        self.weight = nn.Parameter(torch.empty((o, i), device=device, dtype=BF16))
        self.bias = nn.Parameter(torch.empty(o, device=device, dtype=BF16))


class _GELU(nn.Module):
    pass


def build_image_projector(config, delay_load=False, device="cuda", **kwargs):
    projector_type = getattr(config, "image_projector_type", "linear")
    m = re.match(r"^mlp(\d+)x_gelu$", projector_type or "")
    if not m or int(m.group(1)) != 2:
        raise NotImplementedError(f"image_projector_type={projector_type!r}: only mlp2x_gelu is on the distillation path")
    return nn.Sequential(_Lin(config.mm_hidden_size, config.hidden_size, device), _GELU(),
                         _Lin(config.hidden_size, config.hidden_size, device))


class build_projector(nn.Module):
    def __init__(self, config, delay_load=False, device="cuda", **kwargs):
        super().__init__()
        self.image_spatial_proj = build_image_projector(config, device=device)
        p = self.image_spatial_proj
        self._fc1 = FusedWeight([[p[0].weight]], [[p[0].bias]])
        self._fc2 = FusedWeight([[p[2].weight]], [[p[2].bias]])

    def fused_weights(self):
        return [self._fc1, self._fc2]

    def forward_image(self, tower_out):
        """tower_out: [B, 1+P, Dv] (CLS row still present, skipped by stride) -> [B*P, H]."""
        spec = SimpleNamespace(fc1=self._fc1.ensure(), fc2=self._fc2.ensure())
        return ops.ProjectorBlock.apply(tower_out, spec, *[q for q in self.parameters() if q.requires_grad])
