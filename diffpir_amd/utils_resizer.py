"""The reference's down-sampling operator with its call shape (utils/utils_resizer.py:8-52, used as `degrade_op` at main_ddpir.py:294):

    degrade_op = Resizer((batch_size, C, H, W), 1 / config.sf)
    low = degrade_op(x)

Antialiased cubic resampling by 1 / sf along H then W (band tables built in C++, csrc/api.hip `resizer_band`; kernel
`band_resample_kernel`).  Instances are what `utils_model.grad_and_value(operator=...)` recognises as a differentiable operator.
"""
from __future__ import annotations

from .engine import DeviceArray, _ptr


class Resizer:
    def __init__(self, in_shape, scale_factor=None, engine=None):
        inv = 1.0 / float(scale_factor)
        sf = int(round(inv))
        if sf < 1 or abs(inv - sf) > 1e-6:
            raise NotImplementedError("the engine's Resizer down-samples by an integer factor (scale_factor = 1 / sf), as main_ddpir.py:294 does")
        self.in_shape, self.sf, self.engine = tuple(in_shape), sf, engine

    def to(self, device):              # the reference moves the module to the device; engine arrays already live there
        return self

    def __call__(self, x: DeviceArray) -> DeviceArray:
        eng = self.engine or x.engine
        B, C, H, W = x.shape
        out = eng.empty((B, C, H // self.sf, W // self.sf))
        eng._check(eng.lib.dpir_resize_down(eng.h, _ptr(x), out.ptr, self.sf, B, H, W))
        return out
