"""Mask generation with the reference's interface (utils/utils_inpaint.py:67-137), host side / numpy.

Masks are integer data: they are produced as uint8 {0,1} [B,3,H,W] and must be BIT-EXACT with the
reference for the same numpy global RNG state, so the draw order is preserved exactly:
box:    np.random.randint(l,h) x2 (mask_h, mask_w), then randint(margin, maxt), randint(margin, maxl);
random: np.random.uniform(l,h), then np.random.choice(N*N, int(total*prob), replace=False).
"""
from __future__ import annotations

import numpy as np


def random_sq_bbox(shape, mask_shape, image_size=256, margin=(16, 16)):
    """utils_inpaint.py:67-84."""
    B, C, H, W = shape
    h, w = mask_shape
    margin_height, margin_width = margin
    maxt = image_size - margin_height - h
    maxl = image_size - margin_width - w
    t = np.random.randint(margin_height, maxt)
    l = np.random.randint(margin_width, maxl)
    mask = np.ones((B, C, H, W), np.uint8)
    mask[..., t:t + h, l:l + w] = 0
    return mask, t, t + h, l, l + w


class mask_generator:
    def __init__(self, mask_type, mask_len_range=None, mask_prob_range=None, image_size=256, margin=(16, 16)):
        assert mask_type in ['box', 'random', 'both', 'extreme']
        self.mask_type = mask_type
        self.mask_len_range = mask_len_range
        self.mask_prob_range = mask_prob_range
        self.image_size = image_size
        self.margin = margin

    def _retrieve_box(self, shape):
        l, h = self.mask_len_range
        l, h = int(l), int(h)
        mask_h = np.random.randint(l, h)
        mask_w = np.random.randint(l, h)
        return random_sq_bbox(shape, (mask_h, mask_w), self.image_size, self.margin)

    def _retrieve_random(self, shape):
        total = self.image_size ** 2
        l, h = self.mask_prob_range
        prob = np.random.uniform(l, h)
        mask_vec = np.ones(self.image_size * self.image_size, np.uint8)
        samples = np.random.choice(self.image_size * self.image_size, int(total * prob), replace=False)
        mask_vec[samples] = 0
        mask_b = mask_vec.reshape(1, self.image_size, self.image_size)
        return np.ascontiguousarray(np.broadcast_to(mask_b, shape)).astype(np.uint8)

    def __call__(self, img_shape):
        """img_shape: (B, 3, H, W) (the reference passes the image tensor only for its shape/device)."""
        shape = tuple(img_shape.shape) if hasattr(img_shape, "shape") else tuple(img_shape)
        if self.mask_type == 'random':
            return self._retrieve_random(shape)
        if self.mask_type == 'box':
            return self._retrieve_box(shape)[0]
        if self.mask_type == 'extreme':
            return (1 - self._retrieve_box(shape)[0]).astype(np.uint8)
        raise NotImplementedError("mask_type 'both' returns None in the reference as well")
