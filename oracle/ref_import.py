"""Import the LIVE reference modules from /root/reference (build container only).

TEST INFRASTRUCTURE ONLY.  /root/reference does not exist on the GPU box: nothing in
`-m gpu` tests, smoke() or bench.py may call this.  It is used by oracle/gen_golden.py
(to produce tests/golden/*.npz) and by tests that are skipped when the tree is absent.

The reference's hot-path modules import cv2 / torchvision / motionblur / hdf5storage /
lpips at module scope but never call them on the path, so empty
stub modules are registered before import.
"""
from __future__ import annotations

import os
import sys
import types

REF_ROOT = os.environ.get("DIFFPIR_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "guided_diffusion"))


def _stub(name, **attrs):
    if name in sys.modules:
        return sys.modules[name]
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


_loaded = {}


def load():
    """Returns a namespace with the live reference modules on the path."""
    if _loaded:
        return _loaded["ns"]
    if not available():
        raise RuntimeError(f"reference tree not found at {REF_ROOT}")
    sys.dont_write_bytecode = True          # the tree is read-only
    try:
        import cv2  # noqa: F401
    except Exception:
        _stub("cv2")
    try:
        import torchvision  # noqa: F401
    except Exception:
        tv = _stub("torchvision")
        tv.utils = _stub("torchvision.utils", make_grid=None)
    mb = _stub("motionblur")
    mb.motionblur = _stub("motionblur.motionblur", Kernel=None)
    _stub("hdf5storage")
    _stub("lpips")
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    from guided_diffusion import script_util, gaussian_diffusion, unet, nn as gd_nn  # type: ignore
    from utils import utils_model, utils_sisr, utils_resizer, utils_inpaint, utils_image  # type: ignore
    ns = types.SimpleNamespace(script_util=script_util, gaussian_diffusion=gaussian_diffusion,
                               unet=unet, gd_nn=gd_nn, utils_model=utils_model, utils_sisr=utils_sisr,
                               utils_resizer=utils_resizer, utils_inpaint=utils_inpaint,
                               utils_image=utils_image)
    _loaded["ns"] = ns
    return ns


def build_reference_model(model_config: dict):
    """main_ddpir.py:219-240 verbatim call sequence -> (model.eval(), diffusion)."""
    ns = load()
    args = ns.utils_model.create_argparser(model_config).parse_args([])
    model, diffusion = ns.script_util.create_model_and_diffusion(
        **ns.script_util.args_to_dict(args, ns.script_util.model_and_diffusion_defaults().keys()))
    model.eval()
    for _, v in model.named_parameters():
        v.requires_grad = False
    return model, diffusion
