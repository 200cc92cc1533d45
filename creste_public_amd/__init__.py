"""creste_public_amd -- MI355X-native (gfx950) implementation of CREStE's perception->costmap hot path.

    from creste_public_amd import TerrainNet, MaxEntIRL, LossManager, terrainnet_cfg, maxent_irl_cfg

`creste_public_amd.creste` mirrors the reference's `creste` package paths; `install_as_creste()`
registers it under the name `creste` so the reference's own scripts import it unchanged.
All arithmetic of the path runs in libcreste_hip.so (include/creste_hip.h); there is no CPU or
stock-PyTorch fallback -- ops raise `HipLibraryError` when the library or a GPU tensor is missing.
"""
import importlib
import os
import sys

# More hardware queues for this process's streams (read by the HIP runtime when it starts; default 4): the side streams of the
# pipelined inference / the weight gradients / the IRL prefetch only help on a queue of their own (ops.concurrent_stream
# measures whether they got one).  Only takes effect when the package is imported before the first GPU call.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .config import Cfg, as_cfg, maxent_irl_cfg, terrainnet_cfg  # noqa: F401
from ._lib import HipLibraryError  # noqa: F401


def set_precision(name: str):
    from . import hipnn
    hipnn.set_precision(name)


def get_precision() -> str:
    from . import hipnn
    return hipnn.get_precision()



def invalidate_caches():
    """Drop packed-weight / folded-BN caches after parameter updates made through `.data` (see hipnn)."""
    from . import hipnn
    hipnn.invalidate_caches()


_MIRROR = ["creste", "creste.models", "creste.models.blocks", "creste.models.blocks.conv",
           "creste.models.blocks.effnet", "creste.models.blocks.inpainting",
           "creste.models.blocks.splat_projection", "creste.models.blocks.vin",
           "creste.models.vision_encoder", "creste.models.depth", "creste.models.distillation",
           "creste.models.terrainnet", "creste.models.lfd", "creste.utils", "creste.utils.train_utils",
           "creste.utils.depth_utils", "creste.utils.loss_utils", "creste.utils.projection"]


def install_as_creste():
    """Make `import creste.models.terrainnet` etc. resolve to the HIP-backed mirror."""
    for name in _MIRROR:
        sys.modules[name] = importlib.import_module(f"{__name__}.{name}")
    return sys.modules["creste"]


def __getattr__(name):          # lazy: importing the package must not import torch-heavy modules twice
    table = {"TerrainNet": "creste.models.terrainnet", "MaxEntIRL": "creste.models.lfd",
             "DistillationBackbone": "creste.models.distillation", "DepthCompletion": "creste.models.depth",
             "VIN": "creste.models.blocks.vin", "Camera2MapMulti": "creste.models.blocks.splat_projection",
             "MultiScaleFCN": "creste.models.blocks.conv", "LossManager": "creste.utils.loss_utils",
             "MaxEntIRLLoss": "creste.utils.loss_utils"}
    if name in table:
        return getattr(importlib.import_module(f"{__name__}.{table[name]}"), name)
    raise AttributeError(name)
