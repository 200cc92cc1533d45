"""Host-side mirror of the reference's `creste` package for the perception->costmap->IRL hot path.

Same module paths, class names, constructor arguments, forward signatures, output-dict keys and
state_dict key names as ut-amrl/creste_public's `creste.models.*` / `creste.utils.loss_utils`, so the
reference's train_ssc.py / train_traversability.py / scripts/runtime/compile.py can import it in
place of their own (see `creste_public_amd.install_as_creste`).  The arithmetic runs in
libcreste_hip.so (hand-written gfx950 kernels) -- there is no PyTorch/CPU fallback.
"""
