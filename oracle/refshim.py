"""TEST INFRASTRUCTURE ONLY -- imports the *untouched* reference GRL from /root/reference.

The reference (ofsoundof/GRL-Image-Restoration) is pure Python but depends on five trivial
symbols from packages that are not installed here (timm, fairscale, omegaconf).  This module
registers minimal stand-ins in ``sys.modules`` and then imports ``models.networks.grl.GRL``
from the read-only mount.  None of the stand-ins touches eval-mode arithmetic:

  to_2tuple          used by grl.py:283-284, ops.py:118           (tuple helper)
  trunc_normal_      used by grl.py:153,464                       (init only)
  DropPath           used by mixed_attn_block_efficient.py:500    (identity in eval)
  checkpoint_wrapper used by grl.py:133-134                       (identity)
  OmegaConf.create   used by grl.py:302-308                       (attribute bag)

Only usable where /root/reference exists (the build container).  It is used to
  * pin oracle/grl_oracle.py (our CPU restatement) against the real reference, and
  * generate the golden fixtures under tests/golden/ (oracle/make_golden.py).
Nothing in the product package imports this file.
"""
import os
import sys
import types

import torch.nn as nn

REFERENCE_ROOT = os.environ.get("GRL_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "models", "networks", "grl.py"))


def _to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


class _DropPath(nn.Module):
    """timm semantics (scale_by_keep=True); identity in eval."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1 - self.drop_prob
        m = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        return x * m / keep


def install_shims():
    if "timm.models.layers" not in sys.modules:
        tl = types.ModuleType("timm.models.layers")
        tl.to_2tuple = _to_2tuple
        tl.DropPath = _DropPath
        tl.trunc_normal_ = lambda t, mean=0.0, std=1.0, a=-2.0, b=2.0: nn.init.trunc_normal_(
            t, mean, std, a, b
        )
        sys.modules.setdefault("timm", types.ModuleType("timm"))
        sys.modules.setdefault("timm.models", types.ModuleType("timm.models"))
        sys.modules["timm.models.layers"] = tl
    if "fairscale.nn" not in sys.modules:
        fsn = types.ModuleType("fairscale.nn")
        fsn.checkpoint_wrapper = lambda m, offload_to_cpu=False: m
        sys.modules.setdefault("fairscale", types.ModuleType("fairscale"))
        sys.modules["fairscale.nn"] = fsn
    if "omegaconf" not in sys.modules:
        oc = types.ModuleType("omegaconf")
        oc.OmegaConf = type(
            "OmegaConf", (), {"create": staticmethod(lambda d: types.SimpleNamespace(**d))}
        )
        sys.modules["omegaconf"] = oc


def import_reference_grl():
    """Returns the reference ``GRL`` class (unmodified source)."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    install_shims()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from models.networks.grl import GRL  # noqa: E402

    return GRL


def import_reference_ops():
    """Returns the reference ``models.common.ops`` module (index/mask/table generators)."""
    import_reference_grl()
    import models.common.ops as ops  # noqa: E402

    return ops
