"""MI355X-native hot path of GRL image restoration (gfx950 HIP kernels behind the reference's
model boundary).  ``GRL`` takes the reference's constructor arguments and state_dict."""
from .model import GRL  # noqa: F401
from .optim import FusedAdamW  # noqa: F401
from .train_graph import GraphedTrainStep  # noqa: F401
from .presets import baseline_config, make_config  # noqa: F401

__all__ = ["GRL", "FusedAdamW", "GraphedTrainStep", "make_config", "baseline_config"]
