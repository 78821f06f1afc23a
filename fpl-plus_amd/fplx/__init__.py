"""fplx - MI355X-native hot path of FPL+ (3D U-Net with domain-specific BatchNorm, segmentation
losses, pseudo-label uncertainty filter) behind PyMIC's registry / agent surface.

Everything numeric runs in hand-written HIP kernels (libfplx.so, C ABI in include/fplx.h);
importing the package without the built library raises - there is no CPU fallback.
"""
from . import _lib

_lib.lib()          # fail loudly, now, if the HIP extension is missing

from .net import UNet2D5_dsbn                                      # noqa: E402
from .dsbn import DomainSpecificBatchNorm3d                        # noqa: E402
from .loss import (SegLossDict, DiceLoss, CrossEntropyLoss, DiceLoss_weight, CombinedLoss,  # noqa: E402
                   EntropyTerm, make_loss)
from .infer import Inferer                                         # noqa: E402
from .agent import SegmentationAgent, SegNetDict                   # noqa: E402
from .optim import FusedAdam, get_optimizer, get_lr_scheduler      # noqa: E402
from .train import TrainStep                                       # noqa: E402
from .config import parse_config, synchronize_config               # noqa: E402
from .dataset import NiftyDataset                                  # noqa: E402
from . import filter, ops, ddp, transform, nifti, evaluation                                     # noqa: E402

__all__ = ["UNet2D5_dsbn", "DomainSpecificBatchNorm3d", "SegLossDict", "SegNetDict", "DiceLoss",
           "CrossEntropyLoss", "DiceLoss_weight", "CombinedLoss", "EntropyTerm", "make_loss", "Inferer",
           "SegmentationAgent", "FusedAdam", "get_optimizer", "get_lr_scheduler", "TrainStep",
           "parse_config", "synchronize_config", "filter", "ops", "ddp", "transform", "nifti", "evaluation", "NiftyDataset"]
