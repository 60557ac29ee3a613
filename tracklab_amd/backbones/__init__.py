"""PyTorch-ROCm backbones the adapters call: detector and ReID networks stay
PyTorch (MIOpen / hipBLASLt); everything around them is libtlk. Random-init weights of the named
architectures (no network access for checkpoints); BatchNorm is folded into the convolutions, as any
inference deployment does."""
from .yolox import YOLOX, yolox  # noqa: F401
from .reid import PartBasedReID  # noqa: F401
