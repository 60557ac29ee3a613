"""The backbones the adapters call: network definitions as torch modules (device memory, streams, hipGraph capture); at
fp32 -- the reference's precision -- every convolution, depthwise convolution and SPP block is a libtlk kernel (csrc/tlk_conv.hip,
tlk_dwconv.hip, tlk_spp.hip), at f16 the wide layers stay on the tuned library route (MIOpen / CK / hipBLASLt + libtlk's fused epilogue).
Random-init weights of the named architectures (no network access for checkpoints); BatchNorm is folded into the convolutions, as any
inference deployment does."""
import os as _os

# MIOpen's Find step times every applicable solver once per new convolution shape; its reference "naive" solver takes
# ~140 ms per shape at ReID batch sizes (60+ s of start-up for ResNet-50). It is never the winner: leave it out.
_os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "0")

from .yolox import YOLOX, yolox  # noqa: F401,E402
from .reid import PartBasedReID  # noqa: F401,E402

if _os.environ.get("TLK_MIOPEN_BENCHMARK", "1") == "1":
    # Let MIOpen time its solvers per convolution shape on first use instead of trusting the immediate-mode heuristic:
    # +12 % end-to-end on config3 (206 -> 231 frames/s) for ~10 s of extra start-up; TLK_MIOPEN_BENCHMARK=0 opts out.
    import torch as _torch
    _torch.backends.cudnn.benchmark = True
