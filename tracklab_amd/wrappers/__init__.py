from .track import HipBoTSORT, HipDeepOCSORT, HipBPBReIDStrongSORT, HipByteTrack, HipOCSORT, HipStrongSORT  # noqa: F401
from .detect import HipYOLOX  # noqa: F401
from .reid import HipPartReID  # noqa: F401
from .pose import HipRTMPose  # noqa: F401
from .eval import HipTrackEvalEvaluator  # noqa: F401
