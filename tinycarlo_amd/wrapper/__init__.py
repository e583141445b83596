from .reward import (CTELinearRewardWrapper, CTESparseRewardWrapper, LanelineLinearRewardWrapper,  # noqa: F401
                     LanelineSparseRewardWrapper)
from .termination import CrashTerminationWrapper, CTETerminationWrapper, LanelineCrossingTerminationWrapper  # noqa: F401
from .observation import NoiseObservationWrapper  # noqa: F401
