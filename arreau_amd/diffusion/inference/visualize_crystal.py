"""Only the enum of the reference's visualisation module is on the sampling path
(diffusion/inference/visualize_crystal.py:16-20); plotting itself is out of scope."""
from enum import Enum


class VisualizationSetting(Enum):
    NONE = 0
    LAST = 1
    ALL = 2
    ALL_DETAILED = 3
