"""Backbone names (reference code/yolo3/enums.py:25-28)."""
from enum import Enum, unique


@unique
class BACKBONE(Enum):
    MOBILENETV2x75 = 0
    MOBILENETV2x14 = 1
    EFFICIENTNETB3 = 2
