from .YOLOPoint import *   # noqa: F401,F403  (reference: src/models/__init__.py)
from .YOLOPoint import Model, YOLOPoint, YOLOPointv52
