"""dbot_ros_amd -- MI355X (gfx950) implementation of the dbot_ros particle-filter observation
model (RbSensor likelihood evaluator) behind the reference's sensor-builder plugin surface.

The numeric path is librbsensor_mi355x.so (hand-written HIP, C-ABI in include/rbsensor_mi355x.h).
This package is the thin host-side mirror used by tests and bench.py; it has no CPU fallback.
"""
from .sensor import CameraData, ObjectModel, RbSensor, RbSensorBuilder, RbSensorError  # noqa: F401

__all__ = ["CameraData", "ObjectModel", "RbSensor", "RbSensorBuilder", "RbSensorError"]
