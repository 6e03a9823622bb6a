# AI-TOD-v2 Point-Teacher, point noise m = 60 % (mirror of the reference's aitodv2_point_teacher_60%.py)
import os as _os
import sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import common as _common  # noqa: E402
_sys.path.pop(0)

_base_ = ['../_base_/datasets/aitodv2_detection_point.py', '../_base_/schedules/schedule_1x.py',
          '../_base_/default_runtime.py']
globals().update(_common.make(60))
