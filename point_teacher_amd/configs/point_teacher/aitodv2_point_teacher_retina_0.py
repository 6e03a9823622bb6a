# Point-Teacher (0 % point noise) with a RetinaNet-style STUDENT / teacher head (row N4: another student architecture under the
# same teacher).  Everything but `bbox_head` equals aitodv2_point_teacher_0.py.
import os as _os
import sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import common as _common  # noqa: E402
_sys.path.pop(0)

_base_ = ['../_base_/datasets/aitodv2_detection_point.py', '../_base_/schedules/schedule_1x.py',
          '../_base_/default_runtime.py']


def _make():
    cfg = _common.make(0)
    head = cfg['model']['_model_']['bbox_head']          # the same dictionary object as cfg['detector']['bbox_head']
    head['type'] = 'TS_P2BRetinaHead'
    head['anchor_generator'] = dict(type='AnchorGenerator', octave_base_scale=2, scales_per_octave=1, ratios=[0.5, 1.0, 2.0],
                                    strides=[8])
    for k in ('norm_on_bbox', 'centerness_on_reg', 'center_sampling'):
        head.pop(k, None)
    return cfg


globals().update(_make())
del _make
