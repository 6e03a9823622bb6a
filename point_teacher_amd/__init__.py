"""point_teacher_amd - MI355X-native implementation of the Point-Teacher teacher->student
training hot path behind the reference's registry surface.

Importing the package loads libpt_hip.so (through .hip) and registers every class under the
name the reference's configs use.  There is no CPU compute path."""
import os as _os
import sys as _sys

# `python -m point_teacher_amd.build`: the package is imported before its build module runs, and the library may not exist yet (a
# clean checkout) or be stale (a header that declares an entry point the old library lacks) - only then the imports below are
# skipped (the build module needs none of them).  Any other import without a matching library still fails loudly in .hip.
_BUILDING = 'point_teacher_amd.build' in getattr(_sys, 'orig_argv', ())
if _BUILDING:
    __all__ = []
else:
    from . import hip                  # noqa: F401  (raises loudly when the HIP library is missing)
    from . import functional               # noqa: F401
    from .registry import (BACKBONES, BBOX_ASSIGNERS, BBOX_CODERS, DETECTORS, HEADS, LOSSES, MATCH_COST, NECKS,  # noqa: F401
                           ROI_EXTRACTORS, Config, build_assigner, build_detector, build_from_cfg, build_loss)
    from . import core, losses, nn_modules, head, detectors, obb, obb_head, obb_detectors, datasets, fcos_baseline, retina_baseline, retina_student, faster_rcnn, yolof_baseline   # noqa: F401,E402  (populate the registries)
    from . import ops                      # noqa: F401,E402  (the mmcv.ops signatures of the path)
    from .runtime import Trainer           # noqa: F401,E402

__version__ = '0.1.0'
