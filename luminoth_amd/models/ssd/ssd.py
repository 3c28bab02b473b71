"""SSD model module (reference: luminoth/models/ssd/ssd.py:17-334).

Round 1 status: the HIP kernels SSD needs (3x3/dilated/VALID convs with bias,
2x2 and 3x3 max-pools, IoU targets, per-class NMS `lmh_rcnn_proposal` with
class_agnostic_boxes=1, smooth-L1 / CE) exist in libluminoth_hip.so; the module
wiring (multibox heads, hard-negative mining target, SSD loss) is the next §8
row (S1-S6) and is not built yet.  Constructing it raises instead of silently
falling back to anything else."""


class SSD(object):
    def __init__(self, config, name='ssd', device=None):
        raise NotImplementedError(
            'luminoth_amd: SSD (SURVEY.md §8 rows S1-S6) is not wired yet; Faster R-CNN is the round-1 path')
