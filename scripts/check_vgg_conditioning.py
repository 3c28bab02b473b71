"""Is the Faster R-CNN VGG-16 first-step loss gap between the direct and the Winograd 3x3 kernels (round 1:
128.9 vs 129.0 with raw random-init weights) a kernel error or the conditioning of an un-normalised network?

Runs ONE forward+loss of config 1 (600x800, 20 classes) with (a) raw He-init weights on raw 0..255 pixels and
(b) the same weights with conv1_1 divided by the pixel std, through: HIP direct kernels, HIP Winograd kernels, the
CPU oracle in fp32 and the CPU oracle in fp64 (on the HIP model's own ROIs), and prints every loss and its relative
distance from the fp64 value.  python scripts/check_vgg_conditioning.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch  # noqa: E402

from e2e_util import make_config, synth  # noqa: E402
from luminoth_amd import kernels as K  # noqa: E402
from luminoth_amd.models import get_model  # noqa: E402
from oracle import rng as orng  # noqa: E402
from oracle.model import OracleFasterRCNN  # noqa: E402

KEYS = ('rpn_cls_loss', 'rpn_reg_loss', 'rcnn_cls_loss', 'rcnn_reg_loss')
cfg = make_config('vgg_16', 20, **{'model.base_network.fine_tune_from': 'conv3'})
images, gts = synth(1, 600, 800, 3, 20, 7)
for tag, scale in (('raw init', 1.0), ('conv1_1 / 73.6', 1.0 / 73.6)):
    model = get_model('fasterrcnn')(cfg)
    sd = model.state_dict()
    sd['truncated_base_network/vgg_16/conv1/conv1_1/weights'].mul_(scale)
    model.load_state_dict(sd)
    res = {}
    ov = None
    for name, wino in (('hip direct', False), ('hip winograd', True)):
        K.WINOGRAD = wino
        model._step = 0
        pred = model(images, gts, is_training=True)
        losses = model.loss(pred, return_all=True)
        torch.cuda.synchronize()
        res[name] = {k: float(losses[k]) for k in KEYS}
        if ov is None:
            cp = pred['classification_prediction']
            n = int(cp['num_proposals'][0])
            ov = dict(rois=cp['proposals'][0, :n].cpu().numpy(), roi_labels=cp['target']['cls'][0, :n].cpu().numpy(),
                      roi_targets=cp['target']['bbox_offsets'][0, :n].cpu().numpy())
    for name, dt in (('oracle fp32', torch.float32), ('oracle fp64', torch.float64)):
        o = OracleFasterRCNN(sd, arch='vgg_16', num_classes=20, seed=0, fine_tune_from='conv3', dtype=dt)
        with torch.no_grad():
            out = o.forward_image(images[0], gts[0], orng.image_seed(0, 0, 0), overrides=ov)
        res[name] = {k: float(out[k]) for k in KEYS}
    ref = res['oracle fp64']
    print('== %s (rois of the direct HIP run in every evaluation; the RCNN losses of the Winograd run use its own rois)' % tag)
    for name, r in res.items():
        tot, tot64 = sum(r.values()), sum(ref.values())
        print('  %-13s total %.6f (rel. to fp64 %+.2e)  ' % (name, tot, (tot - tot64) / abs(tot64)) +
              '  '.join('%s %.6f (%+.1e)' % (k[:-5], r[k], (r[k] - ref[k]) / max(abs(ref[k]), 1e-12)) for k in KEYS))
