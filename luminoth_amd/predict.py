"""`lumi predict` re-hosted on the HIP path (reference: luminoth/predict.py:19-291).

    python -m luminoth_amd.predict IMAGE_OR_DIR... -c config.yml [-o k=v] [-f out.json] [--min-prob P]
                                   [--max-detections N] [-k CLASS | -K CLASS]

One JSON line per image: {"file": path, "objects": [{"bbox": [x1,y1,x2,y2], "label": l, "prob": p}, ...]} —
the reference's output format.  Video input (skvideo), drawing (`--save-media-to`, luminoth/vis.py) and the remote
checkpoint registry (`--checkpoint`) are outside the hot-path scope and are rejected with a message.
"""
import argparse
import json
import logging
import os
import sys

import numpy as np

IMAGE_FORMATS = ['jpg', 'jpeg', 'png']
ARRAY_FORMATS = ['npy']           # extension: a raw (H,W,3) array, for boxes without image codecs
VIDEO_FORMATS = ['mov', 'mp4', 'avi']


def get_file_type(filename):
    extension = filename.split('.')[-1].lower()
    if extension in IMAGE_FORMATS or extension in ARRAY_FORMATS:
        return 'image'
    elif extension in VIDEO_FORMATS:
        return 'video'


def resolve_files(path_or_dir, echo=print):
    """predict.py:28-56."""
    if not isinstance(path_or_dir, (tuple, list)):
        path_or_dir = (path_or_dir,)
    paths = []
    for entry in path_or_dir:
        if os.path.isdir(entry):
            paths.extend([os.path.join(entry, f) for f in sorted(os.listdir(entry))
                          if get_file_type(f) in ('image', 'video')])
        elif get_file_type(entry) in ('image', 'video'):
            if not os.path.exists(entry):
                echo('Input {} not found, skipping.'.format(entry))
                continue
            paths.append(entry)
    return paths


def filter_classes(objects, only_classes=None, ignore_classes=None):
    """predict.py:59-66."""
    if ignore_classes:
        objects = [o for o in objects if o['label'] not in ignore_classes]
    if only_classes:
        objects = [o for o in objects if o['label'] in only_classes]
    return objects


def load_image(path):
    if path.lower().endswith('.npy'):
        return np.load(path)
    from PIL import Image
    with open(path, 'rb') as f:
        return np.array(Image.open(f).convert('RGB'))


def predict_image(network, path, only_classes=None, ignore_classes=None, echo=print):
    """predict.py:69-97."""
    try:
        image = load_image(path)
    except (OSError, ValueError) as e:
        echo('Error while processing {}: {}'.format(path, e))
        return None
    objects = network.predict_image(image)
    return filter_classes(objects, only_classes=only_classes, ignore_classes=ignore_classes)


def apply_detection_limits(config, min_prob, max_detections):
    """predict.py:243-256."""
    if config.model.type == 'fasterrcnn':
        if config.model.network.with_rcnn:
            config.model.rcnn.proposals.total_max_detections = max_detections
        else:
            config.model.rpn.proposals.post_nms_top_n = max_detections
        config.model.rcnn.proposals.min_prob_threshold = min_prob
    elif config.model.type == 'ssd':
        config.model.proposals.total_max_detections = max_detections
        config.model.proposals.min_prob_threshold = min_prob
    else:
        raise ValueError("Model type '{}' not supported".format(config.model.type))
    return config


def predict(path_or_dir, config_files, override_params=(), output_path='-', min_prob=0.5, max_detections=100,
            only_class=None, ignore_class=None, echo=print, network_fn=None):
    if only_class and ignore_class:
        echo('Only one of `only-class` or `ignore-class` may be specified.')
        return None
    files = resolve_files(path_or_dir, echo)
    if not files:
        echo('No files to predict found. Accepted formats are: {}.'.format(
            ', '.join(IMAGE_FORMATS + ARRAY_FORMATS + VIDEO_FORMATS)))
        return None
    echo('Found {} files to predict.'.format(len(files)))
    from luminoth_amd.utils.config import get_config
    config = get_config(list(config_files), override_params=list(override_params))
    apply_detection_limits(config, min_prob, max_detections)
    if network_fn is None:
        from luminoth_amd.utils.predicting import PredictorNetwork as network_fn
    network = network_fn(config)
    output = sys.stdout if output_path == '-' else open(output_path, 'w')
    results = []
    try:
        for file in files:
            if get_file_type(file) == 'video':
                echo('Skipping {}: video input is not hosted.'.format(file))
                continue
            echo('Predicting {}...'.format(file))
            objects = predict_image(network, file, only_classes=only_class, ignore_classes=ignore_class, echo=echo)
            if objects is not None:
                output.write(json.dumps({'file': file, 'objects': objects}) + '\n')
                results.append((file, objects))
    finally:
        if output is not sys.stdout:
            output.close()
    return results


def main(argv=None):
    ap = argparse.ArgumentParser(description="Obtain a model's predictions (luminoth/predict.py:173-291).")
    ap.add_argument('path_or_dir', nargs='*')
    ap.add_argument('--config', '-c', dest='config_files', action='append', default=[], help='Config to use.')
    ap.add_argument('--checkpoint', help='(not hosted) remote checkpoint id.')
    ap.add_argument('--override', '-o', dest='override_params', action='append', default=[])
    ap.add_argument('--output', '-f', dest='output_path', default='-')
    ap.add_argument('--save-media-to', '-d', default=None)
    ap.add_argument('--min-prob', type=float, default=0.5)
    ap.add_argument('--max-detections', type=int, default=100)
    ap.add_argument('--only-class', '-k', action='append', default=None)
    ap.add_argument('--ignore-class', '-K', action='append', default=None)
    ap.add_argument('--debug', action='store_true')
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.DEBUG if args.debug else logging.ERROR)
    echo = (lambda m: print(m, file=sys.stderr)) if args.output_path == '-' else print
    if args.checkpoint or not args.config_files:
        echo('The checkpoint registry is not hosted: pass a config with -c (train.job_dir selects the checkpoint).')
        return 2
    if args.save_media_to:
        echo('--save-media-to (drawing) is not hosted; writing JSON only.')
    res = predict(tuple(args.path_or_dir), args.config_files, args.override_params, args.output_path,
                  args.min_prob, args.max_detections, args.only_class, args.ignore_class, echo=echo)
    return 0 if res is not None else 1


if __name__ == '__main__':
    sys.exit(main())
