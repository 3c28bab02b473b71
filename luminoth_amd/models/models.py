"""Model registry — the drop-in boundary of the hot path.

Same contract as luminoth/models/models.py:6-17: `get_model(type)` lower-cases
the name, returns the model class, and raises ValueError for unknown types.
"""


def _registry():
    from luminoth_amd.models.fasterrcnn.fasterrcnn import FasterRCNN
    from luminoth_amd.models.ssd.ssd import SSD
    return {'fasterrcnn': FasterRCNN, 'ssd': SSD}


class _Lazy(dict):
    def __missing__(self, key):
        self.update(_registry())
        if key not in self:
            raise KeyError(key)
        return dict.__getitem__(self, key)

    def keys(self):
        self.update(_registry())
        return dict.keys(self)


MODELS = _Lazy()


def get_model(model_type):
    model_type = model_type.lower()
    try:
        return MODELS[model_type]
    except KeyError:
        raise ValueError('"{}" is not a valid model_type'.format(model_type))


def get_model_defaults(model_type):
    model_type = model_type.lower()
    if model_type == 'fasterrcnn':
        from luminoth_amd.models.fasterrcnn.defaults import DEFAULTS
    elif model_type == 'ssd':
        from luminoth_amd.models.ssd.defaults import DEFAULTS
    else:
        raise ValueError('"{}" is not a valid model_type'.format(model_type))
    return DEFAULTS
