from .models import get_model, get_model_defaults, MODELS  # noqa: F401
