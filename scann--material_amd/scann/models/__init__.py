from .scann_model import SCANN, HipModel, create_model, create_model_pretrained, load_model, normalize_config

__all__ = ["SCANN", "HipModel", "create_model", "create_model_pretrained", "load_model", "normalize_config"]
