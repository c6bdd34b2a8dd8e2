from valle2_amd import MODEL_DICT, get_model_class  # noqa: F401
from valle2_amd.valle_ar import ValleAR  # noqa: F401
from valle2_amd.valle_nar import ValleNAR  # noqa: F401

__all__ = ['ValleAR', 'ValleNAR', 'MODEL_DICT', 'get_model_class']
