from valle2_amd import MODEL_DICT, get_model_class  # noqa: F401
from valle2_amd.valle_ar import ValleAR  # noqa: F401
from valle2_amd.valle_nar import ValleNAR  # noqa: F401

__all__ = ['EncodecPip', 'ValleAR', 'ValleNAR', 'MODEL_DICT', 'get_model_class']


def __getattr__(name):
    # `from valle.models import EncodecPip` (valle/models/__init__.py:1 of the reference) resolves lazily: the adapter over the
    # third-party codec when `encodec` is installed, ImportError otherwise — importing the package never needs it
    if name == 'EncodecPip':
        return get_model_class('EncodecPip')
    raise AttributeError(name)
