"""valle2_amd — MI355X-native AR + NAR codec-token transformer path (drop-in for valle.models).

Python here is host plumbing over libvalle_hip.so (hand-written gfx950 kernels, C ABI in
include/valle_hip.h).  There is no CPU compute path: ops raise `VhError` without a HIP device.
"""
from .config import ConfigValle  # noqa: F401

__all__ = ['ConfigValle', 'MODEL_DICT', 'get_model_class']


def _models():
    from .valle_ar import ValleAR
    from .valle_nar import ValleNAR
    return {'ValleAR': ValleAR, 'ValleNAR': ValleNAR}


class _LazyModelDict(dict):
    """MODEL_DICT of valle/models/__init__.py:5-9.  'EncodecPip' (third-party codec wrapper, out
    of scope) resolves lazily so importing the package never needs `encodec`."""

    def __missing__(self, key):
        if key in ('ValleAR', 'ValleNAR'):
            self.update(_models())
            return self[key]
        if key == 'EncodecPip':
            # valle/models/__init__.py:1,6 — resolved only when asked for: the adapter over the third-party codec when
            # `encodec` is installed (valle2_amd/codec_io.py: layouts only, the codec computes), ImportError otherwise
            try:
                import encodec  # noqa: F401
            except ImportError as e:
                raise ImportError('EncodecPip wraps the third-party `encodec` package, which is not installed here '
                                  '(pip install encodec==0.1.1); the token layouts it exchanges with the models are in '
                                  'valle2_amd.codec_io') from e
            from .codec_io import EncodecPip
            self[key] = EncodecPip
            return EncodecPip
        raise KeyError(key)

    def keys(self):
        return ['EncodecPip', 'ValleAR', 'ValleNAR']


MODEL_DICT = _LazyModelDict()


def get_model_class(model_name: str):
    """valle/models/__init__.py:12-13"""
    return MODEL_DICT[model_name]
