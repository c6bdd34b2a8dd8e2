"""Hyper-parameter record of the AR/NAR codec-token transformer path.

Mirror of the reference constructor contract (`valle/config.py:7-99`): every field name and
default, the three validation errors (`:67-72`), the directory side effect (`:74-77`), the derived
`quantization_factor` / `bos_token` / `eos_token` (`:79-89`) and the `from_dict` / `from_json`
loaders (`:91-99`).  The field table below is data, not logic, so it is kept as one table and the
dataclass is generated from it.
"""
from __future__ import annotations

import json
from dataclasses import field, make_dataclass
from pathlib import Path

# (name, type, default, help) -- order matters: positional construction must match the reference.
_FIELDS = [
    # data
    ('dataset', str, 'keithito/lj_speech', 'Hugging Face dataset'),
    ('num_workers', int, 4, 'Number of workers'),
    # input features
    ('vocab_size', int, 256, 'Vocab size'),
    ('num_audio_tokens', int, 1024, 'Number of audio tokens'),
    ('num_quantizers', int, 8, 'Number of quantizers layers from the audio codec'),
    ('sampling_rate', int, 16000, 'Sampling rate'),
    ('polling_factor', int, 320, 'Polling factor'),
    # model
    ('d_model', int, 256, 'Model dimension'),
    ('n_heads', int, 4, 'Number of heads'),
    ('dim_feedforward', int, 1024, 'Feedforward dimension'),
    ('dropout', float, 0.1, 'Dropout rate'),
    ('activation', str, 'relu', 'Activation function'),
    ('num_layers', int, 8, 'Number of layers'),
    ('norm', str, 'AdaptiveLayerNorm', 'Normalization layer'),
    # optimizer
    ('lr', float, 1e-4, 'Learning rate'),
    ('lr_warmup', int, 1000, 'Learning rate warmup steps'),
    ('betas', tuple, (0.9, 0.98), 'Betas for Adam optimizer'),
    ('weight_decay', float, 0.1, 'Weight decay'),
    ('use_fused_adam', bool, True, 'Use fused Adam optimizer'),
    ('gradient_clip_val', float, 1.0, 'Gradient clipping value'),
    ('grad_accum', int, 1, 'Gradient accumulation steps'),
    # generation
    ('max_audio_len', int, 1024, 'Max length for generation'),
    ('num_beams', int, 4, 'Number of beams for generation'),
    ('use_kv_cache', bool, True, 'Use key-value cache for generation'),
    ('top_k', int, 50, 'Top-k for sampling'),
    ('tok_p', float, 1.0, 'Token probability'),
    ('temperature', float, 1.0, 'Temperature'),
    ('length_penalty', float, 1.0, 'Length penalty'),
    # training
    ('seed', int, 42, 'Seed for reproducibility'),
    ('batch_size', int, 4, 'Batch size'),
    ('valid_batch_size', int, 1, 'Validation batch size'),
    ('max_steps', int, 1000, 'Max steps'),
    ('log_every_n_steps', int, 100, 'Log every n steps'),
    ('ckpt_path', Path, Path('models/checkpoints'), 'Checkpoint path'),
    ('log_path', Path, Path('models/logs'), 'Log path'),
]

_NORMS = ('AdaptiveLayerNorm', 'LayerNorm')
_ACTIVATIONS = ('relu', 'gelu')


def _post_init(self):
    if self.dataset is None:
        raise ValueError('Dataset must be provided')
    if self.norm not in _NORMS:
        raise ValueError('Normalization layer must be AdaptiveLayerNorm or LayerNorm')
    if self.activation not in _ACTIVATIONS:
        raise ValueError('Activation function must be relu or gelu')
    # The reference creates both directories as a side effect of construction; callers
    # (train(), TensorBoard logger) rely on them existing.
    for name in ('ckpt_path', 'log_path'):
        p = Path(getattr(self, name))
        p.mkdir(parents=True, exist_ok=True)
        setattr(self, name, p)


def _from_dict(cls, hparams_dict):
    return cls(**hparams_dict)


def _from_json(cls, json_file):
    with open(json_file, encoding='utf-8') as fh:
        return cls.from_dict(json.load(fh))


ConfigValle = make_dataclass(
    'ConfigValle',
    [(n, t, field(default=d, metadata={'help': h})) for n, t, d, h in _FIELDS],
    namespace={
        '__post_init__': _post_init,
        '__module__': __name__,
        'quantization_factor': property(lambda self: self.sampling_rate // self.polling_factor),
        'bos_token': property(lambda self: self.num_audio_tokens + 1),
        'eos_token': property(lambda self: self.num_audio_tokens),
        'from_dict': classmethod(_from_dict),
        'from_json': classmethod(_from_json),
    },
)
ConfigValle.__doc__ = 'Hyper-parameters of ValleAR / ValleNAR (reference: valle/config.py:7-99).'
