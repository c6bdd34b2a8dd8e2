"""Generate the golden vectors under tests/golden/ by running the REAL reference on CPU.

Runs only where /root/reference exists (the build container); a no-op anywhere else.  The
reference cannot travel, so what is committed is data only: seeded inputs are regenerated from
`valle2_amd.synth` on both sides and this script stores the reference's *outputs* as .npz.

How the reference is imported: its packages need third-party modules that are not
in this image (`lightning`, `coloredlogs`, `torchaudio`, `encodec`) and one function that newer
`transformers` removed (`top_k_top_p_filtering`, pinned 4.38.2).  None of them carries arithmetic of
the path except the last, which is rebuilt from transformers' own still-shipped TopK/TopP warpers.
In-memory stand-ins are registered in `sys.modules` *before* the import; nothing is written to the
reference tree (PYTHONDONTWRITEBYTECODE) and the CWD is a scratch dir (ConfigValle mkdirs, D12).

Usage:  python tests/golden/gen_golden.py            (writes tests/golden/*.npz)
"""
from __future__ import annotations

import logging
import os
import sys
import tempfile
import types
from pathlib import Path

REF = Path('/root/reference')
HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent


def import_reference():
    """Import the reference `valle` package from /root/reference with absent third-party modules
    stubbed in memory.  Returns the dict of modules the generator uses."""
    sys.dont_write_bytecode = True
    import torch.nn as nn
    import transformers  # noqa: F401  (import the real package first)
    import transformers.generation.utils as tgu
    from transformers.generation.logits_process import TopKLogitsWarper, TopPLogitsWarper

    if not hasattr(tgu, 'top_k_top_p_filtering'):
        def top_k_top_p_filtering(logits, top_k=0, top_p=1.0, filter_value=-float('inf'),
                                  min_tokens_to_keep=1):
            # body of transformers==4.38.2 generation/utils.py top_k_top_p_filtering
            if top_k > 0:
                logits = TopKLogitsWarper(top_k=top_k, filter_value=filter_value,
                                          min_tokens_to_keep=min_tokens_to_keep)(None, logits)
            if 0 <= top_p <= 1.0:
                logits = TopPLogitsWarper(top_p=top_p, filter_value=filter_value,
                                          min_tokens_to_keep=min_tokens_to_keep)(None, logits)
            return logits
        tgu.top_k_top_p_filtering = top_k_top_p_filtering

    class _LightningModule(nn.Module):
        def log(self, *a, **k):
            return None

    lightning = types.ModuleType('lightning')
    lightning.LightningModule = _LightningModule
    coloredlogs = types.ModuleType('coloredlogs')
    coloredlogs.ColoredFormatter = logging.Formatter
    torchaudio = types.ModuleType('torchaudio')
    encodec = types.ModuleType('encodec')
    encodec.EncodecModel = object
    for name, mod in (('lightning', lightning), ('coloredlogs', coloredlogs),
                      ('torchaudio', torchaudio), ('encodec', encodec)):
        sys.modules.setdefault(name, mod)

    # the build's own `valle` alias package must not shadow the reference
    for k in [k for k in sys.modules if k == 'valle' or k.startswith('valle.')]:
        del sys.modules[k]
    sys.path.insert(0, str(REF))
    try:
        import valle.config as rconfig
        import valle.models.modules as rmodules
        import valle.models.utils as rutils
        import valle.models.valle_ar as rar
        import valle.models.valle_nar as rnar
        import valle.collate  # noqa: F401  (kept in sys.modules for the collate case)
    finally:
        sys.path.remove(str(REF))
    assert str(REF) in rmodules.__file__, rmodules.__file__
    return dict(config=rconfig, modules=rmodules, utils=rutils, ar=rar, nar=rnar)


def main():
    if not REF.exists():
        print('no /root/reference here: nothing to do')
        return 0
    sys.path.insert(0, str(REPO))
    os.chdir(tempfile.mkdtemp(prefix='golden_cwd_'))
    import numpy as np
    import torch

    from tests.golden import cases  # shared case definitions (inputs + configs)

    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    ref = import_reference()
    out = {}
    only = set(sys.argv[1:])
    for name, fn in cases.REFERENCE_RUNNERS.items():
        if only and name not in only:
            continue
        res = fn(ref)
        out[name] = res
        path = HERE / f'{name}.npz'
        np.savez_compressed(path, **{k: (v.numpy() if hasattr(v, 'numpy') else np.asarray(v))
                                     for k, v in res.items()})
        print(f'{name}: wrote {path.name} ({path.stat().st_size} B) keys={sorted(res)}')
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
