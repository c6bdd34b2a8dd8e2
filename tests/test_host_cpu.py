"""CPU-only checks: the C-ABI library loads and exports every symbol the header declares, the
host-side mirror keeps the reference's contracts (config, registry, mask builders, state_dict
keys), and the product refuses to compute without a HIP device (no silent CPU fallback)."""
import json
import re
from pathlib import Path

import pytest
import torch

REPO = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    from valle2_amd import _lib
    lib = _lib.load_library()
    header = (REPO / 'include' / 'valle_hip.h').read_text()
    declared = set(re.findall(r'\b(vh_[a-z0-9_]+)\s*\(', header)) - {'vh_ar_decoder'}
    assert declared, 'no declarations parsed'
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in include/valle_hip.h but not exported'
    assert int(re.search(r'#define VH_VERSION (\d+)', header).group(1)) == lib.vh_version()


def test_graft_entry_build_runs_and_checks_the_version_against_the_header(capsys):
    """The driver's "does it build" check: make for gfx950 (a no-op when the tree is built), dlopen, every symbol, and the
    library's version equal to the header's — not to a literal that a version bump leaves behind."""
    import __graft_entry__
    __graft_entry__.build()
    assert 'vh_version=' in capsys.readouterr().out
    assert 'vh_version() ==' not in (REPO / '__graft_entry__.py').read_text().replace('vh_version() == declared', '')


def test_no_cpu_fallback_anywhere():
    from valle2_amd import _lib
    from valle2_amd.modules import FeedForward, MultiHeadAttention, TokenEmbedding, Transformer
    from valle2_amd.config import ConfigValle
    if torch.cuda.is_available():
        pytest.skip('box has a GPU')
    with pytest.raises(_lib.VhError):
        _lib.lib()
    x = torch.randn(2, 5, 128)
    for mod, arg in ((MultiHeadAttention(128, 2), x), (FeedForward(128, 256), x),
                     (TokenEmbedding(10, 128), torch.zeros(2, 3, dtype=torch.int64)),
                     (Transformer(ConfigValle(d_model=128, n_heads=2, num_layers=1)), x)):
        with pytest.raises(_lib.VhError):
            mod(arg)


def test_product_never_imports_the_oracle():
    for path in list((REPO / 'valle2_amd').rglob('*.py')) + list((REPO / 'valle').rglob('*.py')):
        text = path.read_text()
        assert not re.search(r'^\s*(from|import)\s+oracle\b', text, re.M), f'{path} imports oracle/'
        assert 'valle_oracle' not in text, f'{path} mentions the oracle module'


def test_config_contract():
    from valle2_amd.config import ConfigValle
    c = ConfigValle()
    assert (c.vocab_size, c.num_audio_tokens, c.num_quantizers, c.d_model, c.n_heads) == (256, 1024, 8, 256, 4)
    assert (c.dim_feedforward, c.num_layers, c.norm, c.activation) == (1024, 8, 'AdaptiveLayerNorm', 'relu')
    assert (c.max_audio_len, c.num_beams, c.use_kv_cache, c.top_k, c.tok_p) == (1024, 4, True, 50, 1.0)
    assert c.betas == (0.9, 0.98) and c.lr == 1e-4 and c.weight_decay == 0.1 and c.lr_warmup == 1000
    assert c.quantization_factor == 50 and c.eos_token == 1024 and c.bos_token == 1025
    assert Path('models/checkpoints').is_dir() and Path('models/logs').is_dir()   # reference side effect
    for bad in (dict(norm='RMSNorm'), dict(activation='swish'), dict(dataset=None)):
        with pytest.raises(ValueError):
            ConfigValle(**bad)
    Path('hp.json').write_text(json.dumps({'d_model': 128, 'n_heads': 2, 'norm': 'LayerNorm'}))
    assert ConfigValle.from_json('hp.json').d_model == 128
    assert ConfigValle.from_dict({'num_layers': 3}).num_layers == 3
    with pytest.raises(TypeError):
        ConfigValle(no_such_field=1)


def test_reference_import_paths_and_registry():
    import valle.config
    import valle.models
    import valle.models.modules as mods
    import valle.models.utils as utils
    from valle.models import MODEL_DICT, get_model_class
    assert get_model_class('ValleAR').__name__ == 'ValleAR'
    assert get_model_class('ValleNAR').__name__ == 'ValleNAR'
    assert set(MODEL_DICT.keys()) == {'EncodecPip', 'ValleAR', 'ValleNAR'}
    with pytest.raises(ImportError):
        get_model_class('EncodecPip')
    for name in ('TokenEmbedding', 'PositionalEncoding', 'AdaptiveLayerNorm', 'MultiHeadAttention',
                 'FeedForward', 'EncoderLayer', 'Transformer'):
        assert hasattr(mods, name)
    for name in ('build_pad_mask', 'build_attn_mask', 'topk_sampling', 'get_best_beam'):
        assert hasattr(utils, name)


# ---- the reference's own CPU tests on mask plumbing (tests/test_models_utils.py:7-59,
# ---- tests/test_modules.py:33-79), same literals
def test_build_attn_mask_literal():
    from valle.models.utils import build_attn_mask
    expected = torch.tensor(
        [[0, 0, 0, 0, 0, 1, 1, 1, 1, 1]] * 5 +
        [[0, 0, 0, 0, 0, 0, 1, 1, 1, 1], [0, 0, 0, 0, 0, 0, 0, 1, 1, 1], [0, 0, 0, 0, 0, 0, 0, 0, 1, 1],
         [0, 0, 0, 0, 0, 0, 0, 0, 0, 1], [0, 0, 0, 0, 0, 0, 0, 0, 0, 0]], dtype=torch.bool)
    mask = build_attn_mask(5, 5, device='cpu')
    assert mask.shape == expected.shape and torch.equal(mask, expected)


@pytest.mark.parametrize('lens,expected', [
    (torch.tensor([5, 5, 5, 5]), torch.zeros(4, 5, dtype=torch.bool)),
    (torch.tensor([5, 4, 3, 2]), torch.tensor([[0, 0, 0, 0, 0], [0, 0, 0, 0, 1], [0, 0, 0, 1, 1],
                                               [0, 0, 1, 1, 1]], dtype=torch.bool))])
def test_build_pad_mask_literal(lens, expected):
    from valle.models.utils import build_pad_mask
    mask = build_pad_mask(lens, device='cpu')
    assert mask.shape == expected.shape and torch.equal(mask, expected)


@pytest.mark.parametrize('d_model,n_heads,batch_size,seq_len,expected', [
    (512, 8, 4, 5, [120, 112, 96, 72]),
    (256, 4, 8, 10, [220, 216, 208, 196, 180, 160, 136, 108])])
def test_merge_masks_zero_counts(d_model, n_heads, batch_size, seq_len, expected):
    from valle.models.modules import MultiHeadAttention
    attention = MultiHeadAttention(d_model=d_model, n_heads=n_heads)
    pad = (torch.arange(seq_len)[None, :] >= (seq_len - torch.arange(batch_size))[:, None]).long()
    attn_mask = torch.triu(torch.ones(seq_len, seq_len), diagonal=1)
    mask = attention.merge_masks(batch_size, attn_mask, pad)
    assert isinstance(mask, torch.Tensor) and mask.shape == (batch_size, n_heads, seq_len, seq_len)
    assert [(mask[i] == 0.0).sum().item() for i in range(batch_size)] == expected
    assert attention.merge_masks(batch_size, None, pad) is None


def test_masks_match_golden_and_oracle():
    from tests.oracle_runners import load_golden
    from valle2_amd.utils import build_attn_mask, build_pad_mask
    gold = load_golden('masks')
    assert torch.equal(build_attn_mask(5, 5, 'cpu'), gold['attn_5_5'])
    assert torch.equal(build_attn_mask(3, 7, 'cpu'), gold['attn_3_7'])
    assert torch.equal(build_pad_mask(torch.tensor([5, 4, 3, 2]), 'cpu'), gold['pad_b'])
    assert build_attn_mask(3, 7, 'cpu')._vh_prefix == (3, 7)


def test_state_dict_keys_match_reference_layout():
    from valle2_amd import get_model_class, synth
    from valle2_amd.config import ConfigValle
    for name, norm in (('ValleAR', 'LayerNorm'), ('ValleNAR', 'AdaptiveLayerNorm'), ('ValleNAR', 'LayerNorm')):
        cfg = ConfigValle(d_model=128, n_heads=2, dim_feedforward=256, num_layers=2, norm=norm)
        model = get_model_class(name)(cfg)
        sd = model.state_dict()
        shapes = synth.state_dict_shapes(cfg, name)
        assert set(sd) == set(shapes), set(sd) ^ set(shapes)
        for k, shp in shapes.items():
            assert tuple(sd[k].shape) == tuple(shp), k
        model.load_state_dict(synth.make_state_dict(cfg, name, seed=0))    # strict load
    n_ar = sum(p.numel() for p in get_model_class('ValleAR')(
        ConfigValle(d_model=128, n_heads=2, dim_feedforward=512, num_layers=2, norm='LayerNorm')).parameters())
    assert n_ar == 691072          # verified against the reference


def test_get_best_beam_and_sampling_contract():
    from tests.oracle_runners import load_golden
    from valle2_amd.utils import get_best_beam, topk_sampling
    from tests.golden.cases import sampling_inputs
    gold = load_golden('sampling')
    _, x, lp = sampling_inputs()
    assert torch.equal(get_best_beam(x, lp, 1024, 1.0), gold['best_beam_1'])
    assert torch.equal(get_best_beam(x, lp, 1024, 0.0), gold['best_beam_2'])
    from valle2_amd import _lib
    if not torch.cuda.is_available():                       # (host logits hop to the device when there is one: test_models_gpu)
        with pytest.raises(_lib.VhError):
            topk_sampling(torch.randn(2, 10), top_k=1)      # no device: no CPU sampler to fall back to


def test_bench_byte_accounting():
    import bench
    b, mean_s = bench.attn_algorithmic_bytes(32, 512, 1024, 512)
    assert abs(mean_s - (1024 + 256)) < 1e-9
    assert b == 4.0 * (2 * 32 * 1280 * 512 + 2 * 32 * 512)


def test_collate_matches_reference_golden():
    """valle/collate.py:19-44 wire format (BOS-prepend / EOS-append / zero pad / lens), pinned by a
    fixture produced by the reference's own ValleARCollate; NAR layout fixed per defect D7."""
    from tests.golden import cases as C
    from tests.oracle_runners import load_golden
    from valle.collate import ValleARCollate, ValleNARCollate, collate_list, get_collate
    gold = load_golden('collate')
    cfg = C.cfg_of(C.AR_TINY)
    items = C.collate_inputs()
    out = ValleARCollate(cfg)(items)
    assert set(out) == set(gold)
    for k in gold:
        assert out[k].dtype == torch.int64 and torch.equal(out[k], gold[k]), k
    nar = ValleNARCollate(cfg)(items)
    assert nar['codes'].shape == (3, 20, 8) and nar['codes_lens'].tolist() == [12, 9, 20]
    assert torch.equal(nar['codes'][1, :9], items[1]['codes'].T) and nar['codes'][1, 9:].sum() == 0
    assert get_collate('ValleAR') is ValleARCollate
    x, lens = collate_list([torch.ones(2), torch.ones(5)])
    assert x.shape == (2, 5) and lens.tolist() == [2, 5]
    with pytest.raises(AssertionError):       # codes must be longer than tokens (collate.py:37)
        ValleARCollate(cfg)([{'codes': torch.zeros(8, 3, dtype=torch.int64),
                              'tokens': torch.zeros(9, dtype=torch.int64)}])


def test_workspace_plans_of_the_gemm_entry_points():
    """Host-side plans behind the workspace queries (pure arithmetic, no GPU): the tile kernel's tail split
    (vh_linear_ex_ws_bytes: K slices of the tiles beyond the last multiple of 256 when those fill <= half of the CUs),
    what vh_linear_ws asks for, and the weight-gradient split over one workgroup per CU."""
    from valle2_amd import _lib
    L = _lib.load_library()                 # host-only entry points: no device needed
    slab = 128 * 128 * 4
    assert L.vh_linear_ex_ws_bytes(10240, 512, 512) == 64 * 4 * slab        # 320 tiles: 64 tail tiles x 4 K slices
    assert L.vh_linear_ex_ws_bytes(10240, 512, 2048) == 64 * 4 * slab
    assert L.vh_linear_ex_ws_bytes(8800, 512, 512) == 20 * 4 * slab          # 276 tiles: K / 8 would be < 128 -> 4 slices
    assert L.vh_linear_ex_ws_bytes(8800, 512, 2048) == 20 * 8 * slab
    assert L.vh_linear_ex_ws_bytes(16384, 512, 512) == 0                     # 512 tiles: no tail
    assert L.vh_linear_ex_ws_bytes(12800, 512, 512) == 0                     # 400 tiles: the tail fills more than half
    assert L.vh_linear_ex_ws_bytes(1000, 512, 2048) == 0                     # fewer than 256 tiles: split-K territory
    assert L.vh_linear_ex_ws_bytes(10240, 1025, 512) == 0                    # ragged N: never split
    assert L.vh_linear_ws_bytes(10240, 512, 512) == 64 * 4 * slab            # vh_linear_ws hands its workspace on
    assert L.vh_linear_ws_bytes(1000, 512, 2048) > 0 and L.vh_linear_ws_bytes(32, 512, 2048) > 0   # split-K plans
    assert L.vh_linear_ws_bytes(32, 512, 512) == 0
    # dW (512, 2048) over 16 320 tokens: 64 tiles -> 4 slices (one workgroup per CU), 4 slabs of the weight's size
    assert L.vh_gemm_tn_ws_bytes(16320, 512, 2048) == 4 * 512 * 2048 * 4


def test_attention_backward_chunk_plan():
    """The five-product attention backward cuts the keys of a (batch row, head) into chunks of 97..256 keys, one workgroup
    each (vh_attn_rows_bwd_chunks: host logic, a simulated launch — DESIGN.md 3.5); the workspace holds one dQ slab per
    chunk for either mask family."""
    from valle2_amd import _lib
    L = _lib.load_library()
    FULL, PREFIX = 0, 1
    for B, h, T in [(16, 8, 1020), (16, 8, 640), (2, 16, 2875), (8, 8, 1020), (32, 8, 640), (1, 2, 300), (2, 2, 70),
                    (1, 1, 256), (1, 1, 257), (3, 5, 1), (64, 8, 1024), (4, 16, 5000)]:
        ncs = []
        for mode in (FULL, PREFIX):
            nc = L.vh_attn_rows_bwd_chunks(B, h, T, mode)
            ncs.append(nc)
            keys = ((T + nc - 1) // nc + 31) // 32 * 32              # the launch's chunk size
            assert -(-T // 256) <= nc <= max(-(-T // 256), -(-T // 128)), (B, h, T, mode, nc)
            assert keys <= 256 and keys * (nc - 1) < T, 'every chunk fits a workgroup and owns at least one key'
            assert L.vh_attn_rows_bwd_chunks(B, h, T, mode) == nc    # cached, deterministic
        need = L.vh_attn_rows_bwd_ws_bytes(B, h, T)
        slabs = max(ncs) * B * h * T * 64 * 4 if max(ncs) > 1 else 0
        assert need >= slabs and need >= B * h * T * 4 and need % 16 == 0
    # the shapes the rule was calibrated on (profiles/r4_sweep_attn_bwd_chunks.log)
    assert L.vh_attn_rows_bwd_chunks(16, 8, 1020, PREFIX) == 4      # AR step of configs[3]
    assert L.vh_attn_rows_bwd_chunks(16, 8, 640, FULL) == 5          # NAR step: 5 x 128 keys beat 3 x 224 and 4 x 160
    assert L.vh_attn_rows_bwd_chunks(8, 8, 1020, PREFIX) == 8        # half the (b, head) pairs: twice the chunks
    assert L.vh_attn_rows_bwd_chunks(2, 2, 70, PREFIX) == 1          # one chunk: dq written directly, no slab, no reduce
    assert L.vh_attn_rows_bwd_ws_bytes(0, 8, 100) == 0


def test_flat_adamw_refuses_cpu_parameters():
    from valle2_amd._lib import VhError
    from valle2_amd.optim import FlatAdamW
    with pytest.raises(VhError, match='HIP device'):
        FlatAdamW([torch.nn.Parameter(torch.zeros(4, 4))])


def test_bench_gpus_flag_spawns_one_rank_per_gpu():
    """`python bench.py --gpus N` without a launcher must itself start N ranks (RANK/LOCAL_RANK/WORLD_SIZE/
    MASTER_* set, one process each) before touching the GPU; under a launcher (WORLD_SIZE set) it must not."""
    import json
    import os
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env['VALLE2_BENCH_SPAWN_PROBE'] = '1'
    for n in (3, 8):                                   # 8 = the whole node the driver runs the scaling bench on
        out = subprocess.run([sys.executable, str(REPO / 'bench.py'), '--gpus', str(n), '--steps', '2'], env=env,
                             capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        ranks = sorted((json.loads(line) for line in out.stdout.splitlines() if line.startswith('{')),
                       key=lambda r: r['rank'])
        assert [r['rank'] for r in ranks] == list(range(n)) and [r['local'] for r in ranks] == list(range(n))
        assert all(r['world'] == n and r['addr'] == '127.0.0.1' for r in ranks)
        assert len({r['port'] for r in ranks}) == 1
        assert all(r['argv'] == ['--gpus', str(n), '--steps', '2'] for r in ranks)
    # under a launcher: one process, the launcher's rank
    env.update(WORLD_SIZE='8', RANK='5', LOCAL_RANK='5')
    out = subprocess.run([sys.executable, str(REPO / 'bench.py'), '--gpus', '8'], env=env, capture_output=True,
                         text=True, timeout=120)
    lines = [json.loads(line) for line in out.stdout.splitlines() if line.startswith('{')]
    assert len(lines) == 1 and lines[0]['rank'] == 5 and lines[0]['world'] == 8
    # default: a single rank, no children
    env = {k: v for k, v in env.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    out = subprocess.run([sys.executable, str(REPO / 'bench.py')], env=env, capture_output=True, text=True, timeout=120)
    lines = [json.loads(line) for line in out.stdout.splitlines() if line.startswith('{')]
    assert len(lines) == 1 and lines[0]['world'] == 1 and lines[0]['rank'] == 0


def test_raise_device_errors_reads_every_flag(monkeypatch):
    """With device=None every device's flag is read and cleared, not only the first (round-3 advisor finding)."""
    import torch
    from valle2_amd import _lib
    flags = {0: torch.zeros(1, dtype=torch.int32), 1: torch.tensor([_lib.DEVERR_TARGET], dtype=torch.int32)}
    monkeypatch.setattr(_lib, '_err_flags', flags)
    with pytest.raises(IndexError, match='target'):
        _lib.raise_device_errors()
    assert int(flags[1]) == 0
    _lib.raise_device_errors()                       # nothing pending any more
