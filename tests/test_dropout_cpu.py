"""Host-side checks of the dropout fields (no GPU): the numpy Philox restatement against the published known-answer
vectors, the field definition's statistics, and the oracle's two dropout modes against each other (its torch-RNG mode is
pinned bit for bit to the real reference by tests/golden/ar_train_dropout.npz / transformer_dropout.npz)."""
import numpy as np
import torch

from oracle import philox as P
from oracle import valle_oracle as O
from tests.golden import cases as C


def test_philox4x32_known_answers():
    """Random123 kat_vectors, philox4x32 with 7 and 10 rounds: (counter, key) -> output."""
    pi_ctr, pi_key = (0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)
    ff = 0xffffffff
    kat = [
        (10, (0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        (10, (ff,) * 4, (ff, ff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        (10, pi_ctr, pi_key, (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
        (7, (0, 0, 0, 0), (0, 0), (0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48)),
        (7, (ff,) * 4, (ff, ff), (0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662)),
        (7, pi_ctr, pi_key, (0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a)),
    ]
    for rounds, ctr, key, want in kat:
        got = tuple(int(x) for x in P.philox4x32(ctr, key, rounds))
        assert got == want, (rounds, [hex(g) for g in got])


def test_field_statistics_and_independence():
    rows, cols, p = 512, 256, 0.1
    n = rows * cols
    a = P.keep_field(1234, 0x401, p, rows, cols).astype(np.float64)
    b = P.keep_field(1234, 0x501, p, rows, cols).astype(np.float64)     # another site
    c = P.keep_field(1235, 0x401, p, rows, cols).astype(np.float64)     # another step
    sigma = (p * (1 - p) / n) ** 0.5
    for f in (a, b, c):
        assert abs(f.mean() - (1 - p)) < 4 * sigma
        # no structure along rows or columns either
        assert np.abs(f.mean(axis=0) - (1 - p)).max() < 5 * (p * (1 - p) / rows) ** 0.5
        assert np.abs(f.mean(axis=1) - (1 - p)).max() < 5 * (p * (1 - p) / cols) ** 0.5
    both = (1 - p) ** 2
    s2 = (both * (1 - both) / n) ** 0.5
    for x, y in ((a, b), (a, c), (b, c), (a[1:], a[:-1]), (a[:, 1:], a[:, :-1])):
        assert abs((x * y).mean() - both) < 4 * s2
    assert P.threshold(0.1) == round(np.float64(np.float32(0.1)) * 2 ** 32)


def test_oracle_mask_mode_replays_its_torch_mode():
    """The oracle run with GIVEN keep fields equals the oracle run that drew them (whose arithmetic is the reference's):
    the mask mode is what the GPU parity tests feed the HIP path's exported fields into."""
    _, sd, batch = C.ar_train_inputs()
    cfg = C.cfg_of(C.AR_TINY_DROPOUT)
    torch.manual_seed(C.DROPOUT_SEED)
    d1 = O.Dropout(cfg.dropout)
    ref = O.ar_logits(sd, cfg, batch, d1)
    assert len(d1.used) == 2 + 3 * cfg.num_layers
    keep = {k: ((v.permute(1, 0, 2) if k.endswith('position_emb.dropout') else v) != 0) for k, v in d1.used.items()}
    again = O.ar_logits(sd, cfg, batch, O.Dropout(cfg.dropout, masks=keep))
    torch.testing.assert_close(again, ref, atol=1e-6, rtol=1e-6)    # (1/(1-p) as a float product vs torch's division)
    gold_free = O.ar_logits(sd, cfg, batch)
    assert float((gold_free - ref).abs().max()) > 1e-2              # and dropout really changed the forward


def test_no_cpu_randomness_is_consumed_when_every_dropout_is_off():
    """Round-4 advisor finding: a forward in eval mode (or with p = 0) drew a seed from torch's global CPU generator anyway,
    advancing the user's random stream (DataLoader shuffling ...) where the reference consumes none."""
    import torch
    from valle2_amd import dropout
    from valle2_amd.config import ConfigValle
    from valle2_amd.modules import Transformer
    tr = Transformer(ConfigValle(d_model=128, n_heads=2, dim_feedforward=256, num_layers=2, dropout=0.1))
    torch.manual_seed(123)
    before = torch.get_rng_state()
    sd = dropout.StackDropout(list(tr.eval().layers))
    assert not sd.any and sd.seed == 0 and dropout.seed_if(0.0, 0.0) == 0
    assert torch.equal(torch.get_rng_state(), before)
    sd = dropout.StackDropout(list(tr.train().layers))            # live dropouts: one draw
    assert sd.any and not torch.equal(torch.get_rng_state(), before)
    mid = torch.get_rng_state()
    assert dropout.seed_if(0.0, 0.1) != 0 and not torch.equal(torch.get_rng_state(), mid)
