"""Pin the CPU oracle against golden vectors produced by the REAL reference
(tests/golden/gen_golden.py, run in the build container where /root/reference is mounted).

Integer/bool/index results must match exactly everywhere.  Floating-point results are bit-exact
on the generating box; on another host the CPU BLAS/oneDNN kernels may reassociate sums, so the
stated tolerance is atol 2e-5 + rtol 2e-5 (fp32, values O(1))."""
import pytest
import torch

from tests.oracle_runners import ORACLE_RUNNERS, PREFIX_KEYS, load_golden

ATOL = RTOL = 2e-5


@pytest.mark.parametrize('name', sorted(ORACLE_RUNNERS))
def test_oracle_matches_reference_golden(name):
    got = ORACLE_RUNNERS[name]()
    gold = load_golden(name)
    if name in PREFIX_KEYS:          # full-size case: the oracle re-runs only the first steps
        assert set(got) == set(PREFIX_KEYS[name])
        gold = {k: gold[k][: got[k].shape[0]] for k in got}
    assert set(got) == set(gold)
    for key, ref in gold.items():
        val = got[key]
        assert tuple(val.shape) == tuple(ref.shape), key
        if ref.dtype.is_floating_point:
            torch.testing.assert_close(val, ref, atol=ATOL, rtol=RTOL, msg=f'{name}:{key}')
        else:
            assert torch.equal(val.to(ref.dtype), ref), f'{name}:{key}'


def test_greedy_tokens_are_exact_and_margins_recorded():
    # the golden carries the per-step top-1/top-2 margin so GPU tests can tell a real
    # divergence from a near-tie (SURVEY.md §7 "hard parts")
    for name in ('ar_generate_tiny', 'ar_generate_mid', 'ar_generate_full', 'ar_generate_big'):
        gold = load_golden(name)
        assert gold['tokens'].dtype == torch.int64
        assert int(gold['steps']) == gold['margin'].numel() == gold['tokens'].numel()
        assert float(gold['margin'].min()) > 0


def test_eos_case_stops_early():
    gold = load_golden('ar_generate_eos')
    assert int(gold['steps']) <= 6 < gold['free_tokens'].numel()
    assert torch.equal(gold['tokens'], gold['free_tokens'][: gold['tokens'].numel()])


def test_reference_mask_literals():
    # the only value-exact tests the reference itself holds (tests/test_models_utils.py:7-59,
    # tests/test_modules.py:33-79): the zero counts of the merged additive mask.
    gold = load_golden('masks')
    assert [int((gold['merge_512'][i] == 0).sum()) for i in range(4)] == [120, 112, 96, 72]
    assert [int((gold['merge_256'][i] == 0).sum()) for i in range(8)] == \
        [220, 216, 208, 196, 180, 160, 136, 108]
    a = gold['attn_5_5']
    assert a.dtype == torch.bool and a[:5, 5:].all() and not a[:5, :5].any()
    assert torch.equal(a[5:, 5:], torch.triu(torch.ones(5, 5, dtype=torch.bool), diagonal=1))
