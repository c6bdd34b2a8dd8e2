"""Stochastic sampling on the device (`vh_sample_step`; valle/models/utils.py:46-68).

The reference draws with torch.multinomial on its own RNG stream, so sample-exact parity is
impossible by construction (SURVEY.md §8c "parity unpinned").  What IS checked, against the oracle's
restatement of the published top_k_top_p_filtering algorithm: the support (which tokens can ever be
drawn), the probabilities (empirical frequencies of 40 000 draws within 4 sigma), the returned
log-prob of every draw (atol 1e-5), tie handling, determinism under a seed, and the degenerate
top_k=1 case (arg-max, lowest index, log-prob exactly 0)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import valle_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda'
N = 40000


def expected_probs(logits_row, top_k, top_p, temperature):
    filt = O._top_k_top_p_filter((logits_row / temperature)[None], top_k=top_k, top_p=top_p)
    return F.softmax(filt, dim=-1)[0]


@pytest.mark.parametrize('top_k,top_p,temperature', [(5, 1.0, 1.0), (50, 1.0, 0.7), (0, 0.8, 1.0),
                                                     (10, 0.6, 1.3), (3, 0.999, 0.5)])
def test_distribution_support_and_logprob(top_k, top_p, temperature):
    from valle2_amd.utils import topk_sampling
    g = torch.Generator().manual_seed(7)
    row = 2.5 * torch.randn(1025, generator=g)
    probs = expected_probs(row, top_k, top_p, temperature)
    tok, lp = topk_sampling(row.to(DEV)[None].expand(N, -1).contiguous(), top_k=top_k, tok_p=top_p,
                            temperature=temperature, seed=1234)
    tok, lp = tok[:, 0].cpu(), lp.cpu()
    assert tok.dtype == torch.int64 and tok.shape == (N,)
    assert bool((probs[tok] > 0).all()), 'a token outside the filtered support was drawn'
    freq = torch.bincount(tok, minlength=1025).float() / N
    sigma = torch.sqrt(probs * (1 - probs) / N)
    assert bool(((freq - probs).abs() <= 4 * sigma + 1e-4).all()), (freq - probs).abs().max()
    torch.testing.assert_close(lp, torch.log(probs[tok]), atol=1e-5, rtol=1e-5)


def test_top_p_walkthrough_and_ties():
    from valle2_amd.utils import topk_sampling
    # probabilities .5 .3 .1 .06 .04 ; top_p = .8 removes ascending cumulative <= .2 → {.04,.06,.1}
    p = torch.tensor([0.3, 0.04, 0.5, 0.1, 0.06])
    tok, _ = topk_sampling(torch.log(p).to(DEV)[None].expand(5000, -1).contiguous(), top_k=0, tok_p=0.8,
                           seed=5)
    assert set(tok[:, 0].cpu().tolist()) == {0, 2}
    # top_k = 2 with a three-way tie at the second place keeps all tied scores
    logits = torch.tensor([1.0, 3.0, 1.0, -2.0, 1.0])
    tok, _ = topk_sampling(logits.to(DEV)[None].expand(5000, -1).contiguous(), top_k=2, seed=6)
    assert set(tok[:, 0].cpu().tolist()) == {0, 1, 2, 4}


def test_seed_determinism_and_greedy_case():
    from valle2_amd.utils import topk_sampling
    logits = torch.randn(64, 1025, generator=torch.Generator().manual_seed(3)).to(DEV)
    a, la = topk_sampling(logits, top_k=50, seed=11)
    b, lb = topk_sampling(logits, top_k=50, seed=11)
    c, _ = topk_sampling(logits, top_k=50, seed=12)
    assert torch.equal(a, b) and torch.equal(la, lb) and not torch.equal(a, c)
    torch.manual_seed(99)
    d1, _ = topk_sampling(logits, top_k=50)
    torch.manual_seed(99)
    d2, _ = topk_sampling(logits, top_k=50)
    assert torch.equal(d1, d2)                                   # torch.manual_seed governs the draw
    logits[1, 7] = logits[1, 900] = 50.0
    gt, gl = topk_sampling(logits, top_k=1)
    exp = torch.argmax(logits.cpu(), dim=-1)
    exp[1] = 7
    assert gt[:, 0].cpu().tolist() == exp.tolist() and float(gl.abs().max()) == 0.0
    # rows without an exact tie agree with the oracle's multinomial-over-a-one-hot (the reference
    # keeps exact ties of the top score and draws among them at random; the device picks the lowest
    # index, documented in include/valle_hip.h)
    rows = [0, 2, 3, 4, 5, 6, 7]
    gold_t, gold_l = O.topk_sampling(logits.cpu()[rows].clone(), top_k=1)
    assert torch.equal(gt[rows].cpu(), gold_t) and torch.equal(gl[rows].cpu(), gold_l)


def test_generate_with_default_sampling_config():
    """ValleAR.generate with the reference's default top_k=50 (valle/config.py:48): runs on device,
    is repeatable under torch.manual_seed, beams differ, the best beam is the arg-max of the
    length-normalised sum of log-probs (valle/models/utils.py:71-88)."""
    from tests.golden import cases as C
    from tests.test_models_gpu import build
    kw, sd, utt = C.ar_generate_inputs('tiny')
    kw = dict(kw, top_k=50, temperature=1.0, num_beams=6, max_audio_len=20)
    m = build('ValleAR', kw, sd)
    utt = [u.to(DEV) for u in utt]
    torch.manual_seed(4)
    out1 = m.generate(*utt)
    lp1 = m.last_generate_stats['sum_logprobs'].clone()
    torch.manual_seed(4)
    out2 = m.generate(*utt)
    assert torch.equal(out1, out2) and out1.dtype == torch.int64 and out1.numel() == 20
    assert int(out1.min()) >= 0 and int(out1.max()) < 1024
    assert bool((lp1 < 0).all()) and lp1.unique().numel() > 1
    # re-derive the best beam from the rows of the same run
    torch.manual_seed(4)
    text = torch.cat(utt[0::2])
    rows = m.generate_batch([text] * 6, [utt[1][:, 0]] * 6)
    lps = m.last_generate_stats['sum_logprobs']
    assert len({tuple(r.tolist()) for r in rows[:, 256:].cpu()}) > 1, 'beams should differ when sampling'
    length = (rows != 1024).sum(-1)
    best = int(torch.argmax(lps / length))
    assert torch.equal(rows[best, 256:], out1)


@pytest.mark.parametrize('top_k,V', [(1, 1025), (2, 5), (50, 1025), (64, 1025), (200, 1025), (1024, 1025), (7, 2048)])
def test_fast_top_k_path_matches_the_sorted_path_in_distribution(top_k, V):
    """top_p == 1 takes the radix-select path (no sort); top_p just below 1 takes the sorted path with the
    same support.  Same support, same per-token log-probabilities, frequencies within 4 sigma."""
    from valle2_amd.utils import topk_sampling
    g = torch.Generator().manual_seed(100 + top_k)
    row = 2.0 * torch.randn(V, generator=g)
    row[3] = row[min(V - 1, 17)]                           # an exact tie somewhere
    if top_k >= 2 and V > 6:
        srt = torch.sort(row, descending=True).values
        row[5] = srt[top_k - 1]                             # and one exactly at the k-th place
    probs = expected_probs(row, top_k, 1.0, 1.0)
    dev_rows = row.to(DEV)[None].expand(N, -1).contiguous()
    tok, lp = topk_sampling(dev_rows, top_k=top_k, tok_p=1.0, seed=77)
    tok, lp = tok[:, 0].cpu(), lp.cpu()
    assert bool((probs[tok] > 0).all()), 'a token outside the top-k support was drawn'
    freq = torch.bincount(tok, minlength=V).float() / N
    sigma = torch.sqrt(probs * (1 - probs) / N)
    assert bool(((freq - probs).abs() <= 4 * sigma + 1e-4).all()), (freq - probs).abs().max()
    torch.testing.assert_close(lp, torch.log(probs[tok]), atol=1e-5, rtol=1e-5)
    # every token of the support is reachable when it is small
    if top_k <= 7:
        assert set(tok.tolist()) == set(torch.nonzero(probs > 0)[:, 0].tolist())
    tok2, lp2 = topk_sampling(dev_rows, top_k=top_k, tok_p=1.0, seed=77)
    assert torch.equal(tok2[:, 0].cpu(), tok) and torch.equal(lp2.cpu(), lp)      # seed-deterministic
