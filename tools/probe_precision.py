"""Which bf16 rounding of the perf-mode stack costs the accuracy at 24 layers?  (configs[4] NAR leg against the REAL reference's
golden logits: 4.8e-2 with every operand narrowed, against SURVEY 8c's 5e-2.)  The stack is restated in torch on the device —
fp32 products over operands that are rounded to bf16 exactly where csrc/bf16.hip rounds them — and run with every rounding on,
each one switched off in turn, and each one alone.  A diagnostic, not a product path.

    python tools/probe_precision.py
"""
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
DEV = 'cuda'
SITES = ('xn1', 'wqkv', 'q', 'k', 'v', 'p', 'attn', 'wo', 'xn2', 'w1', 'hidden', 'w2')


FMT = torch.bfloat16        # (this probe compares the two formats whatever the library was built with)
PEAK = [0.0]


def rb(t, on):
    if on:
        PEAK[0] = max(PEAK[0], float(t.abs().max()))
        return t.to(FMT).float()
    return t


def emulated_forward(transformer, x, cache, *, mode, x_len=0, x_len_dev=None, kv_len=None, embedding=None, scratch=None, x_in=None,
                     sites=frozenset(SITES)):
    from valle2_amd import engine
    cfg = transformer.hparams
    B, T, d = x.shape
    h = cfg.n_heads
    src = x_in if x_in is not None else x
    cur = src.clone()
    ada = engine.adaln_table(transformer, embedding) if cfg.norm != 'LayerNorm' else None
    keymask = None
    if kv_len is not None:
        keymask = (torch.arange(T, device=x.device)[None, :] >= kv_len[:, None].long())[:, None, None, :]
    for i, layer in enumerate(transformer.layers):
        def norm(n, which, t):
            ln = n.norm if hasattr(n, 'norm') else n
            y = F.layer_norm(t, (d,), ln.weight, ln.bias, 1e-5)
            if ada is not None:
                y = ada[i, which, 0] * y + ada[i, which, 1]
            return y
        xn = rb(norm(layer.norm1, 0, cur), 'xn1' in sites)
        qkv = xn @ rb(layer.self_attn.qkv.weight, 'wqkv' in sites).T
        q, k, v = qkv.view(B, T, 3, h, 64).permute(2, 0, 3, 1, 4)
        q, k, v = rb(q, 'q' in sites), rb(k, 'k' in sites), rb(v, 'v' in sites)
        out = torch.empty(B, h, T, 64, device=x.device)
        for b0 in range(B):
            for h0 in range(h):
                s = (q[b0, h0] @ k[b0, h0].T) * 0.125
                if keymask is not None:
                    s = s.masked_fill(keymask[b0, 0], float('-inf'))
                m = s.max(dim=-1, keepdim=True).values
                e = torch.exp(s - m)
                out[b0, h0] = (rb(e, 'p' in sites) @ v[b0, h0]) / e.sum(dim=-1, keepdim=True)
        attn = rb(out.permute(0, 2, 1, 3).reshape(B, T, d), 'attn' in sites)
        cur = cur + attn @ rb(layer.self_attn.out.weight, 'wo' in sites).T + layer.self_attn.out.bias
        xn2 = rb(norm(layer.norm2, 1, cur), 'xn2' in sites)
        hid = rb(F.gelu(xn2 @ rb(layer.ffn.linear_1.weight, 'w1' in sites).T + layer.ffn.linear_1.bias), 'hidden' in sites)
        cur = cur + hid @ rb(layer.ffn.linear_2.weight, 'w2' in sites).T + layer.ffn.linear_2.bias
    x.copy_(cur)
    return x


def main():
    from tests.golden import cases as C
    from tests.oracle_runners import load_golden
    from valle2_amd import ConfigValle, get_model_class, valle_nar
    torch.backends.cuda.matmul.allow_tf32 = False
    gold = load_golden('nar_big')
    kw, sd, batch = C.nar_big_inputs()
    m = get_model_class('ValleNAR')(C.cfg_of(kw))
    m.load_state_dict(sd)
    m = m.to(DEV).eval()

    def run(sites, stage=7):
        orig = valle_nar.transformer_forward_bf16
        valle_nar.transformer_forward_bf16 = lambda *a, **k: emulated_forward(*a, sites=frozenset(sites), **k)
        try:
            with torch.no_grad():
                logits, _ = m.stage_logits(batch, stage, perf_mode=True)
        finally:
            valle_nar.transformer_forward_bf16 = orig
        return float((logits[:, ::C.NAR_BIG_STRIDE].cpu() - gold[f'logits_{stage}']).abs().max())

    with torch.no_grad():
        real, _ = m.stage_logits(batch, 7, perf_mode=True)
    print(f'the kernels: max |logit error| = {float((real[:, ::C.NAR_BIG_STRIDE].cpu() - gold["logits_7"]).abs().max()):.3e}')
    print(f'emulation, nothing narrowed: {run(()):.3e}')
    print(f'emulation, everything narrowed: {run(SITES):.3e}')
    for s in SITES:
        print(f'   all but {s:7s}: {run([t for t in SITES if t != s]):.3e}      only {s:7s}: {run([s]):.3e}', flush=True)
    global FMT
    FMT = torch.float16
    PEAK[0] = 0.0
    print(f'the same stack with fp16 in place of bf16 at every site: {run(SITES):.3e} (stage 7), {run(SITES, 2):.3e} (stage 2); '
          f'largest magnitude narrowed: {PEAK[0]:.1f} (fp16 holds 65504)')
    FMT = torch.bfloat16
    print(f'bf16 again, stage 2: {run(SITES, 2):.3e}')
    groups = {'weights': ('wqkv', 'wo', 'w1', 'w2'), 'scores (q, k)': ('q', 'k'), 'norm outputs': ('xn1', 'xn2'),
              'attention values (v, p, attn)': ('v', 'p', 'attn')}
    for name, g in groups.items():
        print(f'   all but {name}: {run([t for t in SITES if t not in g]):.3e}      only {name}: {run(g):.3e}', flush=True)


if __name__ == '__main__':
    main()
