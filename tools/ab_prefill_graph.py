"""Is the perf-mode stack forward faster as a replayed hipGraph than as the C composite's back-to-back launches?  (The composite issues
its ~85 kernels from one C call; what a graph can remove is the dispatch gap between dependent kernels.)  configs[1] prompt pass
(32 x 1024, prefix mask) and configs[2] NAR stage (64 x 1024, AdaLN, full mask); fp32 stack alongside.  tools/ab_prefill_graph.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from valle2_amd import ConfigValle, get_model_class, synth  # noqa: E402
from valle2_amd import kernels as K  # noqa: E402
from valle2_amd.engine import (ForwardScratch, ForwardScratch16, KVCache, transformer_forward,  # noqa: E402
                               transformer_forward_bf16)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = 'cuda'
    kw = dict(d_model=512, n_heads=8, dim_feedforward=2048, num_layers=12, dropout=0.0)
    for name, B, norm, mode, xl, mname in (('prompt pass 32x1024', 32, 'LayerNorm', K.MASK_PREFIX, 256, 'ValleAR'),
                                           ('NAR stage 64x1024', 64, 'AdaptiveLayerNorm', K.MASK_FULL, 0, 'ValleNAR')):
        cfg = ConfigValle(**kw, norm=norm)
        m = get_model_class(mname)(cfg)
        m.load_state_dict(synth.make_state_dict(cfg, mname, seed=0, rich=False))
        m = m.to(dev).eval()
        T, d = 1024, 512
        x0 = torch.randn(B, T, d, device=dev)
        emb = m.stage_embs[0].weight if mname == 'ValleNAR' else None
        for perf in (True, False):
            cache = KVCache(cfg.num_layers, B, cfg.n_heads, T, dev, dtype=K.H16 if perf else torch.float32)
            scratch = (ForwardScratch16 if perf else ForwardScratch)(B * T, d, cfg.dim_feedforward, dev)
            fwd = transformer_forward_bf16 if perf else transformer_forward
            x = x0.clone()

            def run():
                x.copy_(x0)
                fwd(m.transformer, x, cache, mode=mode, x_len=xl, embedding=emb, scratch=scratch)
            with torch.inference_mode():
                t_eager = timeit(run)
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    run()
                torch.cuda.current_stream().wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    run()
                t_graph = timeit(graph.replay)
                t_copy = timeit(lambda: x.copy_(x0))
            print(f'{name} {"perf mode" if perf else "fp32     "}: eager {t_eager:7.3f} ms | graph replay {t_graph:7.3f} ms | graph / eager {t_graph / t_eager:.3f} '
                  f'(the x copy inside both: {t_copy:.3f} ms)', flush=True)
            del graph, cache, scratch
        del m


if __name__ == '__main__':
    main()
