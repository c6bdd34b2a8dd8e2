"""Does a K/V stream that was just read come back faster (Infinity Cache, 256 MB memory-side)?
Times the decode-attention kernel (B=32, h=8) at several context lengths: 'cold' rotates over 12 layer caches
(the decode step's pattern: each launch streams memory not touched for 11 launches), 'warm' re-reads one cache,
'prefetched' reads each rotating cache once with a plain streaming kernel (torch sum) right before the launch.
Developer probe (round 2): sizes the gain a KV prefetch during the GEMM chain could have."""
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from valle2_amd import kernels as K  # noqa: E402

DEV = 'cuda'


def main():
    B, h, S_max = 32, 8, 1536
    q = torch.randn(B, 512, device=DEV)
    out = torch.empty(B, 512, device=DEV)
    caches = [(torch.randn(B, h, S_max, 64, device=DEV), torch.randn(B, h, S_max, 64, device=DEV)) for _ in range(12)]
    for S in (512, 768, 1024, 1280, 1536):
        cl = torch.full((B,), S - 1, device=DEV, dtype=torch.int32)
        mb = 2 * B * S * 512 * 4 / 1e6
        res = {}
        for mode in ('cold', 'warm', 'prefetched'):
            ts = []
            for rnd in range(5):
                evs = []
                for i in range(24):
                    kc, vc = caches[0] if mode == 'warm' else caches[i % 12]
                    if mode == 'prefetched':
                        (kc[:, :, :S].sum() + vc[:, :, :S].sum())          # stream it once (allocates in the cache?)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    K.attn_decode(q, kc, vc, out, cl, 1, 1, None)
                    e1.record()
                    evs.append((e0, e1))
                torch.cuda.synchronize()
                ts += [a.elapsed_time(b) * 1e3 for a, b in evs[4:]]
            res[mode] = statistics.median(ts)
        print(f'S={S:5d} ({mb:6.1f} MB): ' + '  '.join(f'{m} {t:6.2f} us ({mb / t * 1e-3:5.2f} TB/s)' for m, t in res.items()),
              flush=True)


if __name__ == '__main__':
    main()
