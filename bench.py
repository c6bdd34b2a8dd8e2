"""Headline benchmark: acoustic tokens/s of AR greedy decoding on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch = one full greedy `generate` of
BASELINE.json configs[1] per GPU: 12-layer/512-dim AR decoder, 32 rows, 1024-token prompt
(256 text + BOS + 767 codec tokens) → 512 new tokens, fp32, synthetic tokens and seeded
random-init weights, inputs resident in HBM before the timed region.  value = new tokens of all
ranks / wall time (prefill + 511 graph-replayed decode steps included).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: the utterance batch shards across ranks (32 rows per GPU, weights replicated, no
data-path collective: SURVEY.md §8e "inference: replicas only"); ranks meet at a barrier before and
after the timed region and the slowest rank's time is used (RCCL all-reduce MAX).

`python bench.py --gpus N` with no WORLD_SIZE in the environment starts the N ranks itself (one child
process per GPU, RCCL rendezvous on 127.0.0.1) BEFORE this process touches the GPU — the parent never
initialises HIP and never execs; under `torch.distributed.run` the environment's ranks are used as is.

Extra objects on the same JSON line: `roofline` (decode-attention kernel, HIP-event timed, HBM
bound), `prefill` (event-timed prompt pass against the fp32 MFMA peak), `decode_step` (one replayed
decode step against the HBM peak, all algorithmic bytes of the step), `train` (configs[3] forward +
backward + gradient all-reduce + clip/AdamW step time, AR and NAR, on every rank: the RCCL leg),
`cpu_baseline` (the oracle timed on this box's host cores on a bounded sample; rank 0, N=1 only),
`nar` (configs[2]: one NAR stage forward, B S / t_stage, and `all_stages`, the seven stages as generate() chains them,
B T_target 7 / t_all — SURVEY 8d's two forms of the metric), `beams` (the reference's own signature:
`generate()` of ONE utterance with num_beams=32, timed beside the 32-distinct-utterances headline),
`config5` (configs[4]'s AR leg: 24L/1024d, 8 rows, decode at context ~2.7 k against the HBM peak).
`roofline.traffic` is measured in this run: after the timed region a FRESH child process runs one
generate under `rocprofv3 --pmc FETCH_SIZE` (and one under `--pmc WRITE_SIZE`) and the decode-attention
kernel's counters are read from its CSV (the committed constant under profiles/ is the fallback).
The optional legs run under a deadline: if one stalls, the line is printed with what was measured and
the process exits.
"""
from __future__ import annotations

import argparse
import csv
import glob
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import threading
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TF = 157.3   # fp32 MFMA dense peak
MFMA_BF16_PEAK_TF = 2500.0 # bf16 / fp16 MFMA DENSE peak (MI355X_MICROARCH.md: ~2.5 PF, the F16 forms take the same cycles; the 5 PF
                           # headline figure is with 2:1 sparsity).  The 16-bit operand format of perf mode is the library's (`h16_name`).

AR = dict(d_model=512, n_heads=8, dim_feedforward=2048, num_layers=12, dropout=0.0, norm='LayerNorm',
          top_k=1, use_kv_cache=True)
ROWS, TEXT, FRAMES, NEW = 32, 256, 767, 512
NAR_B, NAR_TEXT, NAR_FRAMES = 64, 256, 768


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-nar', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-train', action='store_true')
    ap.add_argument('--no-beams', action='store_true')
    ap.add_argument('--no-config5', action='store_true')
    ap.add_argument('--no-perf-mode', action='store_true')
    ap.add_argument('--no-rows64', action='store_true')
    ap.add_argument('--no-default-generate', action='store_true')
    ap.add_argument('--no-traffic', action='store_true', help='do not start the rocprofv3 --pmc child processes')
    ap.add_argument('--traffic-child', action='store_true', help=argparse.SUPPRESS)   # one generate, nothing else
    ap.add_argument('--extras-deadline', type=float, default=600.0,
                    help='seconds the optional legs (roofline, nar, train, cpu_baseline) may take in total')
    ap.add_argument('--small', action='store_true', help='tiny shapes for a functional check')
    ap.add_argument('--allow-partial-group', action='store_true',
                    help='N > 1: print the line even if RCCL did not carry all N ranks through an all-reduce (default: exit 3)')
    return ap.parse_args()


def log(msg):
    print(f'[bench {time.strftime("%H:%M:%S")}] {msg}', file=sys.stderr, flush=True)


def host_cores():
    """CPU cores this process may really use: cgroup quota, else affinity (os.cpu_count() reports
    the whole host on a shared box and oversubscribing torch's pool stalls for minutes)."""
    n = len(os.sched_getaffinity(0))
    for quota_f, period_f in (('/sys/fs/cgroup/cpu.max', None),
                              ('/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpu/cpu.cfs_period_us')):
        try:
            if period_f is None:
                quota, period = Path(quota_f).read_text().split()
            else:
                quota, period = Path(quota_f).read_text().strip(), Path(period_f).read_text().strip()
            if quota not in ('max', '-1'):
                n = min(n, max(1, int(int(quota) / int(period))))
        except (OSError, ValueError):
            pass
    # Where neither a cgroup quota nor an affinity mask says how many cores are this process's (os.cpu_count() = the
    # whole host), cap at the share a GPU box gives one GPU — VALLE2_CORES_PER_GPU, 16 on this pool (driver note) —
    # oversubscribing torch's pool stalled the baseline for minutes
    try:
        import torch
        n = min(n, int(os.environ.get('VALLE2_CORES_PER_GPU', '16')) * max(1, torch.cuda.device_count()))
    except Exception:
        pass
    return n


def attn_algorithmic_bytes(rows, d_model, s0, new):
    """Mean algorithmic HBM bytes of one decode-attention launch over the decode steps 1..new-1:
    every K and V element of the row's context read once + q read + out written (SURVEY.md §8d:
    2*B*S*d elements per layer per step, e=4)."""
    lens = [s0 + t for t in range(1, new)]          # keys attended at step t (incl. the new one)
    mean_s = sum(lens) / len(lens)
    return 4.0 * (2 * rows * mean_s * d_model + 2 * rows * d_model), mean_s


def cpu_baseline(cfg_kw, sd, utt, rows, new, gpu_tokens=None):
    """Oracle (CPU restatement, proven bit-identical to the reference on the golden fixtures) on
    this host's cores: prefill once, then 16 decode steps; extrapolated to `new` tokens.  The 17 tokens
    it produces are also compared with what the HIP path generated for the same utterance (full-size
    parity check; steps whose top-1/top-2 margin is below 1e-4 do not count as mismatches)."""
    import torch

    from oracle import valle_oracle as O          # checker/baseline only (never the product path)
    from valle2_amd import ConfigValle
    torch.set_num_threads(host_cores())
    times = {}
    for n in (1, 17):
        log(f'cpu_baseline: oracle generate with {n} step(s) on {torch.get_num_threads()} threads')
        cfg = ConfigValle(**dict(cfg_kw, num_beams=rows, max_audio_len=n))
        trace = {} if n == 17 else None
        t0 = time.perf_counter()
        toks = O.ar_generate(sd, cfg, *utt, trace=trace)
        times[n] = time.perf_counter() - t0
    t_prefill, t_step = times[1], (times[17] - times[1]) / 16
    total = t_prefill + (new - 1) * t_step
    parity = None
    if gpu_tokens is not None:
        ref = [int(t) for t in toks][:17]
        got = [int(t) for t in gpu_tokens[:len(ref)]]
        margins = [float(m) for m in (trace or {}).get('margin', [])] + [float('inf')] * len(ref)
        bad = [i for i, (a, b) in enumerate(zip(ref, got)) if a != b and margins[i] > 1e-4]
        parity = {'tokens_compared': len(ref), 'mismatches': len(bad),
                  'note': 'HIP greedy tokens of utterance 0 vs the oracle at full size'}
        log(f'cpu_baseline: full-size greedy parity on {len(ref)} tokens: {len(bad)} mismatches')
    return {'value': rows * new / total, 'unit': 'tokens/s', 'cores': torch.get_num_threads(),
            'kind': 'port', 'parity': parity,
            'sample': f'oracle ar_generate, same shapes: 1 prefill ({t_prefill:.2f} s) + 16 decode steps '
                      f'({t_step * 1e3:.1f} ms/step at S~{TEXT + FRAMES + 1}), extrapolated to {new} tokens'}


_CHILDREN = []      # Popen handles of profiler children in flight: the extras watchdog ends them (by PID) before it exits


def _run_child(cmd, timeout, **kw):
    """subprocess.run(check=True) whose process (its own session: the profiler starts python3 under it) is on record
    while it runs, so that a deadline can end the whole group instead of orphaning it on the GPU."""
    proc = subprocess.Popen(cmd, start_new_session=True, **kw)
    _CHILDREN.append(proc)
    try:
        rc = proc.wait(timeout=timeout)
    except subprocess.TimeoutExpired:
        _kill_group(proc)
        raise
    finally:
        if proc in _CHILDREN:
            _CHILDREN.remove(proc)
    if rc != 0:
        raise subprocess.CalledProcessError(rc, cmd)


def _kill_group(proc):
    import signal
    try:
        os.killpg(proc.pid, signal.SIGKILL)           # the exact process group this script started
    except (ProcessLookupError, PermissionError):
        pass
    try:
        proc.wait(timeout=10)
    except Exception:                                  # noqa: BLE001
        pass


def measure_attn_traffic(rows, new, budget_s=480.0):
    """HBM bytes per decode-attention launch from PMC counters, measured NOW: a fresh child process per counter
    (`rocprofv3 --kernel-trace --pmc <counter> -- python3 bench.py --traffic-child`, one configs[1] generate, nothing
    else; separate passes, no tracing domain beside --kernel-trace) — never this process re-executed.  Units and the
    gfx950 correction as MI355X_MICROARCH.md section HBM prescribes: both counters are in KB; FETCH_SIZE counts a wide
    coalesced read at 1/2 (128-B requests tallied at 64 B) -> doubled; WRITE_SIZE is exact.  Returns a dict or None."""
    rocprof = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not Path(rocprof).exists() or any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ):
        return None                                       # no profiler here, or this run is itself being profiled
    out = {}
    t_end = time.monotonic() + budget_s
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        left = t_end - time.monotonic()
        if left < 60:                                     # what remains of the extras deadline cannot hold a pass
            log('traffic: not enough of the extras deadline left for a PMC pass')
            return None
        tmp = tempfile.mkdtemp(prefix='vh_pmc_')
        env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR',
                                                                'MASTER_PORT', 'LOCAL_WORLD_SIZE')}
        env['TMPDIR'] = tmp
        cmd = [rocprof, '--kernel-trace', '--pmc', counter, '-d', tmp, '-o', 'pmc', '--output-format', 'csv', '--',
               sys.executable, str(Path(__file__).resolve()), '--traffic-child']
        log(f'traffic: {" ".join(cmd[:6])} ... (child process)')
        try:
            _run_child(cmd, min(240.0, left), cwd=tmp, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            vals = []
            for path in glob.glob(f'{tmp}/**/*counter_collection.csv', recursive=True):
                for r in csv.DictReader(open(path)):
                    if 'attn_decode_ring_kernel' in r['Kernel_Name'] and r['Counter_Name'] == counter:
                        vals.append(float(r['Counter_Value']))
            if len(vals) != (new - 1) * AR['num_layers']:
                log(f'traffic: {len(vals)} {counter} rows for the decode-attention kernel, expected '
                    f'{(new - 1) * AR["num_layers"]}')
                return None
            out[counter] = sum(vals) / len(vals)
        except Exception as e:                              # noqa: BLE001 — the committed constant is the fallback
            log(f'traffic: child failed ({type(e).__name__}: {e})')
            return None
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return {'bytes_per_launch': 2.0 * out['FETCH_SIZE'] * 1024 + out['WRITE_SIZE'] * 1024,
            'fetch_size_kb': out['FETCH_SIZE'], 'write_size_kb': out['WRITE_SIZE'],
            'source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, two fresh child processes of this run, '
                      'mean over every decode-attention launch of one generate; bytes = 2 x FETCH_SIZE KB (gfx950 tallies '
                      'a 128-B request at 64 B) + WRITE_SIZE KB'}


def training_flop(cfg, model_name, batch, padded=False):
    """Algorithmic FLOP of one training step = 3 x forward (forward + 2 x backward) over the REAL
    (unpadded) positions of the batch: GEMMs 2 P_L per position and layer, attention 4 d per visible (query, key) pair
    and layer (prefix-LM for AR: text block + causal audio; full for NAR), the head over the predicted positions.
    padded=True: the same rule over the batch as the reference's semantics make it run — every row at the batch's
    maximum lengths (its loss averages over the pad positions too, valle_ar.py:86, so their logits are computed)."""
    d, dff, L = cfg.d_model, cfg.dim_feedforward, cfg.num_layers
    p_l = 4 * d * d + 2 * d * dff
    fwd = 0.0
    # NAR: ONE prefix for the whole batch, taken from the PADDED frame count as the model does (prefix_len_of(codes.shape[1]),
    # the reference's _prepare_audio_codes, valle_nar.py:179): a row's predicted positions are its real frames beyond it
    t_pad = int(batch['codes'].shape[1])
    prefix = min(t_pad // 3, 3 * cfg.quantization_factor)
    for b in range(batch['tokens'].shape[0]):
        x, y = int(batch['tokens_lens'][b]), int(batch['codes_lens'][b])
        if padded:
            x, y = int(batch['tokens_lens'].max()), t_pad if model_name != 'ValleAR' else int(batch['codes_lens'].max())
        if model_name == 'ValleAR':
            pairs = x * x + y * x + y * (y + 1) // 2
            head = 2.0 * d * (cfg.num_audio_tokens + 1) * y
        else:
            pairs = (x + y) * (x + y)           # (real positions; the reference masks no padding here, D6: `padded` counts that)
            head = 2.0 * d * cfg.num_audio_tokens * max(0, y - prefix)
        fwd += 2.0 * L * p_l * (x + y) + 4.0 * d * pairs * L + head
    return 3.0 * fwd


def config5_leg(dev):
    """BASELINE.json configs[4], AR leg: 24L/1024d/h16/dff4096, 8 rows (num_beams), 400 text + 2250-frame context —
    the END of a 30 s utterance, where the KV stream is longest — 256 greedy tokens; the decode steps against the HBM
    peak with SURVEY 8d's algorithmic bytes (weights + head once per step, every K/V element once per step)."""
    import torch

    from valle2_amd import ConfigValle, get_model_class, synth
    log('config5: 24L/1024d AR decode at context ~2.7 k')
    rows, text, frames, new = 8, 400, 2250, 256
    cfg = ConfigValle(d_model=1024, n_heads=16, dim_feedforward=4096, num_layers=24, dropout=0.0, norm='LayerNorm',
                      num_beams=rows, top_k=1, max_audio_len=new)
    sd = synth.silence_eos(synth.make_state_dict(cfg, 'ValleAR', seed=0, rich=False), cfg)
    m = get_model_class('ValleAR')(cfg)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    utts = [synth.synth_utterance(cfg, text // 2, text // 2, frames, seed=7 + u) for u in range(rows)]
    texts = [torch.cat([u[0], u[2]]).to(dev) for u in utts]
    firsts = [u[1][:, 0].to(dev) for u in utts]
    m.generate_batch(texts, firsts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 2
    for _ in range(reps):
        out = m.generate_batch(texts, firsts)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    assert out.shape[1] == frames + 1 + new
    st = m.last_generate_stats
    d, L = cfg.d_model, cfg.num_layers
    l_pl = 4 * d * d + 2 * d * cfg.dim_feedforward
    elems = [L * l_pl + (cfg.num_audio_tokens + 1) * d + 2 * L * rows * (st['s0'] + t) * d + 2 * L * rows * d
             for t in range(1, new)]
    dec_bytes = 4.0 * sum(elems)
    gbs = dec_bytes / (st['decode_ms'] * 1e-3) / 1e9
    # ---- the joint run of the configuration as BASELINE.json words it: 8 whole 30 s utterances (2250 frames at 75 Hz):
    # AR first codebook from a 225-frame acoustic prompt (context 626 -> 2875), then the NAR leg below on ITS output
    prompt_frames, target = 225, 2250
    j_texts = texts
    j_firsts = [f[:prompt_frames] for f in firsts]
    m.generate_batch(j_texts, j_firsts, max_new=64)                  # warm this shape's allocations
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    j_out = m.generate_batch(j_texts, j_firsts, max_new=target)
    torch.cuda.synchronize()
    dt_ar_full = time.perf_counter() - t0
    assert j_out.shape[1] == prompt_frames + 1 + target, 'EOS must not end the joint run early'
    st_j = m.last_generate_stats
    del m
    torch.cuda.empty_cache()
    res = {'workload': 'configs[4] AR leg: 24L/1024d/h16/dff4096 greedy generate_batch, 8 rows, 400 text + BOS + 2250 '
                       'codec tokens -> 256 new tokens (context 2651..2907), fp32',
           'value': rows * new / dt, 'unit': 'tokens/s', 'ms_per_generate': dt * 1e3, 'prefill_ms': st['prefill_ms'],
           'decode_ms_per_step': st['decode_ms'] / (new - 1), 'algorithmic_bytes_total': dec_bytes,
           'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit_bw': 'GB/s', 'frac': gbs / HBM_PEAK_GBS, 'bound': 'hbm',
           'n_split': st['n_split']}
    # ---- NAR leg of the same configuration: the 7 remaining codebooks of a whole 30 s utterance per row (valle_nar.py:107-165):
    # 400 text + 225-frame acoustic prompt + 2250 target frames = 2875 positions per row, one 24-layer forward per stage
    log('config5: 24L/1024d NAR, 7 stages over 8 x 2875 positions')
    ncfg = ConfigValle(d_model=1024, n_heads=16, dim_feedforward=4096, num_layers=24, dropout=0.0,
                       norm='AdaptiveLayerNorm')
    nsd = synth.make_state_dict(ncfg, 'ValleNAR', seed=0, rich=True)
    nar = get_model_class('ValleNAR')(ncfg)
    nar.load_state_dict(nsd)
    nar = nar.to(dev).eval()
    g = torch.Generator().manual_seed(11)
    n_texts = [torch.randint(0, ncfg.vocab_size, (text,), generator=g).to(dev) for _ in range(rows)]
    n_prompts = [torch.randint(0, ncfg.num_audio_tokens, (prompt_frames, ncfg.num_quantizers), generator=g).to(dev)
                 for _ in range(rows)]
    n_firsts = [j_out[b, prompt_frames + 1:].clamp(max=ncfg.num_audio_tokens - 1) for b in range(rows)]   # the AR run's codes
    nar.generate_batch(n_texts, n_prompts, n_firsts, greedy=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    codes = nar.generate_batch(n_texts, n_prompts, n_firsts, greedy=True)
    torch.cuda.synchronize()
    dtn = time.perf_counter() - t0
    assert len(codes) == rows and tuple(codes[0].shape) == (target, ncfg.num_quantizers)
    stages = ncfg.num_quantizers - 1
    pos = text + prompt_frames + target
    flop = stages * (2.0 * L * l_pl * rows * pos + 4.0 * d * pos * pos * L * rows
                     + 2.0 * d * ncfg.num_audio_tokens * rows * target)
    res['nar'] = {'workload': 'configs[4] NAR leg: 24L/1024d AdaLN stack, 8 rows x (400 text + 225-frame prompt + 2250 target '
                              'frames), codebooks 2..8 (7 stages, greedy), fp32',
                  'ms_total': dtn * 1e3, 'ms_per_stage': dtn * 1e3 / stages,
                  'value': rows * target * stages / dtn, 'unit': 'codec tokens/s',
                  'flop': flop, 'tflops': flop / dtn / 1e12, 'peak_tflops': MFMA_F32_PEAK_TF,
                  'frac': flop / dtn / 1e12 / MFMA_F32_PEAK_TF, 'bound': 'mfma'}
    # the same seven stages in PERF MODE (SECONDARY, never the metric): bf16 operands on the bf16 matrix cores (csrc/bf16.hip)
    nar.generate_batch(n_texts, n_prompts, n_firsts, greedy=True, perf_mode=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    codes16 = nar.generate_batch(n_texts, n_prompts, n_firsts, greedy=True, perf_mode=True)
    torch.cuda.synchronize()
    dt16 = time.perf_counter() - t0
    same = float(torch.stack([(a == b).float().mean() for a, b in zip(codes, codes16)]).mean())
    res['nar']['perf_mode'] = {'label': f'SECONDARY, not the metric: the seven stage forwards with {h16_name()} operands / fp32 accumulators',
                               'ms_total': dt16 * 1e3, 'ms_per_stage': dt16 * 1e3 / stages, 'tflops': flop / dt16 / 1e12,
                               'mfma_bf16_peak_tflops': MFMA_BF16_PEAK_TF, 'frac_of_bf16_peak': flop / dt16 / 1e12 / MFMA_BF16_PEAK_TF,
                               'vs_f32': dtn / dt16, 'greedy_codes_equal_to_f32_run': same}
    audio_s = rows * target / 75.0
    res['joint'] = {'workload': 'configs[4] as worded: 8 utterances of 30 s (2250 frames x 8 codebooks), 400 text + 225-frame prompt: '
                                'AR generate of codebook 1 (2250 steps, context 626 -> 2875) + the NAR leg above on its output',
                    'ar_ms': dt_ar_full * 1e3, 'ar_prefill_ms': st_j['prefill_ms'],
                    'ar_decode_ms_per_step': st_j['decode_ms'] / (target - 1), 'nar_ms': dtn * 1e3,
                    'total_ms': (dt_ar_full + dtn) * 1e3, 'audio_seconds': audio_s,
                    'value': rows * target * ncfg.num_quantizers / (dt_ar_full + dtn), 'unit': 'codec tokens/s (8 codebooks)',
                    'x_realtime': audio_s / (dt_ar_full + dtn)}
    del nar
    torch.cuda.empty_cache()
    return res


def spawn_ranks(n):
    """`bench.py --gpus N` without a launcher: start N ranks of this script, one per GPU.  Runs before
    anything in this process has touched the GPU (torch is not even imported here); the parent only
    waits, forwards the exit code and — if a rank dies — ends the others (by PID, not by pattern)."""
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env))
    rc = 0
    while procs:
        time.sleep(0.2)
        for p in list(procs):
            code = p.poll()
            if code is None:
                continue
            procs.remove(p)
            if code != 0:
                rc = rc or code
                log(f'rank process {p.pid} exited with {code}: stopping the other ranks')
                for q in procs:
                    q.terminate()
    return rc


def default_generate_leg(dev, sd, utt):
    import torch

    from valle2_amd import ConfigValle, get_model_class
    cfg = ConfigValle(d_model=512, n_heads=8, dim_feedforward=2048, num_layers=12, dropout=0.0, norm='LayerNorm')
    assert (cfg.num_beams, cfg.top_k, cfg.tok_p, cfg.temperature, cfg.max_audio_len) == (4, 50, 1.0, 1.0, 1024)
    model = get_model_class('ValleAR')(cfg)
    model.load_state_dict(sd)                       # the headline's weights (EOS row zeroed: logit 0 never reaches the top 50)
    model = model.to(dev).eval()
    utt = [u.to(dev) for u in utt]
    torch.manual_seed(0)
    model.generate(*utt)
    torch.cuda.synchronize()
    reps, t0 = 3, time.perf_counter()
    for _ in range(reps):
        toks = model.generate(*utt)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    st = model.last_generate_stats
    steps = st['steps_run']
    attn_launches = (2 + (1 if st['n_split'] > 1 else 0)) if st.get('shared_prompt') else (2 if st['n_split'] > 1 else 1)
    per_layer = 2 + attn_launches + (2 if st['ffn_fused'] else 3)
    out = {'metric': 'generate(prompt_tokens, prompt_codes, target_tokens) at the reference generation defaults '
                     '(num_beams=4, top_k=50, tok_p=1.0, temperature=1.0, max_audio_len=1024), 12L/512d, '
                     f'{utt[0].numel() + utt[2].numel()} text + {utt[1].shape[0]} prompt frames',
           'value': cfg.num_beams * steps / dt, 'unit': 'tokens/s (all 4 beams)',
           'utterance_tokens_per_s': steps / dt, 'ms_per_generate': dt * 1e3, 'steps': steps,
           'tokens_returned': int(toks.numel()), 'prefill_ms': st['prefill_ms'],
           'decode_us_per_step': st['decode_ms'] / max(1, steps - 1) * 1e3, 'n_split': st['n_split'],
           'launches_per_step': cfg.num_layers * per_layer + 2,
           'shared_prompt': bool(st.get('shared_prompt')),
           # round 6: the decoder of a shape (graphs, caches, counters) survives generate(): what the timed calls spent on the
           # host outside enqueueing the prompt pass and the replays (~1.7 ms of capture + construction before)
           'decoder_reused': bool(st.get('decoder_reused')), 'host_outside_ms': st.get('host_outside_ms'),
           'host_decoder_ms': st.get('host_decoder_ms'),
           'launches': 'per layer: QKV, '
                       + (f'shared-prompt attention (prefix launch + suffix x{st["n_split"]} key splits'
                          if st.get('shared_prompt') else f'decode attention (x{st["n_split"]} key splits')
                       + (' + combine)' if st['n_split'] > 1 else ')') + ', out-projection, FeedForward '
                       + ('(one split launch + slab reduce)' if st['ffn_fused'] else '(linear_1, split-K linear_2, reduce)')
                       + '; + head + sample step'}
    del model
    torch.cuda.empty_cache()
    return out


def train_leg(dev, world, rank, small):
    """configs[3]: 12L/512d AR and NAR forward + backward + gradient all-reduce (RCCL when world > 1,
    overlapped with backward) + clip/AdamW, per-GPU batch 16, LibriTTS-shaped ragged synthetic batches."""
    import torch

    from valle2_amd import ConfigValle, dp, get_model_class, synth
    out = {'config': 'configs[3]: 12L/512d fwd+bwd, per-GPU batch 16, tokens 40..120, codes 225..900, fp32 grads, '
                     f'DP x{world}' + (' (RCCL flat-bucket all-reduce overlapped with backward)' if world > 1 else ''),
           'bound': 'mfma', 'peak_tflops': MFMA_F32_PEAK_TF,
           'flop_rule': '3 x forward (fwd + 2 x bwd) over the real, unpadded positions of this rank: GEMMs 2 P_L per '
                        'position and layer + attention 4 d per visible pair and layer + head (SURVEY 8d); NAR head positions = '
                        "a row's real frames beyond the batch's ONE prefix min(T_pad // 3, 150) (the model's rule, from the "
                        "padded frame count) in both variants; *_frac = this rank's TFLOP/s / 157.3"}
    kw = dict(d_model=512, n_heads=8, dim_feedforward=2048, num_layers=12, dropout=0.0, batch_size=16)
    if small:
        kw.update(d_model=128, n_heads=2, dim_feedforward=512, num_layers=2, batch_size=4)
    out['dropout_rule'] = ('*_ms_per_step: config.dropout = 0 (PositionalEncoding dropout 0.1 still live in train mode, D9); '
                           '*_ms_per_step_dropout: the reference default config.dropout = 0.1 (valle/config.py:26) — dropout1 / '
                           'dropout2 / FeedForward dropout as fields regenerated in the GEMM epilogues and the LayerNorm '
                           'backward (valle2_amd/dropout.py), same batches')
    for name, norm in (('ValleAR', 'LayerNorm'), ('ValleNAR', 'AdaptiveLayerNorm')):
      for p_drop in (0.0, 0.1):
        cfg = ConfigValle(**dict(kw, dropout=p_drop), norm=norm)
        torch.manual_seed(0)
        model = get_model_class(name)(cfg).to(dev).train()
        opt = model.configure_optimizers()['optimizer']
        reducer = dp.GradReducer(opt.flat_grad, opt.slots)
        warm, timed = 3, 5
        rows, flop, flop_pad = 0, 0.0, 0.0
        batches = []                     # every step's batch is in HBM before the clock starts (bench contract): ids on
        for i in range(warm + timed):    # the device, the length vectors on the host, as collate hands them over
            if name == 'ValleAR':
                batch = synth.synth_ar_batch(cfg, cfg.batch_size, seed=100 + i + 1000 * rank)
            else:
                batch = synth.synth_nar_batch(cfg, cfg.batch_size, n_tokens=80, n_frames=560, seed=100 + i + 1000 * rank)
            batches.append({k: (v if k.endswith('_lens') else v.to(dev)) for k, v in batch.items()})
        for i, batch in enumerate(batches):
            if i == warm:
                torch.cuda.synchronize()
                if world > 1:
                    torch.distributed.barrier()
                t0 = time.perf_counter()
            loss = model.training_step(batch, **({'stage': 1 + i % 7} if name == 'ValleNAR' else {}))
            loss.backward()
            reducer.finish()
            opt.sync_touched()
            opt.step(grad_scale=1.0 / world, max_norm=cfg.gradient_clip_val, zero_grad=True)
            if i >= warm:
                rows += batch['codes'].shape[0] * (batch['codes'].shape[1] + batch['tokens'].shape[1])
                flop += training_flop(cfg, name, batch)
                flop_pad += training_flop(cfg, name, batch, padded=True)
        torch.cuda.synchronize()
        dt = dp.max_over_ranks(time.perf_counter() - t0, dev)
        key = 'ar' if name == 'ValleAR' else 'nar'
        if not p_drop:
            # the exchange by itself (nothing to overlap with): every bucket launched back to back on the zeroed flat
            # gradient, as finish() does for a step whose hooks launched nothing — what the step would pay without overlap
            out[f'{key}_exchange'] = {'algorithm': reducer.algorithm, 'bucket_mb': reducer.bucket_bytes / 2 ** 20,
                                      'buckets': len(reducer.buckets), 'launches_per_step': reducer.launches_per_step,
                                      'bytes_sent_per_rank': reducer.bytes_per_step(world), 'alone_ms': None}
            if world > 1 and reducer.active:
                reducer.finish()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(5):
                    reducer.finish()
                torch.cuda.synchronize()
                out[f'{key}_exchange']['alone_ms'] = dp.max_over_ranks(time.perf_counter() - t1, dev) / 5 * 1e3
        if p_drop:
            out[f'{key}_ms_per_step_dropout'] = dt / timed * 1e3
            out[f'{key}_dropout_over_plain'] = (dt / timed * 1e3) / out[f'{key}_ms_per_step']
            out[f'{key}_loss_dropout'] = float(loss.detach())
        else:
            out[f'{key}_ms_per_step'] = dt / timed * 1e3
            out[f'{key}_positions_per_s'] = world * rows / dt          # rank 0's rows x world (shapes are seeded per rank)
            out[f'{key}_allreduce_bytes'] = 4 * opt.numel if world > 1 else 0
            # this rank's algorithmic FLOP per step (real positions, 3 x forward) against the fp32 MFMA peak of ONE GPU
            out[f'{key}_flop_per_step'] = flop / timed
            out[f'{key}_tflops'] = flop / dt / 1e12
            out[f'{key}_frac'] = flop / dt / 1e12 / MFMA_F32_PEAK_TF
            out[f'{key}_frac_padded_positions'] = flop_pad / dt / 1e12 / MFMA_F32_PEAK_TF   # the work the step really runs
            out[f'{key}_loss'] = float(loss.detach())
        reducer.remove()
        del model, opt, reducer
        torch.cuda.empty_cache()
    return out


def h16_name():
    """'fp16' (default build) or 'bf16' (-DVH_PERF_BF16): the 16-bit operand format of the perf-mode kernels and the narrow K/V cache."""
    import torch
    from valle2_amd._lib import h16_dtype
    return 'fp16' if h16_dtype() == torch.float16 else 'bf16'


def main():
    args = parse()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if os.environ.get('VALLE2_BENCH_SPAWN_PROBE'):     # test hook: show what a spawned rank was given, no GPU
        print(json.dumps({'rank': rank, 'local': local, 'world': world, 'addr': os.environ.get('MASTER_ADDR'),
                          'port': os.environ.get('MASTER_PORT'), 'argv': sys.argv[1:]}), flush=True)
        return
    os.chdir(tempfile.mkdtemp(prefix='vh_bench_'))
    import torch
    import torch.distributed as dist

    from valle2_amd import ConfigValle, dp, get_model_class, synth

    # rehearsal knobs (a one-GPU box, several ranks on `gloo`): VALLE2_FORCE_DEVICE, VALLE2_DIST_BACKEND
    local = int(os.environ.get('VALLE2_FORCE_DEVICE', local))
    if local >= torch.cuda.device_count():
        raise SystemExit(f'bench.py: rank {rank} wants GPU {local} but this node shows {torch.cuda.device_count()}')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    backend = os.environ.get('VALLE2_DIST_BACKEND', 'nccl')   # "nccl" is RCCL on ROCm; no-op at N=1
    try:
        dp.init_distributed(backend, dev)
    except Exception as e:                                    # the headline needs no collective: time it anyway
        log(f'rank {rank}: {backend} process group failed ({type(e).__name__}: {e}); timing over gloo, no training leg')
        if dist.is_initialized():
            dist.destroy_process_group()
        dp.init_distributed('gloo', dev)
        args.no_train = True
    # barrier and max-time reduction of the timed region run over a host-side gloo group (the inference path has no
    # collective of its own); RCCL carries the training leg's gradient all-reduce
    host_pg = dp.host_group()
    # what the default group (RCCL at N > 1) itself saw: world size from the group and a count carried by a collective
    try:
        seen = dp.ranks_seen(dev)
    except Exception as e:                                    # a communicator that cannot carry ONE all-reduce: say so, keep
        log(f'rank {rank}: all-reduce over the default group failed ({type(e).__name__}: {e}); no training leg')   # the headline
        seen = {'backend': dist.get_backend() if dist.is_initialized() else None, 'group_world_size': world,
                'allreduce_count': None, 'error': f'{type(e).__name__}: {e}'}
        args.no_train = True
    # every rank takes the same branch: a rank that alone skipped the training leg would leave the others inside its
    # RCCL all-reduce until the deadline (agreed over the host-side gloo group, which needs no GPU collective)
    if world > 1:
        flag = torch.tensor([1 if args.no_train else 0, 1 if seen.get('allreduce_count') == world else 0], dtype=torch.int32)
        lo = flag.clone()
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=host_pg)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=host_pg)
        args.no_train = bool(flag[0])
        # ... and the line is only worth printing if the default group really spans the N ranks it claims
        on_rccl = dist.get_backend() == 'nccl'
        if (not bool(lo[1]) or not on_rccl) and os.environ.get('VALLE2_DIST_BACKEND', 'nccl') == 'nccl' and not args.allow_partial_group:
            log(f'rank {rank}: the default process group carried {seen.get("allreduce_count")} of {world} ranks through an all-reduce '
                f'({seen}); refusing to print a {world}-GPU line (--allow-partial-group to time the headline anyway)')
            dist.barrier(group=host_pg)
            raise SystemExit(3)

    rows, text, frames, new = ROWS, TEXT, FRAMES, NEW
    ar_kw = dict(AR)
    if args.small:
        rows, text, frames, new = 4, 32, 63, 24
        ar_kw.update(d_model=128, n_heads=2, dim_feedforward=512, num_layers=2)
    cfg = ConfigValle(**ar_kw, num_beams=rows, max_audio_len=new)
    sd = synth.silence_eos(synth.make_state_dict(cfg, 'ValleAR', seed=0, rich=False), cfg)
    model = get_model_class('ValleAR')(cfg)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    # weak scaling: world*rows DISTINCT utterances, sharded by utterance (rows never deduplicated)
    utts = [synth.synth_utterance(cfg, text // 2, text - text // 2, frames, seed=1234 + u)
            for u in dp.shard_range(world * rows, rank, world)]
    texts = [torch.cat([u[0], u[2]]).to(dev) for u in utts]
    firsts = [u[1][:, 0].to(dev) for u in utts]

    def step():
        out = model.generate_batch(texts, firsts)
        assert out.shape[1] == frames + 1 + new, 'EOS must not end a bench run early'
        return out

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier(group=host_pg)
        torch.cuda.synchronize()

    if args.traffic_child:                 # under rocprofv3 --pmc, started by measure_attn_traffic(): one generate
        step()
        torch.cuda.synchronize()
        return
    log(f'rank {rank}/{world}: model ready, warmup {args.warmup}')
    for _ in range(args.warmup):
        step()
    log('timed region')
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    assert int((out[:, frames + 1:] == cfg.eos_token).sum()) == 0, 'EOS appeared in a bench run'
    elapsed = dp.max_over_ranks(elapsed, dev, group=host_pg)
    ms_per_step = elapsed / args.steps * 1e3
    log(f'{ms_per_step:.1f} ms per generate')
    value = world * rows * new * args.steps / elapsed

    result = {
        'metric': 'acoustic tokens/sec (AR decode)', 'value': value, 'unit': 'tokens/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
        'data': 'synthetic',
        'config': {'workload': 'configs[1]: 12-layer/512-dim AR decoder greedy generate, batch 32, '
                               '1024-token prompt -> 512 new tokens, 1xMI355X per rank'
                               if not args.small else 'SMALL functional check (not the metric)',
                   'rows_per_gpu': rows, 'prompt_tokens': text + frames + 1, 'new_tokens': new,
                   'sharding': f'utterance-batch x{world}',
                   'timed_region': 'embed + prefill + (new-1) hipGraph-replayed decode steps'},
        # proof that the N ranks of this line were N ranks of ONE RCCL communicator: the default group's backend ("nccl" =
        # RCCL on ROCm), its own world size, and the sum of one 1 per rank through a device all-reduce on it
        'distributed': dict(seen, env_world_size=world),
        'rccl_ranks_seen': seen['allreduce_count'] if seen['backend'] == 'nccl' else (1 if world == 1 else None),
    }

    # ---- optional legs under a deadline: the line is printed exactly once, by whoever gets there first
    emitted = threading.Lock()

    def emit(note=None):
        if not emitted.acquire(blocking=False):
            return
        if note:
            result['extras_note'] = note
        if rank == 0:
            print(json.dumps(result), flush=True)

    def on_deadline():
        result['legs_cut_by_deadline'] = True
        emit(f'optional legs exceeded {args.extras_deadline:.0f} s: line printed with what was measured (legs_cut_by_deadline)')
        for proc in list(_CHILDREN):                   # a PMC pass still in flight: end it, do not orphan it on the GPU
            _kill_group(proc)
        os._exit(0)                                    # the line is out and says so itself ('legs_cut_by_deadline': true)

    t_extras = time.monotonic()
    watchdog = threading.Timer(args.extras_deadline, on_deadline)
    watchdog.daemon = True
    watchdog.start()

    l_pl = 4 * cfg.d_model ** 2 + 2 * cfg.d_model * cfg.dim_feedforward   # matmul weights per layer
    st = model.last_generate_stats
    if rank == 0 and 'prefill_ms' in st:
        # phase split of the LAST timed generate (HIP events on the launch stream, recorded inside generate_batch)
        x, y = text, frames + 1
        pairs = x * x + y * x + y * (y + 1) // 2                            # unmasked (query, key) pairs per row
        flop = 2.0 * cfg.num_layers * l_pl * rows * (x + y) + 4.0 * cfg.d_model * pairs * cfg.num_layers * rows
        prefill_flop = flop
        tf = flop / (st['prefill_ms'] * 1e-3) / 1e12
        result['prefill'] = {'ms': st['prefill_ms'], 'flop': flop, 'tflops': tf, 'peak_tflops': MFMA_F32_PEAK_TF,
                             'frac': tf / MFMA_F32_PEAK_TF, 'bound': 'mfma', 'dtype': 'f32',
                             'note': 'embed + 12-layer prompt pass + head + first sample of the last timed generate; '
                                     'flop = GEMMs over all prompt rows + attention over unmasked pairs only (SURVEY 8d)'}
        step_elems = [cfg.num_layers * l_pl + (cfg.num_audio_tokens + 1) * cfg.d_model
                      + 2 * cfg.num_layers * rows * (st['s0'] + t) * cfg.d_model
                      + 2 * cfg.num_layers * rows * cfg.d_model for t in range(1, new)]
        dec_bytes = 4.0 * sum(step_elems)
        gbs = dec_bytes / (st['decode_ms'] * 1e-3) / 1e9
        result['decode_step'] = {'ms_per_step': st['decode_ms'] / max(1, new - 1), 'steps': new - 1,
                                 'algorithmic_bytes_total': dec_bytes, 'achieved': gbs, 'peak': HBM_PEAK_GBS,
                                 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS, 'bound': 'hbm',
                                 'tokens_per_s_decode_only': rows * (new - 1) / (st['decode_ms'] * 1e-3),
                                 'note': 'all hipGraph-replayed decode steps of the last timed generate (weights + KV '
                                         'read once per step, SURVEY 8d), EOS polls included'}
    if rank == 0 and not args.no_roofline:
        # dominant kernel: decode attention (KV streaming).  HIP events on the launch stream for
        # every launch of an eager (non-graph) pass over the same 511 steps.
        log('roofline: event-timed eager pass')
        model.generate_batch(texts, firsts, profile_attn=True)
        torch.cuda.synchronize()
        st = model.last_generate_stats
        bytes_per_launch, mean_s = attn_algorithmic_bytes(rows, cfg.d_model, st['s0'], new)
        # Duration of a launch = elapsed time between the start and stop events attached to the kernel's own
        # dispatch packet (hipExtLaunchKernelGGL on the launch stream): the kernel's begin/end timestamps, which is
        # also what rocprofv3 --kernel-trace reports (profiles/r1_bench_kernel_stats.md).  Kept beside it for
        # reference: a bracket of two marker events recorded around the launch (upper bound, it carries event +
        # dispatch overhead) and the same bracket with nothing inside (its floor).
        raw_s, floor_s = st['attn_mean_ms'] * 1e-3, (st.get('attn_floor_ms') or 0.0) * 1e-3
        kern_s = (st.get('attn_kernel_ms') or 0.0) * 1e-3
        dur_s = kern_s if kern_s > 0 else raw_s
        achieved = bytes_per_launch / dur_s / 1e9
        result['roofline'] = {
            'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': achieved / HBM_PEAK_GBS, 'traffic': None, 'traffic_unit': 'bytes/launch',
            'kernel': 'attn_decode_ring_kernel<8, 2> (vh_attn_decode)', 'launches': (new - 1) * cfg.num_layers,
            'avg_launch_us': dur_s * 1e6,
            'timing': 'kernel start/stop events' if kern_s > 0 else 'marker-event bracket',
            'marker_bracket_us': raw_s * 1e6, 'marker_floor_us': floor_s * 1e6,
            'algorithmic_bytes_per_launch': bytes_per_launch, 'mean_context': mean_s,
            'note': 'duration = mean over every decode-attention launch of an eager pass of the elapsed time between '
                    'the HIP start/stop events attached to that kernel dispatch on the launch stream; '
                    'marker_bracket_us = two marker events recorded around the same launch (event + dispatch overhead '
                    'included; not used), marker_floor_us = that bracket with nothing inside; traffic = PMC FETCH_SIZE/WRITE_SIZE from separate '
                    'rocprofv3 --pmc passes of this workload, committed under profiles/ (null when absent)'}
        # (the PMC children get what is left of the extras deadline, less a reserve for the legs after them)
        left = args.extras_deadline - (time.monotonic() - t_extras) - 240.0
        measured = None if (args.small or args.no_traffic or world > 1) else measure_attn_traffic(rows, new, left)
        pmc = REPO / 'profiles' / 'attn_decode_traffic.json'
        if measured:
            result['roofline']['traffic'] = measured['bytes_per_launch']
            result['roofline']['traffic_source'] = measured['source']
            result['roofline']['traffic_fetch_size_kb'] = measured['fetch_size_kb']
            result['roofline']['traffic_write_size_kb'] = measured['write_size_kb']
            result['roofline']['traffic_over_algorithmic'] = measured['bytes_per_launch'] / bytes_per_launch
            result['roofline']['traffic_measured_in_this_run'] = True
        elif pmc.exists() and not args.small:
            t = json.loads(pmc.read_text())
            if t.get('rows') == rows and t.get('new_tokens') == new:
                result['roofline']['traffic'] = t['bytes_per_launch']
                result['roofline']['traffic_source'] = t['source']
                result['roofline']['traffic_measured_in_this_run'] = False   # committed-profile data (PMC passes
                # cannot run inside the timed process: rocprofv3 --pmc serialises kernels)
        # whole-step view of the same roofline: all algorithmic bytes of the decode steps
        step_elems = [cfg.num_layers * l_pl + (cfg.num_audio_tokens + 1) * cfg.d_model
                      + 2 * cfg.num_layers * rows * (st['s0'] + t) * cfg.d_model
                      + 2 * cfg.num_layers * rows * cfg.d_model for t in range(1, new)]
        result['roofline']['decode_algorithmic_bytes_total'] = 4.0 * sum(step_elems)

    if rank == 0 and world == 1 and not args.no_beams:       # secondary objects: the N = 1 line only (N > 1 measures scaling)
        # the reference's own signature (valle_ar.py:136-138): ONE utterance, num_beams = 32 replicated rows, greedy.  The
        # beams share one prompt, so since round 5 generate() runs the prompt pass for ONE row and reads the prompt's K/V
        # once per decode step for all beams (vh_attn_decode_shared); the beams themselves (their tokens, their own K/V rows,
        # their scores) are never deduplicated.  `independent_rows`: the same call with the beams decoded as 32 independent
        # rows (round 4's form, VALLE2_SHARED_PROMPT=0), which costs what the 32-distinct-utterances headline costs.
        log('beams: generate() of one utterance with num_beams=32 (shared prompt K/V, and as independent rows)')
        utt0 = [u.to(dev) for u in utts[0]]              # inputs resident in HBM, as for the headline
        text0, first0 = torch.cat([utt0[0], utt0[2]]), utt0[1][:, 0]
        legs = {}
        for name, shared in (('shared', True), ('independent_rows', False)):
            model.generate_batch([text0] * rows, [first0] * rows, shared_prompt=shared)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                toks_b = model.generate_batch([text0] * rows, [first0] * rows, shared_prompt=shared)
            torch.cuda.synchronize()
            legs[name] = ((time.perf_counter() - t0) / reps, dict(model.last_generate_stats), toks_b)
        dtb, st_b, toks_b = legs['shared']
        dti, st_i, toks_i = legs['independent_rows']
        toks = model.generate(*utt0)                     # the reference's entry point itself (shared by default)
        assert model.last_generate_stats['shared_prompt'] and torch.equal(toks, toks_b[0, frames + 1:])
        # algorithmic bytes of the shared form's decode steps: weights + head + the prompt's K/V ONCE + every beam's own rows
        s0_b = st_b['s0']
        bytes_b = 4.0 * sum(cfg.num_layers * l_pl + (cfg.num_audio_tokens + 1) * cfg.d_model
                            + 2 * cfg.num_layers * (s0_b + rows * t) * cfg.d_model
                            + 2 * cfg.num_layers * rows * cfg.d_model for t in range(1, new))
        gbs_b = bytes_b / (st_b['decode_ms'] * 1e-3) / 1e9
        result['beams'] = {'metric': 'acoustic tokens/sec, generate(prompt_tokens, prompt_codes, target_tokens) with '
                                     f'num_beams={rows} (one utterance: prompt K/V shared by the beams, beams not deduplicated)',
                           'value': rows * new / dtb, 'unit': 'tokens/s', 'ms_per_generate': dtb * 1e3,
                           'tokens_returned': int(toks.numel()), 'decode_ms': st_b['decode_ms'],
                           'decode_us_per_step': st_b['decode_ms'] / (new - 1) * 1e3,
                           'prefill_ms': st_b['prefill_ms'],
                           'algorithmic_bytes_total': bytes_b, 'bytes_rule': '4 (L P_L + V d + 2 L (S0 + B t) d + 2 L B d) per step',
                           'achieved': gbs_b, 'peak': HBM_PEAK_GBS, 'unit_bw': 'GB/s', 'frac': gbs_b / HBM_PEAK_GBS,
                           'vs_distinct_rows': (rows * new / dtb) / (value / world),
                           'independent_rows': {'value': rows * new / dti, 'ms_per_generate': dti * 1e3,
                                                'decode_us_per_step': st_i['decode_ms'] / (new - 1) * 1e3,
                                                'prefill_ms': st_i['prefill_ms']},
                           'shared_over_independent': dti / dtb,
                           'decoder_reused': bool(st_b.get('decoder_reused')), 'host_outside_ms': st_b.get('host_outside_ms'),
                           'same_tokens_as_independent_rows': bool(torch.equal(toks_b, toks_i))}

    if rank == 0 and world == 1 and not args.no_default_generate and not args.small:
        # what a user of the reference calls: generate() with the reference's OWN generation defaults (valle/config.py:
        # num_beams = 4, top_k = 50, tok_p = 1, temperature = 1, max_audio_len = 1024) on the 12L/512d model — one utterance,
        # 4 sampled beams, 1024 steps.  B x heads = 32 (row, head) streams: the decode attention splits the keys 8 ways
        # (+ a combine launch), the GEMMs see 4 rows: a launch-floor-bound step, reported as such
        log('default_generate: generate() with the reference generation defaults')
        result['default_generate'] = default_generate_leg(dev, sd, utts[0])

    if rank == 0 and world == 1 and not args.no_perf_mode:
        # SURVEY section 7's perf mode, a LABELLED SECONDARY line (narrower storage than the reference: never the headline,
        # never `dtype`): the same generate over a bf16 K/V cache, everything else fp32
        log(f'perf_mode: the same generate over a {h16_name()} K/V cache')
        out_p = model.generate_batch(texts, firsts, perf_mode=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            out_p = model.generate_batch(texts, firsts, perf_mode=True)
        torch.cuda.synchronize()
        dtp = (time.perf_counter() - t0) / reps
        st_p = model.last_generate_stats
        kv_elems = sum(2 * cfg.num_layers * rows * (st_p['s0'] + t) * cfg.d_model for t in range(1, new))
        other = sum(cfg.num_layers * l_pl + (cfg.num_audio_tokens + 1) * cfg.d_model + 2 * cfg.num_layers * rows * cfg.d_model
                    for t in range(1, new))
        dec_bytes_p = 2.0 * kv_elems + 4.0 * other
        gbs_p = dec_bytes_p / (st_p['decode_ms'] * 1e-3) / 1e9
        result['perf_mode'] = {
            'label': f'SECONDARY, not the metric: K/V cache stored as {h16_name()} (weights, activations, softmax, accumulators fp32); '
                     'teacher-forced logits within atol 5e-2 of the reference (tests/test_fullsize_golden_gpu.py), greedy '
                     'tokens not guaranteed',
            'storage': f'{h16_name()} K/V cache, fp32 weights', 'h16_format': h16_name(), 'value': rows * new / dtp, 'unit': 'tokens/s',
            'ms_per_generate': dtp * 1e3, 'decode_ms_per_step': st_p['decode_ms'] / (new - 1),
            'algorithmic_bytes_total': dec_bytes_p, 'achieved': gbs_p, 'peak': HBM_PEAK_GBS, 'unit_bw': 'GB/s',
            'frac': gbs_p / HBM_PEAK_GBS, 'vs_f32_headline': (rows * new / dtp) / (value / world),
            'greedy_tokens_equal_to_f32_run': float((out_p == out).float().mean()),
            # the prompt pass of this mode runs on the bf16 matrix cores (round 5): same flop count as `prefill`, priced
            # against the bf16 dense peak of MI355X_MICROARCH.md (~2.5 PF/s; AMD's 5 PF figure is with 2:1 sparsity)
            'prefill': {'storage': f'{h16_name()} operands (weights narrowed once per weight set, activations in the producers\' epilogues), '
                                   'fp32 accumulators, residual stream, LayerNorm statistics, softmax',
                        'ms': st_p['prefill_ms'], 'prefill_bf16': bool(st_p.get('prefill_bf16')),
                        'tflops': prefill_flop / (st_p['prefill_ms'] * 1e-3) / 1e12, 'mfma_bf16_peak_tflops': MFMA_BF16_PEAK_TF,
                        'frac_of_bf16_peak': prefill_flop / (st_p['prefill_ms'] * 1e-3) / 1e12 / MFMA_BF16_PEAK_TF,
                        # (against the headline run's prompt pass — `st` has been overwritten by the roofline leg's eager pass by now)
                        'vs_f32_prefill': (result['prefill']['ms'] / st_p['prefill_ms']) if 'prefill' in result else None}}

    if rank == 0 and world == 1 and not args.no_rows64 and not args.small:
        # A LABELLED SECONDARY line: the same fp32 generate with TWICE the rows BASELINE.json names (64 distinct utterances,
        # the most one decode launch serves).  The GEMM chain of a decode step costs per step, not per row, so the whole
        # step's HBM fraction rises with the rows: what the 32-row step lacks to north_star's 50 % is not in the kernels.
        log('rows64: the same generate with 64 distinct utterances')
        utts64 = [synth.synth_utterance(cfg, text // 2, text - text // 2, frames, seed=1234 + u) for u in range(2 * rows)]
        texts64 = [torch.cat([u[0], u[2]]).to(dev) for u in utts64]
        firsts64 = [u[1][:, 0].to(dev) for u in utts64]
        model.generate_batch(texts64, firsts64)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            out64 = model.generate_batch(texts64, firsts64)
        torch.cuda.synchronize()
        dt64 = (time.perf_counter() - t0) / reps
        st64 = model.last_generate_stats
        bytes64 = 4.0 * sum(cfg.num_layers * l_pl + (cfg.num_audio_tokens + 1) * cfg.d_model
                            + 2 * cfg.num_layers * 2 * rows * (st64['s0'] + t) * cfg.d_model
                            + 2 * cfg.num_layers * 2 * rows * cfg.d_model for t in range(1, new))
        gbs64 = bytes64 / (st64['decode_ms'] * 1e-3) / 1e9
        result['rows64'] = {
            'label': 'SECONDARY, not the metric: configs[1] with 64 rows instead of 32 (fp32, same prompt and new tokens)',
            'rows': 2 * rows, 'value': 2 * rows * new / dt64, 'unit': 'tokens/s', 'ms_per_generate': dt64 * 1e3,
            'prefill_ms': st64['prefill_ms'], 'decode_ms_per_step': st64['decode_ms'] / (new - 1),
            'algorithmic_bytes_total': bytes64, 'achieved': gbs64, 'peak': HBM_PEAK_GBS, 'unit_bw': 'GB/s',
            'frac': gbs64 / HBM_PEAK_GBS, 'vs_32_rows': (2 * rows * new / dt64) / (value / world),
            'first_32_rows_equal_to_headline_run': bool((out64[:rows] == out).all())}
        del utts64, texts64, firsts64, out64
        torch.cuda.empty_cache()

    if rank == 0 and world == 1 and not args.no_config5 and not args.small:
        result['config5'] = config5_leg(dev)

    if rank == 0 and world == 1 and not args.no_nar and not args.small:
        log('nar: one stage forward of configs[2]')
        ncfg = ConfigValle(**dict(AR, norm='AdaptiveLayerNorm'))
        nsd = synth.make_state_dict(ncfg, 'ValleNAR', seed=0, rich=True)
        nar = get_model_class('ValleNAR')(ncfg)
        nar.load_state_dict(nsd)
        nar = nar.to(dev).eval()
        nb = synth.synth_nar_batch(ncfg, NAR_B, NAR_TEXT, NAR_FRAMES, seed=1234)
        nb = {k: v.to(dev) for k, v in nb.items()}
        nar.stage_logits(nb, 3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for i in range(reps):
            nar.stage_logits(nb, 1 + i)
        torch.cuda.synchronize()
        t_stage = (time.perf_counter() - t0) / reps
        s = NAR_TEXT + NAR_FRAMES
        flop = (2 * 12 * l_pl * NAR_B * s + 4 * 512 * s * s * 12 * NAR_B
                + 2 * 512 * 1024 * NAR_B * (NAR_FRAMES - 150))
        result['nar'] = {'metric': 'NAR fwd tokens/sec (one stage, configs[2])',
                         'value': NAR_B * s / t_stage, 'unit': 'tokens/s', 'ms_per_stage': t_stage * 1e3,
                         'tflops': flop / t_stage / 1e12, 'mfma_f32_peak_tflops': MFMA_F32_PEAK_TF,
                         'frac_of_mfma_peak': flop / t_stage / 1e12 / MFMA_F32_PEAK_TF}
        if not args.no_perf_mode:
            # the same stage forward in perf mode (SECONDARY, never the metric): bf16 operands on v_mfma_f32_32x32x16_bf16
            nar.stage_logits(nb, 3, perf_mode=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(reps):
                nar.stage_logits(nb, 1 + i, perf_mode=True)
            torch.cuda.synchronize()
            t16 = (time.perf_counter() - t0) / reps
            result['nar']['perf_mode'] = {
                'label': f'SECONDARY, not the metric: the stage forward with {h16_name()} operands / fp32 accumulators (teacher-forced '
                         'logits within atol 5e-2 of the reference: tests/test_bf16_gpu.py)',
                'value': NAR_B * s / t16, 'unit': 'tokens/s', 'ms_per_stage': t16 * 1e3, 'tflops': flop / t16 / 1e12,
                'mfma_bf16_peak_tflops': MFMA_BF16_PEAK_TF, 'frac_of_bf16_peak': flop / t16 / 1e12 / MFMA_BF16_PEAK_TF,
                'vs_f32_stage': t_stage / t16}
        # the same configuration as the reference's generate() runs it (valle_nar.py:107-165): the seven stages one after the
        # other, every stage feeding the next its codes — SURVEY 8(d)'s second form of the metric, B T_target 7 / t_all
        log('nar: the seven stages of configs[2] (generate_batch)')
        g = torch.Generator().manual_seed(4321)
        prefix, target = 150, NAR_FRAMES - 150
        g_texts = [torch.randint(0, ncfg.vocab_size, (NAR_TEXT,), generator=g).to(dev) for _ in range(NAR_B)]
        g_prompts = [torch.randint(0, ncfg.num_audio_tokens, (prefix, ncfg.num_quantizers), generator=g).to(dev) for _ in range(NAR_B)]
        g_firsts = [torch.randint(0, ncfg.num_audio_tokens, (target,), generator=g).to(dev) for _ in range(NAR_B)]
        stages = ncfg.num_quantizers - 1
        all_stages = {}
        for mode in ([False] if args.no_perf_mode else [False, True]):
            nar.generate_batch(g_texts, g_prompts, g_firsts, greedy=True, perf_mode=mode)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            codes = nar.generate_batch(g_texts, g_prompts, g_firsts, greedy=True, perf_mode=mode)
            torch.cuda.synchronize()
            t_all = time.perf_counter() - t0
            assert len(codes) == NAR_B and tuple(codes[0].shape) == (target, ncfg.num_quantizers)
            entry = {'ms_total': t_all * 1e3, 'value': NAR_B * target * stages / t_all, 'unit': 'codec tokens/s',
                     'tflops': stages * flop / t_all / 1e12}
            if mode:
                all_stages['perf_mode'] = dict(entry, label=f'SECONDARY, not the metric: {h16_name()} operands / fp32 accumulators',
                                               vs_f32=all_stages['ms_total'] / entry['ms_total'])
            else:
                all_stages = dict(entry, workload='configs[2] as generate() runs it: 64 rows x (256 text + 150-frame prompt + 618 '
                                                  'target frames), codebooks 2..8 (7 stages, greedy), fp32',
                                  frac_of_mfma_peak=stages * flop / t_all / 1e12 / MFMA_F32_PEAK_TF)
        result['nar']['all_stages'] = all_stages
        del nar, nb

    if not args.no_train:
        # every rank takes part: this is the path's one real exchange (gradient all-reduce over RCCL/xGMI)
        log('train: configs[3] AR + NAR steps')
        try:
            del model
            torch.cuda.empty_cache()
            result['train'] = train_leg(dev, world, rank, args.small)
        except Exception as e:                      # never lose the headline line to the secondary leg
            result['train'] = {'error': f'{type(e).__name__}: {e}'}
            log(f'train leg failed: {e}')

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # same model, same first utterance, all `rows` beams, on the host cores of this box
        result['cpu_baseline'] = cpu_baseline(ar_kw, sd, utts[0], rows, new,
                                              gpu_tokens=out[0, frames + 1:].cpu())

    watchdog.cancel()
    emit()
    if world > 1:
        dist.barrier(group=host_pg)
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
