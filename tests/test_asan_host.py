"""SURVEY.md section 5 / VERDICT r4 item 7: the host side of libvalle_hip.so under AddressSanitizer + UBSan, CPU only.
`make -C valle2_amd/csrc asan` compiles every csrc/*.hip with `-fsanitize=address,undefined -fno-gpu-sanitize` (host code
sanitized, device code untouched: no GPU sanitizer, no XNACK) and links tests/abi/asan_host.cpp against it; the driver calls
every entry point that needs no device — argument rejection with null / misaligned / out-of-range arguments, every workspace
planner over a grid of shapes, the decoder's create / destroy / state errors, the thread-local error string from 8 threads —
handing over pointers into a FREED block, so a host-side dereference of a device pointer is a use-after-free report."""
import os
import shutil
import subprocess
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
CSRC = REPO / 'valle2_amd' / 'csrc'


def test_host_shim_is_clean_under_asan_and_ubsan():
    if shutil.which('make') is None or not Path('/opt/rocm/bin/hipcc').exists():
        pytest.skip('make / hipcc not available')
    build = subprocess.run(['make', '-C', str(CSRC), '-j4', 'asan'], capture_output=True, text=True, timeout=1500)
    assert build.returncode == 0, build.stdout[-3000:] + build.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0:halt_on_error=1', UBSAN_OPTIONS='print_stacktrace=1')
    env.pop('LD_PRELOAD', None)
    run = subprocess.run([str(CSRC / 'asan' / 'asan_host')], capture_output=True, text=True, timeout=600, env=env)
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-6000:]
    assert 'checks passed' in run.stdout and 'ERROR: AddressSanitizer' not in run.stderr and 'runtime error' not in run.stderr
    n = int(run.stdout.split('asan_host:')[1].split()[0])
    assert n > 1000, run.stdout
