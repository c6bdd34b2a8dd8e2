"""The C ABI used from plain C (tests/abi/c_abi_smoke.c): compiled with gcc against include/valle_hip.h
and linked with libvalle_hip.so + the HIP runtime — no Python, no torch in that process."""
import os
import shutil
import subprocess
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
ROCM = Path(os.environ.get('ROCM_PATH', '/opt/rocm'))


def _build(tmp_path):
    lib = REPO / 'valle2_amd' / 'csrc' / 'libvalle_hip.so'
    if not lib.exists():
        import __graft_entry__
        __graft_entry__.build()
    if shutil.which('gcc') is None or not (ROCM / 'include' / 'hip' / 'hip_runtime_api.h').exists():
        pytest.skip('gcc or the HIP headers are not available')
    exe = tmp_path / 'c_abi_smoke'
    cmd = ['gcc', '-std=c11', '-Wall', '-D__HIP_PLATFORM_AMD__', f'-I{ROCM}/include', f'-I{REPO}/include',
           str(REPO / 'tests' / 'abi' / 'c_abi_smoke.c'), f'-L{lib.parent}', '-lvalle_hip', f'-L{ROCM}/lib',
           '-lamdhip64', '-lm', f'-Wl,-rpath,{lib.parent}', f'-Wl,-rpath,{ROCM}/lib', '-o', str(exe)]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_c_program_links_and_host_only_checks_pass(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([str(exe), '--no-gpu'], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert 'host-only checks passed' in out.stdout


@pytest.mark.gpu
def test_c_program_runs_the_kernels(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert 'all checks passed' in out.stdout
