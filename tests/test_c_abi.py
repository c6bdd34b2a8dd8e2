"""The C ABI used from plain C: tests/abi/c_abi_smoke.c (LayerNorm / linear forms) and tests/abi/c_abi_decode.c (the whole
AR decoder life-cycle — embed, prompt pass, head, greedy step, vh_ar_decoder_create / step / capture / replay / destroy —
against the REAL reference's tokens, plus both attention kernels against double-precision loops), compiled with gcc
against include/valle_hip.h and linked with libvalle_hip.so + the HIP runtime: no Python, no torch in those processes."""
import os
import shutil
import subprocess
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
ROCM = Path(os.environ.get('ROCM_PATH', '/opt/rocm'))


def _build(tmp_path, name='c_abi_smoke'):
    lib = REPO / 'valle2_amd' / 'csrc' / 'libvalle_hip.so'
    if not lib.exists():
        import __graft_entry__
        __graft_entry__.build()
    if shutil.which('gcc') is None or not (ROCM / 'include' / 'hip' / 'hip_runtime_api.h').exists():
        pytest.skip('gcc or the HIP headers are not available')
    exe = tmp_path / name
    cmd = ['gcc', '-std=c11', '-Wall', '-D__HIP_PLATFORM_AMD__', f'-I{ROCM}/include', f'-I{REPO}/include',
           str(REPO / 'tests' / 'abi' / f'{name}.c'), f'-L{lib.parent}', '-lvalle_hip', f'-L{ROCM}/lib',
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


def test_decode_program_compiles_and_the_model_file_is_written(tmp_path):
    """CPU half of the plain-C decoder test: the program compiles and links against the header, and the exporter writes
    the model file (weights from the seeded generator + the reference's golden tokens) with the size its layout implies."""
    _build(tmp_path, 'c_abi_decode')
    from tests.abi.export_tiny_model import export
    path = export(tmp_path / 'tiny.bin')
    d, dff, L, vt, va, n_text, n_prompt, n_new, n_pe = 128, 512, 2, 256, 1024, 128, 256, 64, 321
    floats = vt * d + (va + 2) * d + 2 * n_pe * d + (va + 1) * d + L * (4 * d * d + 2 * d * dff + 6 * d + dff) + n_new
    assert os.path.getsize(path) == 12 * 4 + floats * 4 + (n_text + n_prompt + n_new) * 8


@pytest.mark.gpu
def test_c_program_drives_the_decoder_through_the_abi(tmp_path):
    """vh_embed_sum_pe -> vh_transformer_forward -> vh_linear -> vh_greedy_step ->
    vh_ar_decoder_create / step / capture / replay / destroy with hipMalloc'd buffers and a hipStream_t from a process
    that holds no Python: the 64 greedy tokens of every beam equal the real reference's (ar_generate_tiny.npz)."""
    exe = _build(tmp_path, 'c_abi_decode')
    from tests.abi.export_tiny_model import export
    path = export(tmp_path / 'tiny.bin')
    out = subprocess.run([str(exe), str(path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert 'all checks passed' in out.stdout and 'equal the reference' in out.stdout
    assert 'a second decoder (vh_head_greedy' in out.stdout        # the one-launch head continued on the caller-owned state
    print(out.stdout)
