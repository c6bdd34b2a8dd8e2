"""Static checks on the compiled gfx950 code of the LDS-DMA tile GEMMs (forward NT kernel and the TN weight-gradient kernel) and of the five-product attention backward (no GPU needed; hipcc cross-compiles).

The kernel issues its DMA from inline assembly that writes M0, a reserved register the compiler does not track
through a clobber list, and relies on its main loop holding no vector ALU instruction besides the MFMAs.  Both are
properties of the generated code, so they are asserted on the generated code:
  * every reference to m0 in gemm_tile_dma_kernel<*> is one of our `s_mov_b32 m0, ...`;
  * the K loop (the inner loop with the MFMAs) contains no VALU instruction other than v_mfma_*, no scratch
    access, and the expected DMA / LDS-read counts per two K steps.
usage: python tools/check_isa.py   (exit code 0 = ok)
"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
HIPCC = '/opt/rocm/bin/hipcc'
ATTN_WAIT_KERNELS = ['_Z21attn_bwd_fused_kernelILi8EEv7BwdArgs', '_Z21attn_bwd_fused_kernelILi4EEv7BwdArgs', '_Z19attn_bwd_dkv_kernel7BwdArgs',
                     '_Z16attn_rows_kernel8RowsArgs']
BF16_WAIT_KERNELS = ['_Z13attn16_kernel10Attn16Args']


def compile_asm():
    with tempfile.TemporaryDirectory() as td:
        out = Path(td) / 'gemm.s'
        subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only',
                        f'-I{REPO / "include"}', str(REPO / 'valle2_amd/csrc/gemm.hip'), '-o', str(out)],
                       check=True, capture_output=True)
        return out.read_text()


def check(asm):
    problems = []
    kernels = re.findall(r'^(_Z20gemm_tile_dma_kernelILi\dE\w*):[^\n]*\n(.*?)s_endpgm', asm, re.S | re.M)
    if len(kernels) < 3:
        problems.append(f'expected >= 3 instantiations of gemm_tile_dma_kernel, found {len(kernels)}')
    # the weight-gradient kernel (round 2): same rules, 64 ds_read2_b32 per two K steps instead of 32 ds_read_b128
    tn = re.findall(r'^(_Z19gemm_tile_tn_kernel\w*):[^\n]*\n(.*?)s_endpgm', asm, re.S | re.M)
    if len(tn) != 1:
        problems.append(f'expected gemm_tile_tn_kernel once, found {len(tn)}')
    expected = {name: (128, 16, 32) for name, _ in kernels}
    expected.update({name: (128, 16, 64) for name, _ in tn})
    for name, body in kernels + tn:
        lines = [l.strip() for l in body.splitlines()]
        code = [l for l in lines if l and not l.startswith(';')]
        for l in code:
            if re.search(r'\bm0\b', l) and not l.startswith('s_mov_b32 m0,'):
                problems.append(f'{name}: m0 used outside the DMA sequence: {l}')
        heads = [i for i, l in enumerate(lines) if 'Loop Header' in l]
        loop = None
        for hi in heads:                       # the loop that holds the MFMAs
            label = lines[hi].split(':')[0]
            ends = [i for i in range(hi, len(lines)) if lines[i].startswith('s_cbranch') and lines[i].endswith(label)]
            if ends and any(x.startswith('v_mfma') for x in lines[hi:ends[-1]]):
                loop = [x for x in lines[hi:ends[-1] + 1] if x and not x.startswith(';') and not x.endswith(':')]
        if loop is None:
            problems.append(f'{name}: K loop not found')
            continue
        ops = [x.split()[0] for x in loop]
        valu = [o for o in ops if o.startswith('v_') and not o.startswith('v_mfma')]
        if valu:
            problems.append(f'{name}: vector ALU instructions in the K loop: {sorted(set(valu))}')
        if any(o.startswith('scratch_') for o in ops):
            problems.append(f'{name}: scratch access in the K loop')
        n_mfma = sum(o.startswith('v_mfma') for o in ops)
        n_dma = sum(o.startswith('global_load_lds') for o in ops)
        n_lds = sum(o.startswith('ds_read') for o in ops)
        if (n_mfma, n_dma, n_lds) != expected[name]:
            problems.append(f'{name}: K loop (two steps) has {n_mfma} MFMA / {n_dma} DMA / {n_lds} ds_read, '
                            f'expected {expected[name]}')
    return problems


def compile_attention_asm():
    with tempfile.TemporaryDirectory() as td:
        out = Path(td) / 'attention.s'
        subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only',
                        '-mllvm', '-amdgpu-kernarg-preload-count=16',
                        f'-I{REPO / "include"}', str(REPO / 'valle2_amd/csrc/attention.hip'), '-o', str(out)],
                       check=True, capture_output=True)
        return out.read_text()


def check_attention(asm):
    """The five-product attention backward (csrc/attention.hip, attn_bwd_fused_kernel<8>) sits at the register limit of
    two waves per SIMD and depends on a hand-pinned schedule of its dQ loop:
      * no scratch access anywhere in the eight-wave kernel (a spill there is a memory round trip inside every tile);
      * the dQ loop (the inner loop with the 16x16x4 MFMAs) requests the NEXT key group's two operands before the
        current group's MFMAs and never drains the LDS queue (`s_waitcnt lgkmcnt(0)`) inside the loop."""
    problems = []
    m = re.search(r'^_Z21attn_bwd_fused_kernelILi8EEv7BwdArgs:[^\n]*\n(.*?)^\.Lfunc_end', asm, re.S | re.M)   # (the kernel has an early s_endpgm)
    if not m:
        return ['attn_bwd_fused_kernel<8> not found']
    lines = [l.strip() for l in m.group(1).splitlines()]
    if any(l.startswith('scratch_') for l in lines):
        problems.append('attn_bwd_fused_kernel<8>: scratch access (register spill)')
    heads = [i for i, l in enumerate(lines) if 'Loop Header' in l]
    loop = None
    for hi in heads:
        # (a nested loop's header comment sits on the line after its label)
        label = (lines[hi] if not lines[hi].startswith(';') else lines[hi - 1]).split(':')[0]
        ends = [i for i in range(hi, len(lines)) if lines[i].startswith('s_cbranch') and lines[i].endswith(label)]
        if ends and any(x.startswith('v_mfma_f32_16x16x4') for x in lines[hi:ends[-1]]) and \
                not any(x.startswith('v_mfma_f32_32x32x2') for x in lines[hi:ends[-1]]):
            loop = [x for x in lines[hi:ends[-1] + 1] if x and not x.startswith(';') and not x.endswith(':')]
    if loop is None:
        return problems + ['attn_bwd_fused_kernel<8>: dQ loop not found']
    ops = [x.split()[0] for x in loop]
    first_read = next((i for i, o in enumerate(ops) if o == 'ds_read_b128'), None)
    first_mfma = next((i for i, o in enumerate(ops) if o.startswith('v_mfma')), None)
    if first_read is None or first_mfma is None or first_read > first_mfma:
        problems.append('attn_bwd_fused_kernel<8>: the dQ loop does not request its next operands before its MFMAs')
    for x in loop:
        if x.startswith('s_waitcnt') and 'lgkmcnt(0)' in x:
            problems.append(f'attn_bwd_fused_kernel<8>: the dQ loop drains the LDS queue: {x}')
    return problems


def compile_bf16_asm():
    with tempfile.TemporaryDirectory() as td:
        out = Path(td) / 'bf16.s'
        subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only',
                        '-mllvm', '-amdgpu-kernarg-preload-count=16',
                        f'-I{REPO / "include"}', str(REPO / 'valle2_amd/csrc/bf16.hip'), '-o', str(out)],
                       check=True, capture_output=True)
        return out.read_text()


def check_loop_waits(asm, kernels):
    """Round 6: fragments loaded BEFORE a tile loop and first used INSIDE it made the compiler's wait insertion put
    `s_waitcnt vmcnt(n)` in front of the first MFMAs of every iteration whenever a path around the pre-loop wait existed — a
    wait for the iteration's own prefetches (attn16_kernel, attn_bwd_fused_kernel<8>, attn_bwd_dkv_kernel).  The
    kernels carry an unconditional vmcnt(0) before their loops now; this asserts that no vmcnt wait stands directly in front of
    an MFMA inside any loop of the named kernels."""
    problems = []
    for mangled in kernels:
        m = re.search(r'^%s:[^\n]*\n(.*?)^\.Lfunc_end' % re.escape(mangled), asm, re.S | re.M)
        if not m:
            problems.append(f'{mangled} not found')
            continue
        lines = m.group(1).splitlines()
        inloop = False
        for i, l in enumerate(lines):
            if 'Loop Header' in l or 'in Loop:' in l:
                inloop = True
            elif re.match(r'^\.LBB', l):
                inloop = False
            t = l.strip()
            if inloop and t.startswith('s_waitcnt') and 'vmcnt' in t:
                nxt = next((x.strip() for x in lines[i + 1:i + 4] if x.strip() and not x.strip().startswith(';')), '')
                if nxt.startswith('v_mfma'):
                    problems.append(f'{mangled}: "{t}" in front of an MFMA inside a loop')
    return problems


def check_m0(asm, kernels):
    """Kernels that issue LDS-DMA from inline assembly write M0 behind the compiler's back (no clobber list tracks it): every
    reference to m0 in them must be one of our own `s_mov_b32 m0, ...` (the same rule check() applies to the tile GEMMs)."""
    problems = []
    for mangled in kernels:
        m = re.search(r'^%s:[^\n]*\n(.*?)^\.Lfunc_end' % re.escape(mangled), asm, re.S | re.M)
        if not m:
            problems.append(f'{mangled} not found')
            continue
        n_dma = 0
        for l in m.group(1).splitlines():
            t = l.strip()
            if not t or t.startswith(';'):
                continue
            n_dma += t.startswith('global_load_lds_dwordx4')
            if re.search(r'\bm0\b', t) and not t.startswith('s_mov_b32 m0,'):
                problems.append(f'{mangled}: m0 used outside the DMA sequence: {t}')
        if n_dma == 0:
            problems.append(f'{mangled}: no global_load_lds_dwordx4 found (the staging is expected to be LDS-DMA)')
    return problems


if __name__ == '__main__':
    probs = check(compile_asm()) + check_attention(compile_attention_asm())
    probs += check_loop_waits(compile_attention_asm(), ATTN_WAIT_KERNELS) + check_loop_waits(compile_bf16_asm(), BF16_WAIT_KERNELS)
    probs += check_m0(compile_bf16_asm(), BF16_WAIT_KERNELS)
    for p in probs:
        print('ISA check:', p)
    print('ISA check:', 'FAILED' if probs else 'ok')
    sys.exit(1 if probs else 0)
