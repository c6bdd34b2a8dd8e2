"""The LDS-DMA tile GEMM relies on two properties of its *compiled* code (M0 written only by its own DMA sequence;
no vector ALU instruction but the MFMAs in the K loop).  hipcc cross-compiles without a GPU, so they are checked here."""
import os
import sys
from pathlib import Path

import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / 'tools'))


@pytest.mark.skipif(not os.path.exists('/opt/rocm/bin/hipcc'), reason='hipcc not installed')
def test_tile_dma_kernel_isa_properties():
    import check_isa
    problems = check_isa.check(check_isa.compile_asm())
    assert not problems, '\n'.join(problems)


@pytest.mark.skipif(not os.path.exists('/opt/rocm/bin/hipcc'), reason='hipcc not installed')
def test_attention_backward_isa_properties():
    import check_isa
    problems = check_isa.check_attention(check_isa.compile_attention_asm())
    assert not problems, '\n'.join(problems)


@pytest.mark.skipif(not os.path.exists('/opt/rocm/bin/hipcc'), reason='hipcc not installed')
def test_attention_tile_loops_do_not_wait_for_their_own_prefetch():
    """Round 6: no `s_waitcnt vmcnt` directly in front of an MFMA inside a loop of the attention kernels."""
    import check_isa
    problems = check_isa.check_loop_waits(check_isa.compile_attention_asm(), check_isa.ATTN_WAIT_KERNELS)
    bf16 = check_isa.compile_bf16_asm()
    problems += check_isa.check_loop_waits(bf16, check_isa.BF16_WAIT_KERNELS)
    problems += check_isa.check_m0(bf16, check_isa.BF16_WAIT_KERNELS)         # the LDS-DMA staging writes M0 from inline asm
    assert not problems, '\n'.join(problems)
