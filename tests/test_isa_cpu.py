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
