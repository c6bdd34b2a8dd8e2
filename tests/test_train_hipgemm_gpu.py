"""The same gradient-parity checks as tests/test_train_gpu.py with the backward matrix products on
the hand-written vh_gemm_batched (autograd.BACKWARD_GEMM = 'hip') instead of the library GEMMs."""
import pytest
import torch

from tests import test_train_gpu as T

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _hip_backward_gemm():
    from valle2_amd import autograd as A
    old = A.BACKWARD_GEMM, A.ATTENTION_BACKWARD
    A.BACKWARD_GEMM, A.ATTENTION_BACKWARD = 'hip', 'materialized'   # the path that runs on vh_gemm_batched
    yield
    A.BACKWARD_GEMM, A.ATTENTION_BACKWARD = old


def test_ar_gradients_hip_gemm():
    T.test_ar_training_step_gradients_vs_oracle_and_reference()


@pytest.mark.parametrize('stage', [2, 7])
def test_nar_gradients_hip_gemm(stage):
    T.test_nar_training_step_gradients_vs_oracle(stage)


@pytest.mark.parametrize('mode', ['prefix', 'full'])
def test_attention_backward_hip_gemm(mode):
    T.test_qkv_attention_backward(mode)


def test_both_engines_agree():
    """One AR step with each engine: losses equal, every gradient within 1e-4 relative."""
    from tests.golden import cases as C
    from tests.test_models_gpu import build
    from valle2_amd import autograd as A
    kw, sd, batch = C.ar_train_inputs()
    grads = {}
    for engine in ('hip', 'library'):
        A.BACKWARD_GEMM = engine
        model = build('ValleAR', kw, sd)
        loss = model.training_step({k: v.clone() for k, v in batch.items()})
        loss.backward()
        grads[engine] = {n: p.grad.clone() for n, p in model.named_parameters()}, float(loss.detach())
    # the forward is the same kernels; the mean loss is an fp32 atomic sum (order varies per run)
    assert abs(grads['hip'][1] - grads['library'][1]) < 1e-5 * abs(grads['library'][1])
    for n, g in grads['hip'][0].items():
        ref = grads['library'][0][n]
        assert float((g - ref).norm() / ref.norm().clamp_min(1e-12)) < 1e-4, n
