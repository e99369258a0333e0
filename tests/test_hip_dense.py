"""GPU parity of the dense-matrix covariance path (fh_dense_matvec / fh_dense_rank2, free_hunch_amd.dense) against
the golden vectors captured from the reference's update_covariance / update_bfgs and against the CPU oracle."""
import numpy as np
import pytest
import torch

import inputs
from test_oracle_golden import DENSE_CASES, check_dense_chain, oracle_dense_updates

pytestmark = pytest.mark.gpu
F64 = torch.float64


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.mark.parametrize("bs,d", [(1, 1), (2, 5), (3, 15), (2, 64), (2, 255), (1, 1030), (2, 2048)])
def test_matvec_and_rank2_kernels(dev, bs, d):
    """Raw kernels against float64 torch on the CPU: every size class (odd d = scalar path, ragged last rows/strips)."""
    from free_hunch_amd import dense
    A = inputs.randn((bs, d, d), 100 + d)
    x, u1, v1, u2, v2 = (inputs.randn((bs, d), 200 + d + k) for k in range(5))
    a1, a2 = inputs.randn((bs,), 300 + d), inputs.randn((bs,), 301 + d)
    Ad = A.to(dev)
    tol = 1e-13 * d * max(1.0, float(A.abs().max()) * float(x.abs().max()))
    ref = (A @ x[..., None])[..., 0]
    assert float((dense.matvec(Ad, x.to(dev)).cpu() - ref).abs().max()) < tol
    reft = (A.transpose(1, 2) @ x[..., None])[..., 0]
    assert float((dense.matvec(Ad, x.to(dev), trans=True).cpu() - reft).abs().max()) < tol
    y0 = inputs.randn((bs, d), 400 + d)
    y = y0.to(dev).clone()
    dense.matvec(Ad, x.to(dev), alpha=0.5, beta=-2.0, out=y)
    assert float((y.cpu() - (0.5 * ref - 2.0 * y0)).abs().max()) < 4 * tol
    eye = torch.eye(d, dtype=F64)
    refr = 0.7 * (A + a1[:, None, None] * u1[:, :, None] * v1[:, None, :]
                  + a2[:, None, None] * u2[:, :, None] * v2[:, None, :]) + 1.25 * eye
    got = dense.rank2(Ad, u1.to(dev), v1.to(dev), a1.to(dev), u2.to(dev), v2.to(dev), a2.to(dev), scale=0.7, shift=1.25)
    assert float((got.cpu() - refr).abs().max()) < 1e-13 * max(1.0, float(refr.abs().max()))
    one = dense.rank2(Ad, u1.to(dev), v1.to(dev), a1.to(dev))  # single term, no scaling
    assert float((one.cpu() - (A + a1[:, None, None] * u1[:, :, None] * v1[:, None, :])).abs().max()) < 1e-13 * 30
    inplace = Ad.clone()
    dense.rank2(inplace, shift=-3.0, out=inplace)  # aliasing allowed
    assert float((inplace.cpu() - (A - 3.0 * eye)).abs().max()) == 0.0


def test_bad_arguments_raise(dev):
    from free_hunch_amd import dense
    A = torch.zeros(2, 4, 4, dtype=F64, device=dev)
    with pytest.raises(ValueError):
        dense.matvec(A.float(), torch.zeros(2, 4, device=dev))
    with pytest.raises(ValueError):
        dense.matvec(A, torch.zeros(2, 5, dtype=F64, device=dev))
    with pytest.raises(ValueError):
        dense.matvec(torch.zeros(2, 4, 4, dtype=F64), torch.zeros(2, 4, dtype=F64))  # CPU tensors: no fallback


@pytest.mark.parametrize("tag,seed,bs,d", DENSE_CASES)
def test_dense_updates_vs_reference_golden(dev, gold, tag, seed, bs, d):
    """free_hunch_amd.dense.update_covariance / update_bfgs through the scripted chain vs the reference's outputs."""
    from free_hunch_amd import dense
    case = inputs.dense_case(seed, bs, d)
    chain = inputs.dense_chain(case, dense.update_covariance, dense.update_bfgs, to=lambda t: t.to(dev))
    check_dense_chain(gold, chain, tag, d, inputs.randn((bs, d), 3000 + seed), 1e-9)


def test_dense_updates_vs_oracle_d4096(dev):
    """Same chain at d = 4096 (bs = 1) against the CPU oracle; relative 1e-8 of each matrix' scale."""
    from free_hunch_amd import dense
    torch.set_num_threads(max(1, torch.get_num_threads()))
    case = inputs.dense_case(77, 1, 4096, n_steps=2)
    tu, su = oracle_dense_updates()
    hip = inputs.dense_chain(case, dense.update_covariance, dense.update_bfgs, to=lambda t: t.to(dev))
    for (what, i, *cpu_state), (_, _, *gpu_state) in zip(inputs.dense_chain(case, tu, su), hip):
        for nm, a, b in zip(("C", "Ci", "H", "Hi", "score", "mean"), cpu_state, gpu_state):
            if a is None:
                continue
            err = float((a - b.cpu()).abs().max())
            assert err < 1e-8 * max(1.0, float(a.abs().max())), (what, i, nm, err)
