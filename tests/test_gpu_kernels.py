"""Kernel-level checks through the C ABI against plain fp32 numpy/torch references (GPU box only)."""
import numpy as np
import pytest
import torch

from lpformer_amd import _lib, graph
from lpformer_amd.link_transformer import gemm, layernorm_
from oracle import lpformer_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("m,n,k", [(1, 1, 4), (96, 64, 64), (129, 132, 132), (300, 128, 388), (257, 260, 1436),
                                   (1000, 512, 256), (77, 5, 58), (4096, 128, 128)])
@pytest.mark.parametrize("relu", [False, True])
def test_gemm_f32(m, n, k, relu):
    g = torch.Generator().manual_seed(m * 7 + n)
    kp = (k + 3) // 4 * 4
    a = torch.randn(m, kp, generator=g)[:, :k].to(DEV)
    a_full = torch.zeros(m, kp, device=DEV)
    a_full[:, :k] = a
    w_full = torch.zeros(n, kp, device=DEV)
    w_full[:, :k] = torch.randn(n, k, generator=g).to(DEV)
    bias = torch.randn(n, generator=g).to(DEV)
    add = torch.randn(m, n, generator=g).to(DEV)
    out = gemm(a_full[:, :k], w_full[:, :k], bias, addend=add, relu=relu)
    ref = a_full[:, :k].double() @ w_full[:, :k].double().T + bias.double() + add.double()
    if relu:
        ref = ref.clamp_min(0)
    assert (out.double() - ref).abs().max().item() <= 2e-5 * max(1.0, k ** 0.5)


def test_gemm_asymmetric_layout():
    """A = I against an asymmetric W catches a transposed accumulator map."""
    n = 64
    a = torch.eye(n, device=DEV)
    w = torch.arange(n * n, dtype=torch.float32, device=DEV).reshape(n, n) / 100.0
    out = gemm(a, w)
    torch.testing.assert_close(out, w.T.contiguous())


@pytest.mark.parametrize("d", [32, 64, 96, 128, 256])
def test_spmm_fused_epilogue(d):
    rng = np.random.default_rng(d)
    n = 500
    ei = rng.integers(0, n, size=(2, 4000))
    ei[:, :600] = np.stack([np.zeros(600, np.int64), rng.integers(0, n, 600)])  # one long row
    w = rng.random(ei.shape[1]).astype(np.float32) + 0.5
    struct = graph.gcn_structure_csr(ei, w, n).to_device(DEV)
    a_hat = graph.gcn_norm_device(struct)
    ref_rp, ref_col, ref_val = O.gcn_norm(ei, w, n)
    np.testing.assert_array_equal(a_hat.col.cpu().numpy(), ref_col)
    np.testing.assert_allclose(a_hat.val.cpu().numpy(), ref_val, rtol=2e-6, atol=1e-7)
    h = rng.standard_normal((n, d)).astype(np.float32)
    res = rng.standard_normal((n, d)).astype(np.float32)
    bias, g1, b1, g2, b2 = (rng.standard_normal(d).astype(np.float32) for _ in range(5))
    t = lambda v: torch.from_numpy(v).to(DEV)  # noqa: E731
    deg = a_hat.rowptr[1:] - a_hat.rowptr[:-1]
    hubs = torch.nonzero(deg > 128).flatten().to(torch.int32)
    assert hubs.numel() >= 1
    for use_ln, relu, use_res, use_ln2 in [(True, True, True, True), (False, False, False, False),
                                           (True, False, False, True), (False, True, True, False)]:
      for long_rows in (None, hubs):  # without and with the hub-row kernel
          out = torch.empty(n, d, device=DEV)
          th, tr = t(h), t(res)
          args = [t(bias), t(g1) if use_ln else None, t(b1) if use_ln else None, tr if use_res else None,
                  t(g2) if use_ln2 else None, t(b2) if use_ln2 else None]
          _lib.check(_lib.hip().lpf_spmm_csr_f32(
              n, d, a_hat.rowptr.data_ptr(), a_hat.col.data_ptr(), a_hat.val.data_ptr(), th.data_ptr(), d,
              out.data_ptr(), d, args[0].data_ptr(), _lib.ptr(args[1]), _lib.ptr(args[2]), _lib.ptr(args[3]), d,
              _lib.ptr(args[4]), _lib.ptr(args[5]), 1 if relu else 0, _lib.ptr(long_rows),
              0 if long_rows is None else long_rows.numel(), torch.cuda.current_stream().cuda_stream))
          y = O.spmm(ref_rp, ref_col, ref_val, h) + bias
          if use_ln:
              y = O.layer_norm(y, g1, b1)
          if relu:
              y = np.maximum(y, 0)
          if use_res:
              y = res + y
          if use_ln2:
              y = O.layer_norm(y, g2, b2)
          assert np.abs(out.cpu().numpy() - y).max() <= 5e-5


@pytest.mark.parametrize("d", [3, 64, 132, 260, 1000])
def test_layernorm(d):
    g = torch.Generator().manual_seed(d)
    x = torch.randn(333, d, generator=g).to(DEV)
    w, b = torch.randn(d, generator=g).to(DEV), torch.randn(d, generator=g).to(DEV)
    ref = torch.relu(torch.nn.functional.layer_norm(x, (d,), w, b))
    out = layernorm_(x.clone(), w, b, relu=True)
    assert (out - ref).abs().max().item() <= 2e-5
