"""Kernel-level checks through the C ABI against plain fp32 numpy/torch references (GPU box only)."""
import numpy as np
import pytest
import torch

from lpformer_amd import _lib, graph
from lpformer_amd.link_transformer import DenseChain, gemm, layernorm_
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


@pytest.mark.parametrize("m,n,k", [(5000, 128, 128), (4099, 256, 64), (70001, 36, 32), (4096, 132, 60), (9001, 64, 128),
                                   (4097, 8, 4), (6000, 256, 128)])
def test_gemm_rows_kernel(m, n, k):
    """The tall-and-skinny path of lpf_gemm_f32 (gemm_rows_kernel: M >= 4096, K <= 128, N <= 256, both multiples of 4):
    every epilogue combination against fp64 torch, strided inputs and output, a ragged last tile; a row's result does
    not depend on the rows around it (row blocks are bitwise the rows of the whole product)."""
    g = torch.Generator().manual_seed(m + 3 * n + k)
    a_buf = torch.randn(m, k + 8, generator=g).to(DEV)
    a = a_buf[:, 4:4 + k]                                  # row stride k + 8, 16-byte aligned start
    w = torch.randn(n, k, generator=g).to(DEV)
    bias = torch.randn(n, generator=g).to(DEV)
    add = torch.randn(m, n, generator=g).to(DEV)
    ref0 = a.double() @ w.double().T
    tol = 2e-5 * max(1.0, k ** 0.5)
    for use_bias, use_add, relu in ((False, False, False), (True, True, True), (True, False, False), (False, True, True)):
        out_buf = torch.full((m, n + 4), 7.0, device=DEV)
        out = gemm(a, w, bias if use_bias else None, addend=add if use_add else None, relu=relu, out=out_buf[:, :n])
        ref = ref0 + (bias.double() if use_bias else 0) + (add.double() if use_add else 0)
        if relu:
            ref = ref.clamp_min(0)
        assert (out.double() - ref).abs().max().item() <= tol
        assert (out_buf[:, n:] == 7.0).all()               # nothing written past the N columns
    whole = gemm(a, w, bias)
    lo = 4100 if m > 9000 else 0
    if m - lo >= 4096:
        blk = gemm(a[lo:], w, bias)
        assert torch.equal(blk, whole[lo:])
    # the same rows through the tiled kernel (M < 4096 takes it) agree to rounding
    small = gemm(a[:1000], w, bias)
    assert (small - whole[:1000]).abs().max().item() <= tol


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


def _chain_ref(x, w1, b1, add, g, be, relu, w2, b2):
    h = x.double() @ w1.double().T + b1.double()
    if add is not None:
        h = h + add.double()
    if g is not None:
        mu = h.mean(1, keepdim=True)
        var = ((h - mu) ** 2).mean(1, keepdim=True)
        h = (h - mu) / torch.sqrt(var + 1e-5) * g.double() + be.double()
    if relu:
        h = h.clamp_min(0)
    if w2 is not None:
        h = h @ w2.double().T + b2.double()
    return h


@pytest.mark.parametrize("m,k1,n1,n2", [(1, 64, 64, 64), (300, 128, 128, 128), (1000, 132, 132, 128), (77, 68, 68, 64),
                                        (513, 256, 256, 256), (200, 260, 260, 256), (129, 32, 32, 32),
                                        (90, 36, 36, 32), (4096, 128, 128, 128)])
def test_dense_chain_two_layers(m, k1, n1, n2):
    g_ = torch.Generator().manual_seed(m + k1)
    r = lambda *s: torch.randn(*s, generator=g_).to(DEV)  # noqa: E731
    x, w1, b1, lg, lb, w2, b2 = r(m, k1), r(n1, k1) / k1 ** 0.5, r(n1), r(n1), r(n1), r(n2, n1) / n1 ** 0.5, r(n2)
    dc = DenseChain("t")
    out = dc.run(dc.tables(w1, b1, lg, lb, w2, b2), x, relu=True)
    assert out is not None
    ref = _chain_ref(x, w1, b1, None, lg, lb, True, w2, b2)
    assert (out.double() - ref).abs().max().item() <= 3e-5


@pytest.mark.parametrize("m,k1,n1", [(5, 64, 64), (700, 128, 128), (333, 388, 128), (257, 196, 64), (100, 256, 256),
                                     (64, 512, 512), (31, 772, 256)])
@pytest.mark.parametrize("mode", ["plain", "addend_ln", "relu", "dot"])
def test_dense_chain_single_layer_variants(m, k1, n1, mode):
    g_ = torch.Generator().manual_seed(m * 3 + k1)
    r = lambda *s: torch.randn(*s, generator=g_).to(DEV)  # noqa: E731
    x, w1, b1 = r(m, k1), r(n1, k1) / k1 ** 0.5, r(n1)
    dc = DenseChain("t")
    if mode == "plain":
        out = dc.run(dc.tables(w1, b1), x, relu=False)
        ref = _chain_ref(x, w1, b1, None, None, None, False, None, None)
    elif mode == "relu":
        out = dc.run(dc.tables(w1, b1), x, relu=True)
        ref = _chain_ref(x, w1, b1, None, None, None, True, None, None)
    elif mode == "addend_ln":
        add, lg, lb = r(m, n1 + 4)[:, :n1], r(n1), r(n1)
        buf = torch.full((m, n1 + 8), 7.0, device=DEV)
        out = dc.run(dc.tables(w1, b1, lg, lb), x, relu=False, addend=add, out=buf[:, :n1])
        ref = _chain_ref(x, w1, b1, add, lg, lb, False, None, None)
        assert (buf[:, n1:] == 7.0).all()  # strided output view: nothing written past the logical width
    else:
        w2, b2 = r(1, n1) / n1 ** 0.5, r(1)
        t = dc.tables(w1, b1, None, None, w2, b2)
        out = dc.run(t, x, relu=True)
        ref = _chain_ref(x, w1, b1, None, None, None, True, w2, b2).squeeze(1)
        prob = dc.run(t, x, relu=True, want_logit=False)
        assert (prob.double() - torch.sigmoid(ref)).abs().max().item() <= 1e-5
    assert out is not None
    assert (out.double() - ref).abs().max().item() <= 3e-5


@pytest.mark.parametrize("in_mode", [1, 2])
@pytest.mark.parametrize("d", [64, 128, 256])
def test_dense_chain_gathered_inputs(in_mode, d):
    """in_mode 1 (X[a]*X[b], two square layers = elementwise_lin) and 2 (X[a]+X[b], one layer = the q projection)."""
    g_ = torch.Generator().manual_seed(in_mode + d)
    r = lambda *s: torch.randn(*s, generator=g_).to(DEV)  # noqa: E731
    n, m = 500, 1111
    xn = r(n, d)
    batch = torch.randint(0, n, (2, m), generator=g_).to(DEV)
    w1, b1, lg, lb, w2, b2 = r(d, d) / d ** 0.5, r(d), r(d), r(d), r(d, d) / d ** 0.5, r(d)
    dc = DenseChain("t")
    if in_mode == 1:
        out = dc.run(dc.tables(w1, b1, lg, lb, w2, b2), xn, relu=True, batch=batch, in_mode=1)
        ref = _chain_ref(xn[batch[0]] * xn[batch[1]], w1, b1, None, lg, lb, True, w2, b2)
    else:
        out = dc.run(dc.tables(w1, b1), xn, relu=False, batch=batch, in_mode=2)
        ref = _chain_ref(xn[batch[0]] + xn[batch[1]], w1, b1, None, None, None, False, None, None)
    assert out is not None
    assert (out.double() - ref).abs().max().item() <= 3e-5


@pytest.mark.parametrize("d,sd", [(32, 32), (64, 64), (128, 128), (256, 256), (128, 36)])
def test_dense_chain_side_gather(d, sd):
    """lpf_dense_chain_side_f32: the elementwise branch's launch also gathers side[m] = S[a] + S[b] from a second table
    (the attention's query from lin_l(X)) -- bitwise lpf_pair_gather_f32's sum, main output bitwise unchanged; a ragged
    last block, a side width that is not the model width, a strided output."""
    g_ = torch.Generator().manual_seed(d + sd)
    r = lambda *s: torch.randn(*s, generator=g_).to(DEV)  # noqa: E731
    n, m = 700, 1111
    xn, stab = r(n, d), r(n, sd + 4)[:, :sd]
    batch = torch.randint(0, n, (2, m), generator=g_).to(DEV)
    batch[:, :3] = torch.tensor([[0, n - 1, 5], [0, 0, 5]])
    w1, b1, lg, lb = r(d, d) / d ** 0.5, r(d), r(d), r(d)
    dc = DenseChain("t")
    t = dc.tables(w1, b1, lg, lb)
    plain = dc.run(t, xn, relu=True, batch=batch, in_mode=1)
    side = torch.full((m, sd + 4), 7.0, device=DEV)
    both = dc.run(t, xn, relu=True, batch=batch, in_mode=1, side=(stab, side[:, :sd]))
    assert plain is not None and both is not None and torch.equal(plain, both)
    want = torch.empty(m, sd, device=DEV)
    _lib.check(_lib.hip().lpf_pair_gather_f32(m, sd, _lib.ptr(batch), batch.stride(0), n, _lib.ptr(stab), stab.stride(0),
                                              None, 0, _lib.ptr(want), sd, torch.cuda.current_stream().cuda_stream))
    assert torch.equal(side[:, :sd], want) and torch.equal(want, stab[batch[0]] + stab[batch[1]])
    assert (side[:, sd:] == 7.0).all()


def test_dense_chain_unsupported_shape_is_reported():
    x, w1, b1 = torch.randn(8, 96, device=DEV), torch.randn(96, 96, device=DEV), torch.randn(96, device=DEV)
    dc = DenseChain("t")
    assert dc.run(dc.tables(w1, b1), x, relu=False) is None  # nt1 = 6 single layer is not built
    t = dc.tables(w1, b1)
    rc = _lib.hip().lpf_dense_chain_f32(8, 0, _lib.ptr(x), 96, None, 0, 0, 96, _lib.ptr(t["w1p"]), 96, _lib.ptr(t["b1"]),
                                        None, 0, None, None, 0, None, 0, None, _lib.ptr(torch.empty(8, 96, device=DEV)),
                                        96, None, None)
    assert rc == -2


# ------------------------------------------------------------------------------------------ device PPR producer
def _assert_same_csr(a, b):
    np.testing.assert_array_equal(a.rowptr, b.rowptr)
    np.testing.assert_array_equal(a.col, b.col)
    np.testing.assert_array_equal(a.val.view(np.uint32), b.val.view(np.uint32))


@pytest.mark.parametrize("case", ["ppr_push_small", "ppr_push_powerlaw"])
def test_gpu_ppr_push_matches_reference_vectors(case):
    """lpf_ppr_push_f64 + lpf_ppr_pack_csr against the vectors recorded from the reference's calc_ppr."""
    import lpformer_amd
    from tests.golden_util import GOLDEN_DIR
    z = np.load(f"{GOLDEN_DIR}/{case}.npz")
    n = int(z["n"])
    for eps in z["eps_list"]:
        tag = f"{eps:g}".replace("-", "m").replace(".", "p")
        for waves in (4, 64):
            csr = lpformer_amd.ppr.calc_ppr_gpu(z["edge_index"], n, 0.15, float(eps), device=DEV, n_waves=waves)
            np.testing.assert_array_equal(np.repeat(np.arange(n), np.diff(csr.rowptr)), z[f"row_{tag}"])
            np.testing.assert_array_equal(csr.col, z[f"col_{tag}"])
            np.testing.assert_array_equal(csr.val.view(np.uint32), z[f"val_{tag}"].view(np.uint32))


def test_gpu_ppr_push_matches_host_on_weighted_power_law_graph():
    """Bit-identical to the host producer on a 20k-node Chung-Lu graph with hubs, self loops and isolated nodes."""
    import lpformer_amd
    from lpformer_amd import data as D
    n = 20000
    ei, _ = D.chung_lu_graph(n, 90000, gamma=2.3, seed=5, max_weight=1)
    ei = np.concatenate([ei, np.stack([np.arange(7), np.arange(7)])], axis=1)  # a few self loops
    both = np.concatenate([ei, ei[::-1]], axis=1)
    for eps in (1e-3, 1e-4):
        host = lpformer_amd.ppr.calc_ppr(both, n, 0.15, eps)
        gpu = lpformer_amd.ppr.calc_ppr_gpu(both, n, 0.15, eps, device=DEV)
        _assert_same_csr(host, gpu)


def test_gpu_ppr_push_directed_and_retry_path():
    """Directed graph with dangling nodes; a tiny first pool forces the exact-size second run."""
    import lpformer_amd
    rng = np.random.default_rng(3)
    n = 300
    ei = rng.integers(0, n, size=(2, 2500))
    host = lpformer_amd.ppr.calc_ppr(ei, n, 0.15, 1e-4)
    gpu = lpformer_amd.ppr.calc_ppr_gpu(ei, n, 0.15, 1e-4, device=DEV, n_waves=8, pool_capacity=1000)
    _assert_same_csr(host, gpu)
    assert host.rowptr[-1] > 0


# ------------------------------------------------------------------------------------------ device-built PPR indexes
@pytest.mark.parametrize("theta", [0.0, 1e-4, 1e-2])
def test_device_ppr_indexes_match_host_twins(theta):
    """lpf_ppr_filter_count/_fill and lpf_self_ppr against graph.prefilter_nonhop / prefilter_onehop / self_ppr, with
    PPR values jittered around the threshold so that the fp32 +1-1 round trip decides."""
    import lpformer_amd
    from lpformer_amd import data as D
    n = 3000
    ei, _ = D.chung_lu_graph(n, 12000, gamma=2.3, seed=9, max_weight=0)
    ppr = lpformer_amd.calc_ppr(ei, n, 0.15, 2e-4)
    rng = np.random.default_rng(0)
    val = ppr.val.copy()
    pick = rng.random(val.size) < 0.2
    val[pick] = np.float32(max(theta, 1e-3)) * (1 + rng.integers(-6, 7, pick.sum()).astype(np.float32) * np.float32(2 ** -23))
    ppr = graph.CSR(ppr.rowptr, ppr.col, val.astype(np.float32), n)
    adj = graph.mask_csr(ei, n, symmetric=True)
    dppr, dadj = ppr.to_device(DEV), adj.to_device(DEV)
    for mode, host_fn in ((0, graph.prefilter_nonhop), (1, graph.prefilter_onehop)):
        want = host_fn(ppr, theta)
        got = graph.ppr_filter_device(dppr, mode, theta).to_host()
        np.testing.assert_array_equal(got.rowptr, want.rowptr)
        np.testing.assert_array_equal(got.col, want.col)
        np.testing.assert_array_equal(got.val.view(np.uint32), want.val.view(np.uint32))
    sp = graph.self_ppr_device(dadj, dppr).cpu().numpy()
    np.testing.assert_array_equal(sp.view(np.uint32), graph.self_ppr(adj, ppr).view(np.uint32))


def test_gemm_tn_weight_gradient():
    """lpf_gemm_tn_f32: C = A^T B with the reduction over the rows split into chunks (dW of a Linear), against fp64
    torch; ragged N / K / M, a single row, and an empty input."""
    from lpformer_amd import train
    torch.manual_seed(11)
    for m, n, k in ((100_003, 128, 128), (4097, 132, 260), (1, 7, 5), (300, 1, 1433), (0, 16, 8), (50_001, 64, 64), (9000, 32, 64),
                    (7000, 40, 200)):
        a = torch.randn(m, n, device=DEV)
        b = torch.randn(m, k, device=DEV)
        got = train._gemm_tn(a, b)
        want = (a.double().t() @ b.double())
        scale = max(1.0, float(want.abs().max()))
        assert got.shape == (n, k) and float((got.double() - want).abs().max()) <= 2e-5 * scale * max(1.0, m ** 0.5 / 30)
        assert torch.equal(got, train._gemm_tn(a, b))      # deterministic reduction order
        # with the column sums of A (the bias gradient) from the same pass: the product bitwise the same
        got2, cs = train._gemm_tn(a, b, colsum=True)
        assert torch.equal(got2, got) and cs.shape == (n,)
        want_cs = a.double().sum(0)
        assert float((cs.double() - want_cs).abs().max()) <= 2e-5 * max(1.0, float(want_cs.abs().max())) * max(1.0, m ** 0.5 / 30)
        assert torch.equal(cs, train._gemm_tn(a, b, colsum=True)[1])


def test_layernorm_backward_kernel():
    """lpf_layernorm_bwd_f32 (through train.layer_norm) against torch autograd in fp64: dx, dgamma, dbeta; several
    widths, a single row, many rows."""
    from lpformer_amd import train
    torch.manual_seed(5)
    for m, d in ((70_001, 128), (513, 64), (1, 32), (3000, 256), (200, 132)):
        x = torch.randn(m, d, device=DEV, requires_grad=True)
        g = (torch.rand(d, device=DEV) + 0.5).requires_grad_()
        b = torch.randn(d, device=DEV, requires_grad=True)
        dy = torch.randn(m, d, device=DEV)
        y = train.layer_norm(x, g, b)
        y.backward(dy)
        xr, gr, br = (t.detach().double().requires_grad_() for t in (x, g, b))
        yr = torch.nn.functional.layer_norm(xr, (d,), gr, br)
        yr.backward(dy.double())
        assert (y.double() - yr).abs().max().item() <= 1e-5
        assert (x.grad.double() - xr.grad).abs().max().item() <= 1e-4 * max(1.0, float(xr.grad.abs().max()))
        for got, want in ((g.grad, gr.grad), (b.grad, br.grad)):
            assert (got.double() - want).abs().max().item() <= 1e-4 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("dim,mode", [(32, "all"), (64, "all"), (128, "all"), (256, "1-hop"), (128, "cn")])
def test_pair_rows_kernel_matches_the_record_path(dim, mode):
    """The pair-major attention kernel (csrc/pair_rows.hip: finished rows) against the unit-major kernel + record
    merge (pair_flip.hip + pair_merge.hip / the merge stage of tail_chain.hip) it replaces on the hot path: same
    features, same scores -- on a batch with hub pairs of hundreds of entries (dozens of 16-entry units, their pieces
    merged inside the launch), empty pairs, a == b, and so many pairs that a workgroup's range is staged in several
    chunks.  A row depends on its pair and on where the 16-entry grid of the pair-major order cuts it: the rows of a
    PREFIX of the batch are bitwise the same, those of a permuted batch agree to rounding.  bf16 node table variant
    within the mode's tolerance."""
    import lpformer_amd
    from lpformer_amd import data as D
    rng = np.random.default_rng(dim)
    n = 2500
    ei, w = D.chung_lu_graph(n, 16000, gamma=2.1, seed=dim, max_weight=0)
    star = np.stack([np.zeros(700, np.int64), rng.choice(np.arange(2, n), 700, replace=False)])   # two hubs
    star2 = np.stack([np.ones(650, np.int64), rng.choice(np.arange(2, n), 650, replace=False)])
    allp = np.concatenate([ei, star, star[::-1], star2, star2[::-1]], axis=1)
    _, keep = np.unique(allp[0] * n + allp[1], return_index=True)
    ei = allp[:, keep]
    x = rng.standard_normal((n, 40)).astype(np.float32)
    th = {"all": (0.0, 1e-4, 1e-3), "1-hop": (0.0, 1e-4, 1.0), "cn": (0.0, 1.0, 1.0)}[mode]
    data = D.build_data(ei, x, n, ppr=lpformer_amd.calc_ppr(ei, n, 0.15, 1e-4))
    args = D.train_args_for(dict(thresholds=th, dim=dim, gnn_layers=1, residual=False))
    torch.manual_seed(dim)
    model = lpformer_amd.LinkTransformer(args, data, device=DEV).to(DEV).eval()
    score = lpformer_amd.mlp_score(2 * dim, 2 * dim, 1, 2).to(DEV).eval()
    with torch.no_grad():
        for p in list(model.parameters()) + list(score.parameters()):
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    bs = 140_000 if dim == 128 and mode == "all" else 6_000      # 140 k: > 512 pairs per workgroup (chunked ranges)
    batch = D.sample_pairs(ei, n, bs, seed=dim + 1, frac_edges=0.3)
    batch[:, :8] = np.array([[0, 1, 0, 0, 1, 5, 7, 7], [1, 0, 0, 5, 9, 5, 9, 9]])          # hub pairs, a == b, duplicates
    tb = torch.from_numpy(batch).to(DEV)
    h = model.propagate()
    model.attention_impl = "flip"
    outs = {}
    for rows in (True, False):
        model.attention_rows = rows
        feats, _ = model.calc_pairwise(tb, h)
        lg = model.score_pairs(tb, h, score, logits=True)
        assert model.check_selection()
        outs[rows] = (feats.clone(), lg.clone())
    sel = model.compute_node_mask(tb[:, :64])
    assert sel[0][0].shape[1] > 0 if mode != "cn" else True
    cnt = torch.zeros(bs, dtype=torch.long, device=DEV)
    for t in model.compute_node_mask(tb):
        if t is not None:
            cnt += torch.bincount(t[0][0], minlength=bs)
    assert int(cnt.max()) > 96 and int((cnt == 0).sum()) > 0 and int(((cnt > 0) & (cnt < 16)).sum()) > 0
    scale = max(1.0, float(outs[False][0].abs().max()))
    assert (outs[True][0] - outs[False][0]).abs().max().item() <= 2e-5 * scale
    if dim <= 128:
        assert (outs[True][1] - outs[False][1]).abs().max().item() <= 2e-5 * max(1.0, float(outs[False][1].abs().max()))
    model.attention_rows = True
    k = 1500
    sub, _ = model.calc_pairwise(tb[:, :k].contiguous(), h)
    assert torch.equal(sub, outs[True][0][:k])                     # same pairs in front of it: same cuts, same bits
    perm = torch.randperm(bs, device=DEV)
    pf, _ = model.calc_pairwise(tb[:, perm].contiguous(), h)
    assert (pf - outs[True][0][perm]).abs().max().item() <= 2e-6 * scale
    # the tail with the pairs that select nothing handled apart (lpf_pair_attention_rows_perm_* +
    # lpf_tail_chain_rows_perm_*): the order it leaves, and the same scores as the plain rows tail
    lscale = max(1.0, float(outs[True][1].abs().max()))
    assert model.tail_skip_empty
    ws = {k[0]: v for k, v in model._ws.items()}
    n_full = int(ws["att_nfull"][0])
    order = ws["att_perm"][:bs].long()
    assert n_full == int((cnt > 0).sum())
    assert torch.equal(order[:n_full], torch.nonzero(cnt > 0).flatten())
    assert torch.equal(order[n_full:].flip(0), torch.nonzero(cnt == 0).flatten())
    model.tail_skip_empty = False
    plain = model.score_pairs(tb, h, score, logits=True)
    model.tail_skip_empty = True
    assert (plain - outs[True][1]).abs().max().item() <= 2e-6 * lscale
    assert torch.equal(model.score_pairs(tb, h, score, logits=True), outs[True][1])           # replay: same bits
    for sub in (torch.nonzero(cnt == 0).flatten()[:700], torch.nonzero(cnt > 0).flatten()[:700]):   # all / none empty
        tsub = tb[:, sub].contiguous()
        for _attempt in range(3):     # (far more entries per pair than the big batch: the workspace may have to grow once)
            got = model.score_pairs(tsub, h, score, logits=True)
            if model.check_selection():
                break
        else:
            raise AssertionError("selection workspace did not settle")
        assert (got - outs[True][1][sub]).abs().max().item() <= 2e-6 * lscale
    if dim <= 128:
        model.precision = model.tail_precision = "bf16"
        lg16 = model.score_pairs(tb, h, score, logits=True)
        assert model.check_selection()
        assert (lg16 - outs[True][1]).abs().max().item() <= 5e-3 * max(1.0, float(outs[True][1].abs().max()))
