"""Times lpf_dense_chain_f32 on the collab-like pair-stage shapes (M = 32768, D = 128) against the unfused kernels."""
import sys
import torch

sys.path.insert(0, ".")
from lpformer_amd.link_transformer import DenseChain, gemm, layernorm_  # noqa: E402

DEV = "cuda:0"
import os
M, D, N = int(os.environ.get("DC_M", 32768)), 128, 235868


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    g = torch.Generator().manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g).to(DEV)  # noqa: E731
    xn = r(N, D)
    batch = torch.randint(0, N, (2, M), generator=g).to(DEV)
    rows = []

    def case(name, k1, n1, n2, ln, relu, in_mode=0, addend=False):
        w1, b1 = r(n1, k1) / k1 ** 0.5, r(n1)
        lg, lb = (r(n1), r(n1)) if ln else (None, None)
        w2, b2 = (r(n2, n1) / n1 ** 0.5, r(n2)) if n2 else (None, None)
        dc = DenseChain(name)
        t = dc.tables(w1, b1, lg, lb, w2, b2)
        x = xn if in_mode else r(M, k1)
        add = r(M, n1) if addend else None
        fused = lambda: dc.run(t, x, relu=relu, batch=batch if in_mode else None, in_mode=in_mode, addend=add)  # noqa
        assert fused() is not None
        flops = 2.0 * M * (k1 * n1 + n1 * n2)

        def unfused():
            xin = x
            if in_mode:
                xin = xn[batch[0]] * xn[batch[1]] if in_mode == 1 else xn[batch[0]] + xn[batch[1]]
            h = gemm(xin, w1, b1, addend=add, relu=relu and not ln)
            if ln:
                layernorm_(h, lg, lb, relu=relu)
            if n2:
                h = gemm(h, w2, b2)
            return h
        tf, tu = timeit(fused), timeit(unfused)
        rows.append((name, tf, tu, flops / tf / 1e6))
        print(f"{name:14s} fused {tf:7.1f} us  unfused {tu:7.1f} us  fused {flops / tf / 1e6:6.1f} TF/s", flush=True)

    case("elementwise", 128, 128, 128, True, True, in_mode=1)
    case("q_proj", 128, 128, 0, False, False, in_mode=2)
    case("attn_out", 388, 128, 0, True, False, addend=True)
    case("pairwise_lin", 132, 132, 128, True, True)
    case("score_head", 256, 256, 1, False, True)
    print("total fused %.1f us, unfused %.1f us" % (sum(x[1] for x in rows), sum(x[2] for x in rows)))


if __name__ == "__main__":
    main()
