"""CPU-baseline worker of bench.py: one host process that scores its share of the sample with the numpy oracle.
TEST / MEASUREMENT INFRASTRUCTURE ONLY (the `cpu_baseline` leg of bench.py); never imported by the product.

    python -m oracle.bench_worker <dir> <index> <n_workers>

<dir> holds the inputs as .npy files (memory-mapped here, so all workers share one copy in the page cache) and a pickle
of the small objects.  The worker loads everything, touches the mapped arrays, writes ``ready.<index>``, waits for the
file ``go`` and then scores the chunks ``index, index + n_workers, ...`` of the sample, writing the logits and its own
start / end timestamps.  It imports numpy and the oracle only (no torch, no GPU).
"""
import os
import pickle
import sys
import time

import numpy as np


def main():
    d, idx, nw = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    if root not in sys.path:
        sys.path.insert(0, root)
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=1)   # one BLAS thread per process: the processes are the parallelism
    except ImportError:
        pass
    from oracle import lpformer_oracle as O
    small = pickle.load(open(os.path.join(d, "small.pkl"), "rb"))
    mm = {k: np.load(os.path.join(d, k + ".npy"), mmap_mode="r") for k in
          ("sample", "x_node", "mask_rowptr", "mask_col", "ppr_rowptr", "ppr_col", "ppr_val")}
    sample = np.asarray(mm["sample"])
    okw = dict(x=None, adj_norm=None, adj_mask=(np.asarray(mm["mask_rowptr"]), mm["mask_col"]),
               ppr=(np.asarray(mm["ppr_rowptr"]), mm["ppr_col"], mm["ppr_val"]), P=small["P"], cfg=small["cfg"],
               x_node=mm["x_node"])
    spans = small["spans"][idx::nw]
    O.forward(sample[:, :8], **okw)          # imports, page-ins and first-call costs stay out of the timed part
    open(os.path.join(d, f"ready.{idx}"), "w").close()
    go = os.path.join(d, "go")
    while not os.path.exists(go):
        time.sleep(0.002)
    t0 = time.time()
    out = [O.forward(sample[:, lo:hi], **okw)["logit"] for lo, hi in spans]
    t1 = time.time()
    np.save(os.path.join(d, f"logit.{idx}.npy"), np.concatenate(out) if out else np.zeros(0, np.float32))
    with open(os.path.join(d, f"done.{idx}.tmp"), "w") as f:
        f.write(f"{t0!r} {t1!r}\n")
    os.replace(os.path.join(d, f"done.{idx}.tmp"), os.path.join(d, f"done.{idx}"))


if __name__ == "__main__":
    main()
