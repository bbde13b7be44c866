"""Probe: is one gather of 1-KB rows cheaper than two gathers of 512-byte rows from two tables?  (the side launch of a
step reads X[a], X[b] and Y[a], Y[b]: would a table [X | Y] pay?)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lpformer_amd import _lib
dev = torch.device("cuda:0")
n, bs = 235_868, 32_768
lib = _lib.hip()
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(0)
batches = [torch.randint(0, n, (2, bs), generator=g).to(dev) for _ in range(16)]
def run(width, tables, outs, reps=200):
    def once(i):
        b = batches[i % 16]
        for t, o in zip(tables, outs):
            lib.lpf_pair_gather_f32(bs, width, b.data_ptr(), b.stride(0), n, t.data_ptr(), t.stride(0), None, 0,
                                    o.data_ptr(), o.stride(0), st)
    for i in range(20): once(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(reps): once(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6
x, y = torch.randn(n, 128, device=dev), torch.randn(n, 128, device=dev)
xy = torch.randn(n, 256, device=dev)
o1, o2, o3 = torch.empty(bs, 128, device=dev), torch.empty(bs, 128, device=dev), torch.empty(bs, 256, device=dev)
print(f"two gathers of 512-byte rows from two tables: {run(128, [x, y], [o1, o2]):.1f} us")
print(f"one gather of 1-KB rows from one table:       {run(256, [xy], [o3]):.1f} us")
print(f"one gather of 512-byte rows:                  {run(128, [x], [o1]):.1f} us")
xs = xy[:, :128]
print(f"one gather of 512-byte rows, row stride 1 KB: {run(128, [xs], [o1]):.1f} us")
