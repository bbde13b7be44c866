"""Timing aid: lpf_gemm_tn_f32 (dW = dY^T X, csrc/gemm_f32.hip) at the shapes a training step calls it with."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lpformer_amd import train

dev = torch.device("cuda:0")
for m, n, k in ((235868, 128, 128), (576289, 64, 64), (180000, 128, 128), (16384, 128, 128), (16384, 256, 128), (16384, 128, 260)):
    a, b = torch.randn(m, n, device=dev), torch.randn(m, k, device=dev)
    for _ in range(3):
        train._gemm_tn(a, b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        train._gemm_tn(a, b)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    gb = (m * (n + k) * 4) / 1e9
    print(f"M={m} N={n} K={k}: {us:.1f} us per call (partial + reduce + allocation), {gb / us * 1e6:.0f} GB/s of the two inputs, "
          f"{2 * m * n * k / us / 1e6:.1f} TFLOP/s")
