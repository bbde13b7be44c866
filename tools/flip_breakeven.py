"""Break-even of the two fp32 one-pass attention kernels: first PE layers scaled by a gain, for
each gain the mean flipped hidden units per selected entry (LinkTransformer.flips_per_entry) and the launch time of the
activation-pattern kernel (pair_flip.hip) and of the matrix-core kernel (pair_fused.hip) on the config's own batch.
    LPF_CFG=collab python tools/flip_breakeven.py      (D = 128)      LPF_CFG=ddi ... (D = 256)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lpformer_amd
from lpformer_amd import data as D
from lpformer_amd.profile import KernelTimer
name = os.environ.get("LPF_CFG", "collab")
cfg = D.CONFIGS[name]
n = cfg["n"]; dev = torch.device("cuda:0")
ei, w = D.chung_lu_graph(n, cfg["edges"], gamma=cfg["gamma"], seed=0, max_weight=cfg["max_weight"])
x = np.random.default_rng(1).standard_normal((n, cfg["f_in"])).astype(np.float32)
data = D.build_data(ei, x, n, edge_weight=w, eps=cfg["eps"], ppr_device=dev)
torch.manual_seed(0)
model = lpformer_amd.LinkTransformer(D.train_args_for(cfg), data, device=dev).to(dev).eval()
score = lpformer_amd.mlp_score(model.out_dim, model.out_dim, 1, 2).to(dev).eval()
batch = torch.from_numpy(D.sample_pairs(ei, n, cfg["batch"], seed=1000)).to(dev)
h = model.propagate()
model.use_side_stream = False
encs = [e for e in (model.ppr_encoder_cn, getattr(model, "ppr_encoder_onehop", None),
                    getattr(model, "ppr_encoder_non1hop", None)) if e is not None]
base = [e.linears[0].weight.detach().clone() for e in encs]
print(f"# {name}-like, D = {cfg['dim']}, batch {cfg['batch']}: gain  flips/entry  flip_us  mfma_us")
for gain in [float(g) for g in os.environ.get("LPF_GAINS", "1,2,4,8,12,16,24,32,48,64,128").split(",")]:
    with torch.no_grad():
        for e, b in zip(encs, base):
            e.linears[0].weight.copy_(b * gain)
    flips = model.flips_per_entry()
    t = {}
    for impl in ("flip", "mfma"):
        model.attention_impl = impl
        for _ in range(3):
            model.score_pairs(batch, h, score)
        torch.cuda.synchronize()
        KernelTimer.reset(); KernelTimer.enabled = True
        for _ in range(20):
            model.score_pairs(batch, h, score)
        t[impl] = KernelTimer.summary()["pair_attention_fused"][2] * 1e3
        KernelTimer.enabled = False
    model.attention_impl = "auto"
    print(f"{gain:7.1f} {flips:10.3f} {t['flip']:9.1f} {t['mfma']:9.1f}   auto -> {model.attention_kernel()}", flush=True)
