"""Break-even of the two forms of the activation-pattern attention at D >= 128 (`LinkTransformer._patterns_pay`,
`PT_EXACT_MAX`): first PE layers scaled by a gain and the LayerNorm offsets spread (the pattern table then covers less and
less of the model's entries), for each setting the flipped units per entry LEFT for the exact path and the launch times of
  PT form:          lpf_select4 -> lpf_pair_attention_rows4 (patterns by table, exact path from the nearest pattern)
  type-major form:  lpf_select4 + lpf_select4_regions -> lpf_pair_attention_rows_perm (every entry looks at its units)
    LPF_CFG=collab python tools/pt_breakeven.py      (D = 128)      LPF_CFG=cora ... (D = 256)"""
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
model.attention_impl = "flip"
encs = [e for e in (model.ppr_encoder_cn, getattr(model, "ppr_encoder_onehop", None),
                    getattr(model, "ppr_encoder_non1hop", None)) if e is not None]
base_w = [e.linears[0].weight.detach().clone() for e in encs]
base_b = [e.norm.bias.detach().clone() for e in encs]
gen = torch.Generator(device=dev); gen.manual_seed(7)
noise = [torch.randn(b.shape, device=dev, generator=gen) for b in base_b]
print(f"# {name}-like, D = {cfg['dim']}, batch {cfg['batch']}: gain  spread  flips raw  left  covered  "
      "PT: select + attention us   type-major: select (+ regions) + attention us")
for gain, spread in [(float(a), float(b)) for a, b in (g.split(":") for g in os.environ.get(
        "LPF_GAINS", "1:0,4:0,8:0.1,16:0.2,32:0.3,64:0.4,128:0.5,256:0.5").split(","))]:
    with torch.no_grad():
        for e, bw, bb, nz in zip(encs, base_w, base_b, noise):
            e.linears[0].weight.copy_(bw * gain)
            e.norm.bias.copy_(bb + spread * nz)
    raw, left = model._flip_stats()
    cov = [s["covered"] for s in model._pattern_tables(model._fold())["stats"] if s["covered"] is not None]
    t = {}
    for form, lim in (("pt", float("inf")), ("tm", -1.0)):
        model.PT_EXACT_MAX = lim
        model._pt_choice = None
        assert model._uses_select4() == (form == "pt")
        for _ in range(3):
            model.score_pairs(batch, h, score)
        torch.cuda.synchronize()
        assert model.check_selection()
        KernelTimer.reset(); KernelTimer.enabled = True
        for _ in range(20):
            model.score_pairs(batch, h, score)
        ks = KernelTimer.summary()
        KernelTimer.enabled = False
        t[form] = (sum(ks[k][2] for k in ("select_run", "select_regions", "select_plan") if k in ks) * 1e3,
                   ks["pair_attention_rows"][2] * 1e3)
    print(f"{gain:7.1f} {spread:5.2f} {raw:9.2f} {left:7.2f}  {min(cov):6.3f}   {t['pt'][0]:6.1f} + {t['pt'][1]:6.1f} = {sum(t['pt']):6.1f}   "
          f"{t['tm'][0]:6.1f} + {t['tm'][1]:6.1f} = {sum(t['tm']):6.1f}", flush=True)
