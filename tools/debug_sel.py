"""GPU debug helper: selection (general path) vs the golden fixture, with descriptor dumps."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.golden_util import Fixture
from tests.test_gpu_parity import _build

fx = Fixture("lp_all_d64")
model, _ = _build(fx)
for indexed in (True, False):
    model.use_select_index = indexed
    batch = model._prep_batch(torch.from_numpy(fx["batch"]))
    ws = model._select_device(batch, fx.test_set, None)
    torch.cuda.synchronize()
    bs = batch.shape[1]
    tp = ws.type_ptr.cpu().numpy().reshape(3, bs + 1)
    ent = ws.entries.cpu().numpy().reshape(3, ws.ent_cap, 4)
    desc = ws.desc.cpu().numpy().reshape(-1, 16)[:bs]
    offs = ws.offs.cpu().numpy()
    print("indexed", indexed, "ctl", ws.ctl.tolist(), "totals", tp[:, bs])
    exp = fx["sel_non1hop_ix"]
    for k in (1, 4, 5):
        seg = ent[2, tp[2, k]:tp[2, k + 1]]
        d = desc[k]
        d32 = d.view(np.int32)
        print(" pair", k, "ab", fx["batch"][:, k], "got", seg[:, 1], "want", exp[1][exp[0] == k])
        print("   ra0,rb0,pa0,pb0,ta0,tb0", d[:6], "dA,dB,nPa,nPb,nTa,nTb,a,b", d32[12:20], "xa0,xb0", d[10:12], "dxA,dxB", d32[24:26],
              "offs", offs[k], offs[k + 1])
adj = model._device_graph("mask", model.data["adj_mask"])
rp, col = adj.rowptr.cpu().numpy(), adj.col.cpu().numpy()
t0 = model._device_graph("t0", model.data["ppr"])
t0 = t0.to_host_compact().to_device("cuda:0")
trp, tcol = t0.rowptr.cpu().numpy(), t0.col.cpu().numpy()
for k in (1, 4, 5):
    a, b = fx["batch"][:, k]
    print("pair", k, "N(a)", col[rp[a]:rp[a + 1]], "N(b)", col[rp[b]:rp[b + 1]], "rp", rp[a], rp[b])
    print("   T0a", tcol[trp[a]:trp[a + 1]], "T0b", tcol[trp[b]:trp[b + 1]])
# full dumps for offline analysis
out = {}
for indexed in (True, False):
    model.use_select_index = indexed
    batch = model._prep_batch(torch.from_numpy(fx["batch"]))
    ws = model._select_device(batch, fx.test_set, None)
    torch.cuda.synchronize()
    bs = batch.shape[1]
    out[f"tp_{int(indexed)}"] = ws.type_ptr.cpu().numpy().reshape(3, bs + 1)
    out[f"ent_{int(indexed)}"] = ws.entries.cpu().numpy().reshape(3, ws.ent_cap, 4)
    out["desc"] = ws.desc.cpu().numpy().reshape(-1, 16)[:bs]
    out["offs"] = ws.offs.cpu().numpy()
out["adj_rp"], out["adj_col"] = rp, col
out["t0_rp"], out["t0_col"], out["t0_val"] = trp, tcol, t0.val.cpu().numpy()
np.savez_compressed("gpurun_out/dbg_sel.npz", **out)
