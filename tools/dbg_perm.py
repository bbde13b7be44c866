import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_gpu_configs import _setup
from lpformer_amd import evaluate as E
nfail = 0
for trial in range(40):
    cfg, n, ei, w, x, data, args, model, score, _ = _setup("cora", bs=1024)
    rng = np.random.default_rng(4)
    edges = torch.from_numpy(rng.integers(0, n, size=(5000, 2)))
    if trial == 0:
        loop = torch.cat([score(model(edges[i:i + 1024].t())) for i in range(0, 5000, 1024)])
    junk = torch.randn(trial * 1000 + 10, device="cuda"); del junk
    for rep in range(3):
        sweep = E.score_edges(model, score, edges, batch_size=1024, streams=3)
        torch.cuda.synchronize()
        d = (loop - sweep).abs()
        bad = torch.nonzero(d > 1e-5).flatten().tolist()
        if bad:
            nfail += 1
            print("trial", trial, "rep", rep, "bad idx", bad[:10], sweep[bad[:10]].tolist())
            for k, v in model._ws.items():
                if k[0] == "att_nfull":
                    print("   nfull", k[1], int(v[0]))
                if k[0] == "att_perm_lb":
                    print("   epoch word", k[1], int(v.view(torch.int64)[1024]), v.view(torch.int64)[:8].tolist())
                if k[0] == "att_perm":
                    print("   perm", k[1], v[:48].tolist(), v[896:904].tolist())
print("failures", nfail)
