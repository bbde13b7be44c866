"""Load the committed golden fixtures (tests/golden/*.npz) and rebuild their seed-derived weights."""
import json
import os

import numpy as np

from oracle.fixture_weights import make_state

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LP_CASES = ["lp_all_d64", "lp_all_d128_weighted", "lp_all_d64_residual_valtest", "lp_1hop_d64_dense",
            "lp_all_d256_noln", "lp_all_d64_maskedadj", "lp_all_d64_thcn"]
MASKED_CASES = ["lp_all_d64_maskedadj"]  # also hold the training loop's adjacency-override calls (masked_* keys)
PPR_CASES = ["ppr_push_small", "ppr_push_powerlaw"]


class Fixture:
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.cfg = json.loads(str(self.z["config_json"]))
        self.params = make_state(self.cfg["param_shapes"], self.cfg["seed"])
        self.n = self.cfg["n"]
        self.test_set = self.cfg["test_set"]

    def __getitem__(self, k):
        return self.z[k]

    def __contains__(self, k):
        return k in self.z.files

    # graph used for the recorded call (full_* when test_set, read_datasets.py:97-113)
    @property
    def edge_index(self):
        return self.z["full_edge_index" if self.test_set else "edge_index"].astype(np.int64)

    @property
    def edge_weight(self):
        return self.z["full_edge_weight" if self.test_set else "edge_weight"]

    @property
    def ppr_coo(self):
        p = "ppr_test_" if self.test_set else "ppr_"
        return self.z[p + "row"].astype(np.int64), self.z[p + "col"].astype(np.int64), self.z[p + "val"]

    def state_dicts(self):
        """(model_state, score_state) keyed like the reference's state_dict()."""
        m = {k[len("model."):]: v for k, v in self.params.items() if k.startswith("model.")}
        s = {k[len("score."):]: v for k, v in self.params.items() if k.startswith("score.")}
        return m, s

    def sel_tags(self):
        return [t for t in ("cn", "onehop", "non1hop") if f"sel_{t}_ix" in self.z.files]
