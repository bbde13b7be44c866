"""``torch_geometric.nn.models.LPFormer``-style facade over the same HIP core.

BASELINE.json's north star names the PyG module API.  That class is not part of the reference tree and PyG is not
installed here, so its signature is restated from upstream PyG (>= 2.7) documentation and is *unpinned* (SURVEY.md
section 8b): ``LPFormer(in_channels, hidden_channels, num_gnn_layers=2, gnn_dropout=0.1, num_transformer_layers=1,
num_heads=1, transformer_dropout=0.1, ppr_thresholds=None, gcn_cache=False)`` with
``forward(batch, x, edge_index, ppr_matrix) -> logits`` (score head inside the model, no sigmoid) and the helpers
``propagate``, ``calc_pairwise`` and ``calc_sparse_ppr``.

Numerically it is the reference model (this repo's ``LinkTransformer`` + ``mlp_score``, pinned by the golden vectors)
with the logit taken before the sigmoid.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import graph
from .link_transformer import LinkTransformer, mlp_score
from .ppr import calc_ppr, calc_ppr_gpu


class LPFormer(nn.Module):
    def __init__(self, in_channels: int, hidden_channels: int, num_gnn_layers: int = 2, gnn_dropout: float = 0.1,
                 num_transformer_layers: int = 1, num_heads: int = 1, transformer_dropout: float = 0.1,
                 ppr_thresholds: Optional[Sequence[float]] = None, gcn_cache: bool = False, residual: bool = False,
                 device="cuda"):
        super().__init__()
        th = list(ppr_thresholds) if ppr_thresholds is not None else [0.0, 1e-4, 1e-2]
        if len(th) != 3:
            raise ValueError("ppr_thresholds = [cn, one_hop, more_than_one_hop]")
        self.in_channels, self.hidden_channels = in_channels, hidden_channels
        self.train_args = {"thresh_cn": th[0], "thresh_1hop": th[1], "thresh_non1hop": th[2], "dim": hidden_channels,
                           "trans_layers": num_transformer_layers, "num_heads": num_heads,
                           "att_drop": transformer_dropout, "dropout": transformer_dropout,
                           "gnn_drop": gnn_dropout, "feat_drop": 0.0, "gcn_cache": gcn_cache,
                           "gnn_layers": num_gnn_layers, "residual": residual, "layer_norm": True, "relu": True}
        # the core needs the feature width at construction; graph entries are bound on the first forward
        proto = {"x": torch.zeros(1, in_channels)}
        self.core = LinkTransformer(self.train_args, proto, device=device)
        self.score = mlp_score(self.core.out_dim, self.core.out_dim, 1, 2)
        self._bound = None

    def _bind(self, x: torch.Tensor, edge_index: torch.Tensor, ppr_matrix):
        """(Re)bind graph inputs when they change (identity + version of the tensors)."""
        # identity of the three inputs (strong references: an address can be recycled) + tensor versions
        b = self._bound
        if b is not None and b[0] is x and b[1] is edge_index and b[2] is ppr_matrix and \
                b[3] == (x._version, edge_index._version):
            return
        key = (x, edge_index, ppr_matrix, (x._version, edge_index._version))
        n = x.shape[0]
        ei = edge_index.detach().cpu().numpy()
        data = self.core.data
        data.clear()
        data.update(x=x, num_nodes=n)
        data["adj_t"] = data["full_adj_t"] = graph.csr_from_coo(ei[0], ei[1], np.ones(ei.shape[1], np.float32), n)
        data["adj_mask"] = data["full_adj_mask"] = graph.mask_csr(ei, n, symmetric=True)
        data["ppr"] = data["ppr_test"] = ppr_matrix
        self.core.num_nodes = n
        self.core._graphs.clear()
        self.core._override.clear()
        self.core._x_cache = None
        self.core._z_cache = None
        self.core._y_cache = None
        self.core._enc_cache = None
        self.core._drop_sample_state()     # (the entry sample and the choices made from it belong to the old graph)
        self._bound = key

    def forward(self, batch: torch.Tensor, x: torch.Tensor, edge_index: torch.Tensor, ppr_matrix) -> torch.Tensor:
        """Logits [BS] for the candidate pairs ``batch`` [2, BS].  In training mode (gradients enabled) the autograd path
        of lpformer_amd/train.py: encoder, selection, attention and score head with their dropouts, logits without the
        sigmoid (``BCEWithLogitsLoss``-style loops)."""
        self._bind(x, edge_index, ppr_matrix)
        if self.training and torch.is_grad_enabled():
            from . import train as lpf_train
            return lpf_train.score_train(self.score, self.core(batch), logits=True)
        h = self.core.propagate()
        for _attempt in range(4):
            out = self.core.score_pairs(batch, h, self.score, logits=True)
            # the caller gets logits it will use right away: read the selection status here (a batch that outgrew the
            # workspace sized from earlier batches would otherwise come back as NaN) and score it again if needed
            if self.core.check_selection():
                return out
        raise RuntimeError("LPFormer.forward: the selection workspace could not be sized")

    def propagate(self, x: torch.Tensor, edge_index: torch.Tensor) -> torch.Tensor:
        self._bind(x, edge_index, self.core.data.get("ppr"))
        return self.core.propagate()

    def calc_pairwise(self, batch, X_node, adj_mask=None, ppr_matrix=None):
        if ppr_matrix is not None:
            self.core.data["ppr"] = self.core.data["ppr_test"] = ppr_matrix
        return self.core.calc_pairwise(batch, X_node, adj_mask=adj_mask)[0]

    @staticmethod
    def calc_sparse_ppr(edge_index: torch.Tensor, num_nodes: int, alpha: float = 0.15, eps: float = 5e-5):
        """PPR matrix as a torch sparse COO tensor, bit-identical to the reference's numba code: the MI355X producer
        when ``edge_index`` lives on the GPU, the host C++/OpenMP producer otherwise."""
        if isinstance(edge_index, torch.Tensor) and edge_index.is_cuda:
            return calc_ppr_gpu(edge_index, num_nodes, alpha, eps, device=edge_index.device).to_torch_sparse_coo()
        return calc_ppr(edge_index, num_nodes, alpha, eps).to_torch_sparse_coo()
