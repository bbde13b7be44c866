"""lpformer_amd -- MI355X-native LPFormer link-scoring forward pass (hand-written gfx950 HIP behind a C ABI).

Public surface mirrors the reference (HarryShomer/LPFormer):
    LinkTransformer, mlp_score          drop-ins for src/models/link_transformer.py / other_models.py
    LPFormer                            torch_geometric.nn.models.LPFormer-style facade (logits out)
    calc_ppr, calc_ppr_gpu, get_ppr     drop-ins for src/util/calc_ppr_scores.py (host OpenMP push / MI355X push)
    evaluate                            encoder-once, device-resident evaluation sweep + ranking metrics
    graph, data                         CSR containers and the data-dict builder
"""
from . import evaluate, graph, mask_delta, readers  # noqa: F401
from .graph import RemovedEdges  # noqa: F401
from .graphed import GraphedScorer, PlannedScorer  # noqa: F401
from .link_transformer import MLP, LinkTransformer, mlp_score  # noqa: F401
from .ppr import calc_ppr, calc_ppr_gpu, get_ppr, load_or_calc_ppr, ppr_coo  # noqa: F401
from .pyg_api import LPFormer  # noqa: F401

__all__ = ["LinkTransformer", "mlp_score", "MLP", "LPFormer", "calc_ppr", "calc_ppr_gpu", "get_ppr",
           "load_or_calc_ppr", "ppr_coo", "graph", "evaluate", "GraphedScorer", "PlannedScorer", "RemovedEdges"]
