"""kagnn_amd -- MI355X-native KAN-GNN layer hot path (drop-in for RomanBresson/KAGNN's layers).

Host side: Python mirrors of the reference's ``nn.Module`` surface (``ekan``, ``fastkan``,
``models``, ``graph_models``).  Compute: ``libkagnn_hip.so`` (hand-written HIP for gfx950) behind the
C ABI of ``include/kagnn_hip.h``, reached through ``kagnn_amd.ops``.
"""
from . import _lib, ops                                                          # noqa: F401
from .ekan import KAN, KANLinear                                                 # noqa: F401
from .fastkan import FastKAN, FastKANLayer, RadialBasisFunction, SplineLinear    # noqa: F401
from .models import (FASTKAGATConv, FASTKAGCNConv, FKANLayer, GFASTKAN_Nodes,        # noqa: F401
                     GIFASTKANLayer, GIKANLayer, GKAN_Nodes, KAGATConv, KAGCNConv, KANLayer)

from .norm import BatchNorm1d                                                    # noqa: F401
from .graph_models import (AtomEncoder, BondEncoder, FASTKAGAT, FASTKAGCN, FASTKAGCNRegression, FASTKAGIN, GINEKANLayer,   # noqa: F401
                           KAGAT, KAGCN, KAGCNRegression, KAGIN, KAGINRegression, FASTKAGINRegression,
                           KAGCN_Layer, KAGAT_Layer, FASTKAGCN_Layer, FASTKAGAT_Layer)

__version__ = "0.1.0"
