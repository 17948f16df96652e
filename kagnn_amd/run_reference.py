"""Run one of the reference's scripts UNCHANGED on the HIP path:

    python -m kagnn_amd.run_reference /path/to/KAGNN/node_classification_clean/time_model.py [script args]

The reference's scripts reach the hot path through three plain module names -- ``from ekan import KAN, KANLinear``,
``from fastkan import FastKAN, FastKANLayer``, ``from models import GNN_Nodes, GKAN_Nodes, GFASTKAN_Nodes``
(``node_classification_clean/time_model.py:13-15``, ``utils.py:8``, every ``optuna_*`` script) -- resolved against the script's
own directory.  This launcher installs ``sys.modules['ekan' | 'fastkan' | 'models']`` aliases that point at this package's
modules BEFORE the script starts, puts the script's directory on ``sys.path`` exactly as ``python script.py`` would (its
``utils.py`` etc. are the reference's own and load from there), and ``runpy``s the file: no edit of the reference tree, no
import swap.  Everything else the script needs (torch_geometric loaders, ogb datasets, optuna) is the user's environment.

Which ``models`` surface is installed follows the script's directory (``--flavour`` overrides):
  node                  ``node_classification_clean/models.py``: GKAN_Nodes, GFASTKAN_Nodes, the conv layers
  graph_classification  ``graph_classification/models.py``: KAGIN, FASTKAGIN, KAGCN, KAGAT, FASTKAGCN, FASTKAGAT
  graph_regression      ``graph_regression/models.py``: the same class NAMES with GINE messages + encoders (``KAGINRegression`` ...)
The MLP baselines of those files (``GNN_Nodes``, ``GIN``, ``GCN``, ``GAT``: stock torch_geometric models, outside the hot path) pass
through to the reference's OWN classes when the directory's ``models.py`` imports (i.e. torch_geometric is installed); otherwise
the name resolves to a class whose constructor says so.
"""
from __future__ import annotations

import importlib.util
import os
import runpy
import sys
import types

_BASELINES = {"node": ("GNN_Nodes",), "graph_classification": ("GIN", "GCN", "GAT"), "graph_regression": ("GIN", "GCN")}


def _flavour_of(script: str) -> str:
    """node | graph_classification | graph_regression, from the ``models.py`` next to the script, else from the script's own text
    (which names it imports from ``models``) and path"""
    script_dir = os.path.dirname(script)
    text = ""
    for path in (os.path.join(script_dir, "models.py"), script):
        if os.path.exists(path):
            with open(path, "r", errors="replace") as f:
                text += f.read()
    if "GKAN_Nodes" in text or "GNN_Nodes" in text or "node_classification" in script_dir:
        return "node"
    if "GINEConv" in text or "AtomEncoder" in text or "graph_regression" in script_dir:
        return "graph_regression"
    return "graph_classification"


def _missing(name: str, why: str):
    class _Unavailable:
        def __init__(self, *a, **k):
            raise ImportError(f"{name} is one of the reference's stock torch_geometric baselines (outside the KAN-GNN hot path this "
                              f"package replaces); it is taken from the reference's own models.py, which could not be imported here: {why}")
    _Unavailable.__name__ = _Unavailable.__qualname__ = name
    return _Unavailable


def install(flavour: str, script_dir: str | None = None) -> types.ModuleType:
    """Install the ``ekan`` / ``fastkan`` / ``models`` aliases; returns the ``models`` module object."""
    import kagnn_amd
    from kagnn_amd import ekan, fastkan, graph_models, models as node_models
    if flavour not in _BASELINES:
        raise ValueError("flavour must be 'node', 'graph_classification' or 'graph_regression'")
    sys.modules["ekan"] = ekan
    sys.modules["fastkan"] = fastkan
    mod = types.ModuleType("models")
    mod.__doc__ = f"kagnn_amd's {flavour} model surface under the reference's module name (kagnn_amd.run_reference)"
    mod.KAN, mod.KANLinear, mod.FastKAN, mod.FastKANLayer = ekan.KAN, ekan.KANLinear, fastkan.FastKAN, fastkan.FastKANLayer
    if flavour == "node":
        for n in ("KANLayer", "FKANLayer", "KAGCNConv", "KAGATConv", "GIKANLayer", "FASTKAGCNConv", "FASTKAGATConv", "GIFASTKANLayer",
                  "GKAN_Nodes", "GFASTKAN_Nodes", "make_kan", "make_fastkan"):
            setattr(mod, n, getattr(node_models, n))
    else:
        for n in ("KANLayer", "FKANLayer", "make_kan", "make_fastkan"):
            setattr(mod, n, getattr(node_models, n))
        for n in ("KAGCN_Layer", "KAGAT_Layer", "FASTKAGCN_Layer", "FASTKAGAT_Layer", "AtomEncoder", "BondEncoder"):
            setattr(mod, n, getattr(graph_models, n))
        if flavour == "graph_classification":
            for n in ("KAGIN", "FASTKAGIN", "KAGCN", "KAGAT", "FASTKAGCN", "FASTKAGAT"):
                setattr(mod, n, getattr(graph_models, n))
        else:
            mod.KAGIN, mod.FASTKAGIN = graph_models.KAGINRegression, graph_models.FASTKAGINRegression
            mod.KAGCN, mod.FASTKAGCN = graph_models.KAGCNRegression, graph_models.FASTKAGCNRegression
    # the stock baselines: the reference's own classes when its models.py imports (torch_geometric present)
    ref_models, why = None, "no models.py next to the script"
    path = None if script_dir is None else os.path.join(script_dir, "models.py")
    if path and os.path.exists(path):
        try:
            spec = importlib.util.spec_from_file_location("_kagnn_reference_models", path)
            ref_models = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(ref_models)          # (its `from ekan import ...` already resolves to this package)
        except Exception as ex:                           # noqa: BLE001 -- typically ModuleNotFoundError: torch_geometric
            ref_models, why = None, f"{type(ex).__name__}: {ex}"
    for n in _BASELINES[flavour]:
        setattr(mod, n, getattr(ref_models, n) if ref_models is not None and hasattr(ref_models, n) else _missing(n, why))
    mod.__kagnn_amd__ = kagnn_amd.__version__
    sys.modules["models"] = mod
    return mod


def main(argv=None) -> None:
    argv = list(sys.argv[1:] if argv is None else argv)
    flavour = None
    while argv and argv[0].startswith("--"):
        if argv[0] == "--flavour" and len(argv) > 1:
            flavour, argv = argv[1], argv[2:]
        elif argv[0].startswith("--flavour="):
            flavour, argv = argv[0].split("=", 1)[1], argv[1:]
        else:
            break
    if not argv:
        raise SystemExit("usage: python -m kagnn_amd.run_reference [--flavour node|graph_classification|graph_regression] "
                         "<reference script.py> [script args]")
    script = os.path.abspath(argv[0])
    script_dir = os.path.dirname(script)
    install(flavour or _flavour_of(script), script_dir)
    sys.argv = [script] + argv[1:]
    sys.path.insert(0, script_dir)                        # what `python script.py` does
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
