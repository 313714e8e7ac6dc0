"""Import harness for the *reference* (Yukayo/MobGT at /root/reference) -- fixture generation ONLY.

This file is never imported by the product, the tests, smoke() or bench.py.  It exists so that
`tests/golden/make_golden.py` can be re-run in the build container (where /root/reference is
mounted read-only) to regenerate the committed golden vectors.  Nothing from the reference is
copied: the Cython module is compiled into a scratch dir under /tmp, third-party packages that
are absent from the image are replaced by empty stub modules, and the reference's Python is
imported in place with bytecode writing disabled.
"""
import os
import sys
import types
import subprocess

REF = "/root/reference/graphormer"
SCRATCH = "/tmp/mobgt_ref_scratch"


def _build_algos():
    os.makedirs(SCRATCH, exist_ok=True)
    so = [f for f in os.listdir(SCRATCH) if f.startswith("algos.") and f.endswith(".so")]
    if not so:
        import shutil
        shutil.copy(os.path.join(REF, "algos.pyx"), os.path.join(SCRATCH, "algos.pyx"))
        setup = (
            "from setuptools import setup, Extension\n"
            "from Cython.Build import cythonize\n"
            "import numpy\n"
            "setup(ext_modules=cythonize([Extension('algos', ['algos.pyx'], "
            "include_dirs=[numpy.get_include()])], language_level=2))\n"
        )
        with open(os.path.join(SCRATCH, "setup_algos.py"), "w") as f:
            f.write(setup)
        subprocess.check_call([sys.executable, "setup_algos.py", "build_ext", "--inplace"],
                              cwd=SCRATCH, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    if SCRATCH not in sys.path:
        sys.path.insert(0, SCRATCH)
    import algos  # noqa: F401
    return algos


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install(dataset_info=None):
    """Register stubs and make `import model, model_fqandtoyo, collator, wrapper` work."""
    import torch
    import torch.nn as nn

    sys.dont_write_bytecode = True
    algos = _build_algos()

    class LightningModule(nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device("cpu")

        def log(self, *a, **k):
            pass

    class LightningDataModule:
        pass

    _stub("pytorch_lightning", LightningModule=LightningModule, LightningDataModule=LightningDataModule)

    class _Any:
        def __init__(self, *a, **k):
            pass

    tg = _stub("torch_geometric")
    tg.datasets = _stub("torch_geometric.datasets")
    tg.nn = _stub("torch_geometric.nn", GCNConv=_Any, GATConv=_Any, MessagePassing=_Any)
    tg.utils = _stub("torch_geometric.utils", to_undirected=None, add_self_loops=None, degree=None)
    ogb = _stub("ogb")
    ogb.graphproppred = _stub("ogb.graphproppred", PygGraphPropPredDataset=_Any)
    ogb.lsc = _stub("ogb.lsc")
    ogb.lsc.pcqm4m_pyg = _stub("ogb.lsc.pcqm4m_pyg", PygPCQM4MDataset=_Any)
    _stub("owndata", Foursquare=_Any, FoursquareGraph=_Any, ToyotaGraph=_Any)

    info = dataset_info or {}

    def get_dataset(name):
        d = {"num_class": 1, "evaluator": None, "metric": "NLLLoss", "loss_fn": nn.NLLLoss(ignore_index=0)}
        d.update(info.get(name, {}))
        return d

    _stub("data", get_dataset=get_dataset)
    import pyximport
    pyximport.install = lambda *a, **k: None
    if REF not in sys.path:
        sys.path.insert(0, REF)
    return algos
