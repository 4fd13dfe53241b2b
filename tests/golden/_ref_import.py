"""Import harness for the upstream reference (ONLY usable in the build container).

Used exclusively by tests/golden/make_golden.py to produce golden vectors.  The reference
tree (/root/reference) is read-only and never travels to the GPU box; nothing under tests/
that runs at test time imports this module.  Three third-party modules the reference imports
but this image lacks are stubbed before import (easydict: model.py:6, seaborn: dead import at
model_components.py:7, h5py: eval.py:5 / data_provider.py:8 - used only by dataset ctors we
never call).
"""
import sys
import types

REF_ROOT = "/root/reference"


class EasyDict(dict):
    """Attribute-dict stand-in for easydict==1.9 (requirements.txt:11)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def import_reference():
    m = types.ModuleType("easydict")
    m.EasyDict = EasyDict
    sys.modules.setdefault("easydict", m)
    sys.modules.setdefault("seaborn", types.ModuleType("seaborn"))
    sys.modules.setdefault("h5py", types.ModuleType("h5py"))
    import matplotlib
    matplotlib.use("Agg")
    sys.dont_write_bytecode = True
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import method.model as ref_model
    import method.model_components as ref_comp
    import method.eval as ref_eval
    import method.optimization as ref_optim
    import method.data_provider as ref_data
    return types.SimpleNamespace(model=ref_model, comp=ref_comp, eval=ref_eval, optim=ref_optim,
                                 data=ref_data, EasyDict=EasyDict)
