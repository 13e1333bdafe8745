"""CW attack family.  Exports the class names of the reference's CW package (CW/__init__.py:1-14); the sub-modules are
imported on first use, so ``from hit_adv_amd.CW import CWKNN`` does not pull in scikit-learn for the cluster attacks."""
import importlib

_EXPORTS = {
    'CWPerturb': 'Perturb', 'CWPerturbT': 'PerturbT', 'CWAdd': 'Add', 'CWKNN': 'kNN', 'CWUKNN': 'UKNN', 'CWAOF': 'AOF',
    'CWTAOF': 'TAOF', 'CWUAdvPC': 'UAdvPC', 'CWAdvPC': 'AdvPC', 'CWUAEAOF': 'UAEAOF', 'CWAddClusters': 'Add_Cluster',
    'CWAddObjects': 'Add_Objects',
}
__all__ = sorted(_EXPORTS)


def __getattr__(name):
    if name in _EXPORTS:
        return getattr(importlib.import_module('.' + _EXPORTS[name], __name__), name)
    raise AttributeError("module %r has no attribute %r" % (__name__, name))
