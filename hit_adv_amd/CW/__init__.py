"""CW attack family of the reference (CW/__init__.py:1-14); same class names."""
from .Perturb import CWPerturb  # noqa: F401
from .PerturbT import CWPerturbT  # noqa: F401
from .Add import CWAdd  # noqa: F401
from .kNN import CWKNN  # noqa: F401
from .UKNN import CWUKNN  # noqa: F401
from .AOF import CWAOF  # noqa: F401
from .TAOF import CWTAOF  # noqa: F401
from .UAdvPC import CWUAdvPC  # noqa: F401
from .AdvPC import CWAdvPC  # noqa: F401
from .UAEAOF import CWUAEAOF  # noqa: F401
from .Add_Cluster import CWAddClusters  # noqa: F401
from .Add_Objects import CWAddObjects  # noqa: F401
