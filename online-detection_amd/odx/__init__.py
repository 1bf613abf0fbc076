"""odx — MI355X-native hot path of hsp-iit/online-detection (FALKON classifiers, RLS box
regressors, scoring) behind the reference's Python module API.  See DESIGN.md."""
from . import hip  # noqa: F401  (ctypes binding; loading is lazy)
from . import options  # noqa: F401  (the one options table: odx.options.current() / set() / override())
from .backend import get_backend, set_backend  # noqa: F401
from .solver import SolverOptions, falkon_fit, falkon_fit_lockstep  # noqa: F401
from .falkon import Falkon, FalkonOptions, GaussianKernel, InCoreFalkon  # noqa: F401

__all__ = ["hip", "options", "get_backend", "set_backend", "SolverOptions", "falkon_fit", "falkon_fit_lockstep", "Falkon", "InCoreFalkon",
           "GaussianKernel", "FalkonOptions"]
