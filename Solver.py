"""Reference-compatible module name (`from Solver import Solver`); see mimrl_amd/Solver.py."""
from mimrl_amd.Solver import Solver  # noqa: F401
