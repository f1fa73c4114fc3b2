"""Calculator adapters with the reference's plugin API (`/root/reference/plugin/*_interface`).
Third-party drivers (ase, cslib, i-pi) are imported lazily: none of them is needed to evaluate
energies and forces through `model_calc`."""
from .ase_interface import NNCalculator, build_graph, model_calc  # noqa: F401
