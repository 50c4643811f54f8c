"""MI355X-native Neural-CDE forward/adjoint integrator (drop-in for the reference's
``torchcde.cdeint`` / ``src.ncde.NeuralCDE`` hot path).  Import as ``ncde_amd``."""
from . import data  # noqa: F401
