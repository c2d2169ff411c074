from .model import BayesianModel, GPModel
from .gpr import GPR
from .sgpr import SGPR
from .svgp import SVGP
from .training_mixins import training_loss, training_loss_closure

__all__ = ["BayesianModel", "GPModel", "GPR", "SGPR", "SVGP", "training_loss", "training_loss_closure"]
