from .model import BayesianModel, GPModel
from .gpr import GPR
from .svgp import SVGP
from .training_mixins import training_loss, training_loss_closure

__all__ = ["BayesianModel", "GPModel", "GPR", "SVGP", "training_loss", "training_loss_closure"]
