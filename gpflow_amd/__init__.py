"""gpflow_amd -- MI355X-native dense-GP hot path with the GPflow model/posterior surface."""
__version__ = "0.1.0"
