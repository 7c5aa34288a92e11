from . import smc
from .sp import Algorithm, Marginal, SampleDistribution, Target, marginal
from . import requests

__all__ = ["smc", "requests", "Algorithm", "Marginal", "SampleDistribution", "Target", "marginal"]
