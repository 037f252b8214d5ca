from .train import train
from .val import evaluate
from . import infer
