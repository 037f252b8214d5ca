from . import logger
from .timer import TimeAverager, calculate_eta
from .loss_utils import loss_computation, check_logits_losses
from .utils import (load_entire_model, load_pretrained_model, resume, worker_init_fn, save_array, save, load)
from . import train_profiler
from . import metric
