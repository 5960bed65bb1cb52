"""convasr_amd: the MI355X (gfx950) implementation of convasr's hot path -- logmel frontend, Conv1d+BatchNorm+activation
encoder, log-softmax, CTC loss/gradient and the data-parallel training step -- behind the reference's own models.py API.
Importing this package requires libconvasr_hip.so (python -m convasr_amd.build); there is no CPU implementation."""
from . import _lib, ops, functional, models, train, parallel, optimizers, ctc, datasets, transcribe  # noqa: F401
from .models import *  # noqa: F401,F403

__all__ = ['models', 'ops', 'functional', 'train', 'parallel']
