from __future__ import annotations

from dataclasses import dataclass
from enum import Enum, auto
from typing import Any, Callable, Dict, List, Optional


@dataclass
class TrainingCallbackAttributes:
    optimizers: Optional[Any] = None
    grad_scaler: Optional[Any] = None
    pipeline: Optional[Any] = None
    trainer: Optional[Any] = None


class TrainingCallbackLocation(Enum):
    BEFORE_TRAIN_ITERATION = auto()
    AFTER_TRAIN_ITERATION = auto()
    AFTER_TRAIN = auto()


class TrainingCallback:
    def __init__(self, where_to_run: List[TrainingCallbackLocation], func: Callable, update_every_num_iters: Optional[int] = None,
                 iters: Optional[tuple] = None, args: Optional[List] = None, kwargs: Optional[Dict] = None):
        assert "step" in func.__code__.co_varnames, f"'step: int' must be an argument in the callback function 'func': {func.__name__}"
        self.where_to_run, self.update_every_num_iters, self.iters, self.func = where_to_run, update_every_num_iters, iters, func
        self.args = args if args is not None else []
        self.kwargs = kwargs if kwargs is not None else {}

    def run_callback(self, step: int) -> None:
        if self.update_every_num_iters is not None:
            if step % self.update_every_num_iters == 0:
                self.func(*self.args, **self.kwargs, step=step)
        elif self.iters is not None:
            if step in self.iters:
                self.func(*self.args, **self.kwargs, step=step)

    def run_callback_at_location(self, step: int, location: TrainingCallbackLocation) -> None:
        if location in self.where_to_run:
            self.run_callback(step=step)
