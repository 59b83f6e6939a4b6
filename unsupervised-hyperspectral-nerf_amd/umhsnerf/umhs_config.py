"""Method registration (mirror of ``umhsnerf/umhs_config.py:34-69``): ``umhs_method`` for nerfstudio's
``nerfstudio.method_configs`` entry point (``pyproject.toml``: ``umhsnerf = 'umhsnerf.umhs_config:umhs_method'``).

With nerfstudio importable ``umhs_method`` is a real ``MethodSpecification(config=TrainerConfig(method_name="umhsnerf", ...))``
whose pipeline / datamanager / model configs are this package's classes and whose ``"fields"`` optimizer is the fused HIP Adam.
Only a missing nerfstudio (ImportError) selects the offline form below -- any other failure while building the
specification surfaces.  Offline, ``umhs_method`` holds the same configuration objects in a plain namespace, so the hot path and
its tests do not depend on nerfstudio."""
from __future__ import annotations

from dataclasses import dataclass, field
from types import SimpleNamespace
from typing import Any, Optional, Tuple, Type

from .data.umhs_datamanager import UMHSDataManagerConfig
from .data.umhs_dataparser import UMHSDataParserConfig
from .optim import UMHSAdam
from .umhs_model import UMHSConfig
from .umhs_pipeline import UMHSPipelineConfig

METHOD_NAME = "umhsnerf"  # umhs_config.py:36, pyproject.toml:13, every script (scripts/hotdog.sh:1)
ALIAS_NAME = "umhs"  # what the reference's README tells users to type (README.md:11: ``ns-train umhs``)
DESCRIPTION = "umhs method (MI355X HIP hot path)"


def make_pipeline_config() -> UMHSPipelineConfig:
    """umhs_config.py:42-58: datamanager 9216*4 train / 4096 eval rays per batch, model eval chunks of 512 rays."""
    return UMHSPipelineConfig(
        datamanager=UMHSDataManagerConfig(dataparser=UMHSDataParserConfig(), train_num_rays_per_batch=9216 * 4, eval_num_rays_per_batch=4096),
        model=UMHSConfig(eval_num_rays_per_chunk=512),
    )


TRAINER_FIELDS = dict(  # umhs_config.py:35-41,66-67
    method_name=METHOD_NAME, steps_per_eval_batch=500, steps_per_save=2000, max_num_iterations=30000,
    mixed_precision=False,  # the hot path is fp32 (reference: True = fp16 autocast around tcnn); parity is quoted against fp32
    save_only_latest_checkpoint=False, vis="viewer",
)
OPTIMIZER_FIELDS = dict(lr=2e-2, eps=1e-15)  # AdamOptimizerConfig(lr=2e-2, eps=1e-15), umhs_config.py:61
SCHEDULER_FIELDS = dict(lr_final=0.00001, max_steps=30000)  # ExponentialDecaySchedulerConfig, umhs_config.py:62


def make_nerfstudio_method(method_name: str = METHOD_NAME):
    """The reference's ``MethodSpecification`` with this package's classes.  Raises ImportError without nerfstudio.
    nerfstudio keys the discovered methods by ``config.method_name``, so the ``umhs`` alias is a second specification of its own
    (same configuration, other name), not a second entry point onto the same object."""
    from nerfstudio.configs.base_config import ViewerConfig
    from nerfstudio.engine.optimizers import AdamOptimizerConfig
    from nerfstudio.engine.schedulers import ExponentialDecaySchedulerConfig
    from nerfstudio.engine.trainer import TrainerConfig
    from nerfstudio.plugins.types import MethodSpecification

    @dataclass
    class UMHSAdamOptimizerConfig(AdamOptimizerConfig):
        """Adam for param group "fields": one fused launch over the flat buffer (+ clamp_endmembers, + the gradient exchange).  The
        learning rate comes from nerfstudio's scheduler through ``param_group["lr"]`` (UMHSAdam's own decay stays off)."""

        _target: Type = UMHSAdam

    return MethodSpecification(
        config=TrainerConfig(
            **dict(TRAINER_FIELDS, method_name=method_name),
            pipeline=make_pipeline_config(),
            optimizers={"fields": {"optimizer": UMHSAdamOptimizerConfig(**OPTIMIZER_FIELDS),
                                   "scheduler": ExponentialDecaySchedulerConfig(**SCHEDULER_FIELDS)}},
            viewer=ViewerConfig(num_rays_per_chunk=1 << 12),
        ),
        description=DESCRIPTION,
    )


def _offline_method(method_name: str):
    return SimpleNamespace(
        config=SimpleNamespace(**dict(TRAINER_FIELDS, method_name=method_name), pipeline=make_pipeline_config(),
                               optimizers={"fields": {"optimizer": dict(OPTIMIZER_FIELDS, _target=UMHSAdam), "scheduler": dict(SCHEDULER_FIELDS)}}),
        description=DESCRIPTION,
    )


try:
    umhs_method = make_nerfstudio_method(METHOD_NAME)
    umhs_alias_method = make_nerfstudio_method(ALIAS_NAME)
except ImportError:  # nerfstudio is not installed: same configuration, plain containers
    umhs_method, umhs_alias_method = _offline_method(METHOD_NAME), _offline_method(ALIAS_NAME)
