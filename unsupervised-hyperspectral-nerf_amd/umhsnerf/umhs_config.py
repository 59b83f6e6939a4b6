"""Method registration (mirror of ``umhsnerf/umhs_config.py:34-69``): ``umhs_method`` for nerfstudio's
``nerfstudio.method_configs`` entry point (method name ``umhsnerf``, alias ``umhs``).

With nerfstudio installed this builds the real MethodSpecification/TrainerConfig; without it (offline) it exposes
the same defaults as a plain dict so the hot path and its tests do not depend on nerfstudio."""
from __future__ import annotations

from .umhs_model import UMHSConfig

METHOD_NAME = "umhsnerf"
TRAINER_DEFAULTS = dict(
    method_name=METHOD_NAME, steps_per_eval_batch=500, steps_per_save=2000, max_num_iterations=30000,
    mixed_precision=False,  # fp32 hot path (reference: True -> fp16 autocast on CUDA); parity is quoted vs fp32
    train_num_rays_per_batch=9216 * 4, eval_num_rays_per_batch=4096, eval_num_rays_per_chunk=512,
    optimizers={"fields": {"optimizer": dict(lr=2e-2, eps=1e-15), "scheduler": dict(lr_final=1e-5, max_steps=30000)}},
)

try:  # pragma: no cover - needs nerfstudio
    from nerfstudio.engine.trainer import TrainerConfig  # type: ignore
    from nerfstudio.plugins.types import MethodSpecification  # type: ignore

    from .umhs_pipeline import make_nerfstudio_trainer_config

    umhs_method = MethodSpecification(config=make_nerfstudio_trainer_config(TRAINER_DEFAULTS), description="umhs method (MI355X HIP hot path)")
except Exception:
    umhs_method = {"config": dict(TRAINER_DEFAULTS, model=UMHSConfig(eval_num_rays_per_chunk=512)), "description": "umhs method (MI355X HIP hot path)"}
