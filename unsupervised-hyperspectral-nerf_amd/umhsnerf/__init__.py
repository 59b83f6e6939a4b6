"""umhsnerf -- MI355X-native drop-in for the UMHS / UnMix-NeRF nerfstudio plugin's hot path.

Same package / module / class names as the reference plugin (``umhsnerf.umhs_field.UMHSField``,
``umhsnerf.umhs_renderer.SpectralRenderer``, ``umhsnerf.umhs_config:umhs_method`` ...); the arithmetic
is done by hand-written gfx950 kernels in ``libumhs_hip.so`` (see ``include/umhs_hip.h``).
"""
__version__ = "0.1.0"
