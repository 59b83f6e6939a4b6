"""Data side of the hot path: on-disk format reader, resident image stacks, pixel sampler, ray generator."""
