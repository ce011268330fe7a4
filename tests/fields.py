"""Synthetic fields (one definition for tests, smoke and bench: nalu_amd/synth.py)."""
from nalu_amd.synth import density, quadratic, quadratic_grad, velocity  # noqa: F401
