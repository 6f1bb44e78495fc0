"""Importable stand-in for the reference's compiled extension module of the same name (ops/src/vision.cpp:13-16)."""
from nmrf_amd.ops.functions import ms_deform_attn_backward, ms_deform_attn_forward  # noqa: F401
