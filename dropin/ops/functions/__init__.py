from nmrf_amd.ops.functions import MSDeformAttnFunction, ms_deform_attn_core_pytorch  # noqa: F401
