from nmrf_amd.ops.modules import MSDeformAttn  # noqa: F401
