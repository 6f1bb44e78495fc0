from .functions import MSDeformAttnFunction, ms_deform_attn_core_pytorch  # noqa: F401
from .modules import MSDeformAttn  # noqa: F401
