from nmrf_amd.config import CfgNode, get_cfg  # noqa: F401
