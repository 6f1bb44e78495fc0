from nmrf_amd.frame_utils import InputPadder  # noqa: F401
