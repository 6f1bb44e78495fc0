from nmrf_amd.models import NMRF, build, build_model  # noqa: F401
