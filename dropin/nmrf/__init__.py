from nmrf_amd import __version__  # noqa: F401
