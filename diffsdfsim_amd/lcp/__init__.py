from .lcp import LCPFunction  # noqa: F401
