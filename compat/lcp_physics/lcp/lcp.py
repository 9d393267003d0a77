"""`lcp_physics.lcp.lcp` of the reference (LCPFunction, lcp.py:43-214), served by the MI355X build (see compat/README.md)."""
from diffsdfsim_amd.lcp.lcp import LCPFunction  # noqa: F401
