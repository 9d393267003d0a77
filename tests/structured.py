"""Shim: the structured<->dense helpers live in oracle/lcp_expand.py (shared with bench.py's cpu_baseline leg)."""
from oracle.lcp_expand import *  # noqa: F401,F403
