"""Module-name alias package (compat/README.md): the 2-D world of lcp_physics is not rebuilt."""
