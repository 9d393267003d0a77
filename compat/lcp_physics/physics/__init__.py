"""Module-name alias package (compat/README.md): of the 2-D world of lcp_physics the contact handler (`contacts`) is built, its host classes are not."""
