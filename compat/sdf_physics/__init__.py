"""Module-name alias package (compat/README.md)."""
