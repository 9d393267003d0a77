"""The reference's 2-D layer on the device library (SURVEY.md section 8a R18 and the 2-D halves of R1 / R9 / R15): the analytic
contact handler as a kernel pair (`contacts`), and a world of circles and convex polygons that steps on it and on the dense LCP
kernels (`world`; BASELINE configs[0] end to end without the reference)."""
from .contacts import DiffContactHandler, contacts2d, make_handler, MAXV  # noqa: F401
from .world import Body, Circle, Defaults, Gravity, Hull, Rect, TotalConstraint, World, run_world  # noqa: F401
