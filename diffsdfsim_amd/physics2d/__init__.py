"""2-D analytic contacts: the contact handler of the reference's 2-D world, on the device."""
from .contacts import DiffContactHandler, contacts2d, make_handler, MAXV  # noqa: F401
