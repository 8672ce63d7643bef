"""Alias so that the reference's callers (``import radfoam`` in radfoam_model/scene.py,
render.py, train.py, benchmark.py) pick up the MI355X implementation unchanged."""
from radfoam_amd import *  # noqa: F401,F403
from radfoam_amd import __all__  # noqa: F401
