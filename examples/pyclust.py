"""pysmallk-style ``Hierclust`` / ``Flatclust`` classes: they live in the package (``smallk_amd/pyclust.py``); this module
keeps the example scripts' ``from pyclust import ...`` working."""
from smallk_amd.pyclust import *          # noqa: F401,F403
from smallk_amd.pyclust import Clustering, Flatclust, Hierclust      # noqa: F401
