"""``imread`` as the reference's scripts use it (one greyscale frame per file): see piv_io.imread."""
import os
import sys

_pkg = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _pkg not in sys.path:
    sys.path.append(_pkg)
from piv_io import imread  # noqa: E402,F401
