"""Host-side mirror of the reference's pybind11 module `cslicer`
(cslicer/pyfrontend.cpp:116-148) over the MI355X HIP engine.  See frontend.py."""
from .frontend import bipatite, cslicer, sample, test_list, test_pyfront  # noqa: F401
