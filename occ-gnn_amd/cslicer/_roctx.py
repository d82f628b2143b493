"""Profiler range markers (ROCTX) around the phases of the training step -- the counterpart of the
reference's annotations (python/layers/dist_sageconv.py:52-65 `torch.cuda.nvtx.range_push/pop`,
python/train.py:68).  `rocprofv3 --marker-trace` records them next to the kernel trace.

Markers are diagnostics, not part of the data path: without a ROCTX library they are no-ops.
librocprofiler-sdk-roctx.so is what rocprofv3 listens to; libroctx64.so is the older roctracer one."""
import contextlib
import ctypes as C
import os

_push = _pop = None
if not os.environ.get("CSLICER_NO_ROCTX"):
    for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
        try:
            _l = C.CDLL(name)
            _l.roctxRangePushA.argtypes = [C.c_char_p]
            _l.roctxRangePushA.restype = C.c_int
            _l.roctxRangePop.restype = C.c_int
            _push, _pop = _l.roctxRangePushA, _l.roctxRangePop
            break
        except (OSError, AttributeError):
            continue

enabled = _push is not None


def push(name):
    if _push is not None:
        _push(name.encode())


def pop():
    if _pop is not None:
        _pop()


@contextlib.contextmanager
def range(name):  # noqa: A001  (same word as the reference's nvtx.range_push)
    push(name)
    try:
        yield
    finally:
        pop()
