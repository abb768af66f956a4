"""Named time ranges of the comms drivers: a context manager that shows up as a range in torch.profiler / rocprof marker traces
and adds its host-clock duration to a timer.

Boundary of reference ``train/comms/pt/param_profile.py`` (``paramProfile(timer=None, description="")`` ``:18-40``, ``paramTimer``
``:43-59``): same names and arguments, so driver code written against PARAM (``with paramProfile(timer=t, description="# PARAM
replay 0: block")``) runs unchanged.  The range labels the replay uses are the reference's (commsTraceReplay.py:736-790), so a
chrome trace of ``commsTraceReplay.py --enable-profiler`` reads the same."""
from __future__ import annotations

import logging
import time

from torch.autograd.profiler import record_function

from .comms_utils import paramTimer

logger = logging.getLogger(__name__)

__all__ = ["paramProfile", "paramTimer"]


class paramProfile:
    """``with paramProfile(timer, "label"):`` -- a profiler range named ``label`` around the body; ``timer`` (a paramTimer)
    is advanced by the body's host time, ``intervalNS`` keeps it for the caller"""

    def __init__(self, timer: paramTimer = None, description: str = "") -> None:
        self.description = description
        self.timer = timer
        self.intervalNS = 0.0
        self._range = record_function(description)

    def __enter__(self) -> "paramProfile":
        self._range.__enter__()
        self._t0 = time.monotonic_ns()
        return self

    def __exit__(self, exc_type, exc_value, traceback) -> None:
        self.intervalNS = float(time.monotonic_ns() - self._t0)
        if isinstance(self.timer, paramTimer):
            self.timer.incrTimeNS(self.intervalNS)
        logger.debug(f"{self.description} took {self.intervalNS} ns")
        self._range.__exit__(exc_type, exc_value, traceback)
