import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def pytest_sessionstart(session):
    """ZIGZ_TEST_SLEEPING_WAITS=1 runs the whole GPU suite with zigz_device_set_blocking_sync(0, 1): the waits of every
    commit job then poll the completion word in pinned memory instead of asking the runtime (one extra pass of the suite per
    round; the default pass uses the runtime's waits)."""
    if os.environ.get("ZIGZ_TEST_SLEEPING_WAITS") == "1":
        import zigz_amd
        if zigz_amd.device_count() > 0:
            zigz_amd._ffi.lib.zigz_device_set_blocking_sync(0, 1)
