import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def hip_lib():
    from ohm_tsd_slam_amd import capi
    return capi.load_library()


# Every object of a test that OWNS device memory (a grid context: 256 MB at cfg 2, 4.7 GB at cfg 3, 18.8 GB at map_size 15; the
# facade's node) is destroyed when the test ends, not when the garbage collector gets to it: a session of 100+ GPU tests otherwise
# keeps tens of GB alive at a time, and a free-memory guard late in the session sees the leftovers of the tests before it.
_owners = []


def _track(cls):
    init = cls.__init__
    if getattr(init, "_tsd_tracked", False):
        return

    def tracked(self, *a, **k):
        init(self, *a, **k)
        _owners.append(self)

    tracked._tsd_tracked = True
    cls.__init__ = tracked


@pytest.fixture(autouse=True)
def _close_device_owners():
    from ohm_tsd_slam_amd import capi, facade
    _track(capi.TsdGridDevice)       # (facade.GridView, the non-owning view, does not run this constructor)
    _track(facade.SlamNode)
    del _owners[:]
    yield
    # nodes first (a node owns its grid context), then grids; sensors / batch slots that outlive their grid were detached by tsd_destroy
    for o in sorted(_owners, key=lambda o: isinstance(o, capi.TsdGridDevice)):
        try:
            o.close()
        except Exception:
            pass
    del _owners[:]
