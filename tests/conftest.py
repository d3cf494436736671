import json
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parents[1]
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))
GOLDEN = REPO / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "isolated: the test body runs in a python process of its own (planets of >= 10 M cells: a native fault "
                                       "fails that one test with the child's stderr in the report instead of taking the interpreter, and the tests after it, down)")
    config.addinivalue_line("markers", "soak: long create / erode / destroy cycles in one process; collected last")


CHILD_ENV = "WO_TEST_CHILD"


def pytest_collection_modifyitems(config, items):
    """Light tests first, the isolated >= 10 M-cell planets after them, the soak test last (stable inside each class): with the driver's
    `-x`, whatever stops the run stops it as late as possible, and every row of SURVEY section 8 has had its small-size parity test by then."""
    def klass(item):
        if item.get_closest_marker("soak"):
            return 2
        return 1 if item.get_closest_marker("isolated") else 0
    items.sort(key=klass)


@pytest.hookimpl(tryfirst=True)
def pytest_pyfunc_call(pyfuncitem):
    """A test marked `isolated` is run as `python -m pytest <its node id>` in a child process (which finds WO_TEST_CHILD set and runs the
    body); the parent reports the child's exit status with the tail of its stdout and stderr."""
    import os
    import subprocess
    if pyfuncitem.get_closest_marker("isolated") is None or os.environ.get(CHILD_ENV):
        return None
    tm = pyfuncitem.get_closest_marker("timeout")
    limit = float(tm.args[0]) if tm and tm.args else float(pyfuncitem.config.getini("timeout") or 600)
    env = dict(os.environ, **{CHILD_ENV: "1"})
    cmd = [sys.executable, "-X", "faulthandler", "-m", "pytest", pyfuncitem.nodeid, "-x", "-q", "-s", "-p", "no:cacheprovider", "-o", f"timeout={int(limit)}"]
    try:
        r = subprocess.run(cmd, cwd=str(REPO), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=max(limit - 10, 30))
    except subprocess.TimeoutExpired as ex:
        pytest.fail(f"isolated test timed out after {limit:.0f} s\n--- stdout tail\n{(ex.stdout or b'')[-3000:].decode(errors='replace')}\n--- stderr tail\n{(ex.stderr or b'')[-3000:].decode(errors='replace')}", pytrace=False)
    out, err = r.stdout.decode(errors="replace"), r.stderr.decode(errors="replace")
    if r.returncode != 0:
        how = f"killed by signal {-r.returncode}" if r.returncode < 0 else f"exit status {r.returncode}"
        pytest.fail(f"isolated test failed in its child process ({how})\n--- stdout tail\n{out[-4000:]}\n--- stderr tail\n{err[-4000:]}", pytrace=False)
    sys.stdout.write(out[-2000:])
    return True


def load_golden(name):
    return np.load(GOLDEN / f"{name}.npz")


def golden_cases(g):
    return json.loads(bytes(g["cases_json"]).decode())


POST_TAGS = ("N2000_s1", "N2000_s2", "N10000_s1")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle
