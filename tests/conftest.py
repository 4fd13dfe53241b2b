import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
for p in (os.path.join(ROOT, "dl-dkd_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "golden"),
          os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no libdldkd_hip.so (built artefacts are git-ignored): build it once, like
    __graft_entry__.build() does, so the ABI / planner tests (CPU) and every GPU test find it."""
    so = os.path.join(ROOT, "dl-dkd_amd", "dldkd_amd", "libdldkd_hip.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "dl-dkd_amd", "csrc"), "-j4"], check=True)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def in_child_process(fn):
    """Run a test that creates an RCCL process group in a pytest CHILD process of its own.  destroy_process_group() of this RCCL
    build aborts the interpreter now and then (always at teardown - never in a collective - and not reproducibly: about one run in
    ten of the suite); in the suite's own process that abort would take every later test with it.  The child runs exactly this test
    and, when the body has returned, leaves through os._exit(0) without tearing the group down; a failing assertion is reported by
    the child's pytest and comes back as a non-zero exit code with its output."""
    import functools
    import inspect
    import subprocess

    @functools.wraps(fn)
    def wrapper(*a, **k):
        name = fn.__module__ + "::" + fn.__name__
        if os.environ.get("DLDKD_TEST_CHILD") == name:
            fn(*a, **k)
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(0)
        node = f"{inspect.getsourcefile(fn)}::{fn.__name__}"
        log = []
        for attempt in range(3):
            # a child killed by a SIGNAL (the runtime's abort: a background thread of the process group, seen with and without the
            # teardown) says nothing about the test and is run again; a child that exits with pytest's own code 1 has a failing
            # assertion and is reported at once
            r = subprocess.run([sys.executable, "-m", "pytest", node, "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"],
                               env=dict(os.environ, DLDKD_TEST_CHILD=name, TORCH_NCCL_ENABLE_MONITORING="0"), capture_output=True,
                               text=True, timeout=1500, cwd=ROOT)
            log.append(f"attempt {attempt}: exit {r.returncode}\n{r.stdout[-3000:]}\n{r.stderr[:1500]}\n...\n{r.stderr[-1500:]}")
            if r.returncode >= 0:
                break
        assert r.returncode == 0, f"child pytest of {name}:\n" + "\n".join(log)
    return wrapper
