"""pytest wiring.

Markers
  gpu : needs a real MI355X; run on the GPU box with `-m gpu`.  Everything else runs on CPU.

Fixtures
  oracle : Api bound to oracle/libomx_oracle.so (prefix omxo_) — the CPU checker (test infra only)
  omx    : Api bound to the product libomx_hip.so (prefix omx_) — loading it needs no GPU
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, "oracle") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "oracle"))   # exact_f64.py: the f64 referee of exempted columns (tests only)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (gfx950) device")
    config.addinivalue_line("markers", "exemptions_allowed: a fixed-seed reproducer of a parity rule's mechanism (tests/parity.py): it may pass "
                                       "through the rule, and must be arbitrated against exact f64")


# ---- in-suite soak (tests/test_gpu_soak.py): seeds nobody picked, yet the SAME seeds whenever the same tree is tested.  The base is a
# digest of the product's sources and the parity rules (the GPU box receives the tree without .git, so the commit hash itself is not
# there to read): unseen until the tree exists, reproducible after — a driver run and a builder run of one commit judge the same 60
# sequences (the clock-derived base made a red GPUTEST unreproducible).  OMX_SOAK_SEED overrides it; clock / counter
# bases for wider soaks are tools/soak_suite.sh's job.
def _tree_soak_base():
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "openmeters_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "openmeters_amd", "csrc", "*.[ch]pp"))
                   + glob.glob(os.path.join(ROOT, "include", "*.h")) + [os.path.join(ROOT, "tests", "parity.py")])
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return int(h.hexdigest()[:12], 16) % 1_000_000 + 1


SOAK_BASE = int(os.environ.get("OMX_SOAK_SEED", "0")) or _tree_soak_base()


def soak_seeds(count, salt):
    """`count` seeds for soak case number `salt`, disjoint from the suite's fixed seeds (those are < 1000)"""
    return [SOAK_BASE * 1000 + salt * 40 + i for i in range(count)]


def pytest_collection_modifyitems(config, items):
    """The soak cases run LAST: their seeds are new with every change of the sources, so under `-x` a seed that trips a bar stops the
    session only after every fixed-seed test has been run and reported."""
    items.sort(key=lambda item: item.fspath.basename == "test_gpu_soak.py")   # (stable: everything else keeps its order)


def pytest_report_header(config):
    return f"soak seed base (digest of the source tree; OMX_SOAK_SEED overrides): {SOAK_BASE}"


@pytest.fixture(autouse=True)
def _exemption_scope(request):
    """parity.EXEMPTIONS_ALLOWED is on for the clock-seeded soak cases only: a fixed-seed test that passes through an exemption rule
    (tests/parity.py) fails — the rules exist for seeds nobody picked, and every fixed seed must hold on the plain bars."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import parity
    soak = request.node.fspath.basename == "test_gpu_soak.py" or request.node.get_closest_marker("exemptions_allowed") is not None
    parity.EXEMPTIONS_ALLOWED = soak
    before = len(parity.FIXED_SEED_EXEMPTIONS)
    yield
    parity.EXEMPTIONS_ALLOWED = False
    if not soak and os.environ.get("OMX_ALLOW_FIXED_SEED_EXEMPTIONS") != "1":
        new = parity.FIXED_SEED_EXEMPTIONS[before:]
        assert not new, f"fixed-seed test needed {len(new)} parity exemption(s): {new[:3]}"


def pytest_sessionfinish(session, exitstatus):
    path = os.environ.get("OMX_PARITY_REPORT")
    if path:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import parity
        if parity.LEDGER:
            parity.write_ledger(path)


def _build_oracle() -> str:
    path = os.path.join(ROOT, "oracle", "libomx_oracle.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in os.listdir(os.path.join(ROOT, "oracle"))
            if f.endswith((".cpp", ".hpp"))] + [os.path.join(ROOT, "include", "omx.h")]
    stale = (not os.path.exists(path)) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs)
    if stale:
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    return path


@pytest.fixture(scope="session")
def oracle():
    from openmeters_amd.capi import Api
    return Api(_build_oracle(), "omxo_")


@pytest.fixture(scope="session")
def omx_tuning():
    """the tuning library (make TUNING=1): the superseded kernel forms and the A/B builds, for cross-checks; None when it is not built"""
    from openmeters_amd.capi import Api
    path = os.path.join(os.path.dirname(__file__), "..", "openmeters_amd", "csrc", "libomx_hip_tuning.so")
    return Api(path, "omx_") if os.path.exists(path) else None


@pytest.fixture(scope="session")
def omx():
    import openmeters_amd
    return openmeters_amd.api()


def _backend_params():
    return ["oracle", pytest.param("hip", marks=pytest.mark.gpu)]


@pytest.fixture(params=_backend_params())
def backend(request):
    """Api under test: the CPU oracle (runs everywhere) or the HIP product (-m gpu)."""
    if request.param == "oracle":
        return request.getfixturevalue("oracle")
    api = request.getfixturevalue("omx")
    import openmeters_amd
    assert openmeters_amd.device_available(), "gpu-marked test but no gfx950 device is visible"
    return api
