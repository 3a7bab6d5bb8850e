import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # the built libraries are git-ignored: build them once if this is a fresh checkout
    need = [ROOT / "hgaprec_amd" / "libhpf_hip.so", ROOT / "hgaprec_amd" / "libhgaprec_host.so",
            ROOT / "hgaprec_amd" / "hgaprec", ROOT / "oracle" / "liborc.so"]
    if not all(p.exists() for p in need):
        import __graft_entry__ as g
        g.build()


@pytest.fixture(scope="session")
def orc():
    """the CPU oracle (test infrastructure)"""
    from oracle import orc as o
    o.lib()
    return o
