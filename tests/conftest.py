import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure), built on demand with gcc."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def host_gjk(tmp_path_factory):
    """csrc/gjk_true.h (the textbook GJK behind the robust entry points) compiled for the host with g++."""
    import ctypes as C
    import subprocess
    import numpy as np
    so = str(tmp_path_factory.mktemp("tgjk") / "libtrue_gjk_host.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC",
                           os.path.join(REPO, "tests", "native", "true_gjk_host.cpp"), "-o", so])
    lib = C.CDLL(so)

    def run(p1, p2, eps=1e-10, abs_tol=1e-12, max_iter=64):
        p1 = np.ascontiguousarray(p1, dtype=np.float64).reshape(-1, 3)
        p2 = np.ascontiguousarray(p2, dtype=np.float64).reshape(-1, 3)
        out = np.zeros(11)
        lib.true_gjk_host(p1.ctypes.data_as(C.c_void_p), C.c_int(len(p1)), p2.ctypes.data_as(C.c_void_p), C.c_int(len(p2)),
                          C.c_double(eps), C.c_double(abs_tol), C.c_int(max_iter), out.ctypes.data_as(C.c_void_p))
        return dict(dist=out[0], lower=out[1], c1=out[2:5].copy(), c2=out[5:8].copy(), flag=int(out[8]), iters=int(out[9]),
                    status=int(out[10]))
    return run
