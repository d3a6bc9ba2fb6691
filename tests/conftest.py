import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # build anything that is missing (hipcc cross-compiles without a GPU; a few seconds)
    need = [os.path.join(ROOT, "mach3_amd", "lib", "libm3hip.so"), os.path.join(ROOT, "mach3_amd", "lib", "libm3synth.so"),
            os.path.join(ROOT, "oracle", "libm3oracle_serial.so"), os.path.join(ROOT, "oracle", "libm3oracle_omp.so")]
    if not all(os.path.exists(p) for p in need):
        subprocess.run(["make", "-j8", "-C", ROOT, "all"], check=True, stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def gpu_engine():
    from mach3_amd.engine import Engine
    eng = Engine(0)  # raises M3HipError when there is no device: GPU tests must not pass silently
    yield eng
    eng.close()


@pytest.fixture()
def fresh_engine():
    from mach3_amd.engine import Engine
    eng = Engine(0)
    yield eng
    eng.close()
