import os
import sys

import pytest

# the suites run the library's experiment variants against each other (DEMI_K2_MODE, DEMI_DPOR_HOST_BOOKKEEPING, DEMI_JIT_*):
# the library reads such variables only with this switch on (demi_amd/csrc/knobs.hpp); tests/test_host_cpu.py checks the
# gate itself in processes of their own
os.environ.setdefault("DEMI_EXPERIMENT", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_py
    oracle_py.build()
    return oracle_py


@pytest.fixture(scope="session")
def gpu_ctx():
    from demi_amd import _native
    ctx = _native.Context(0)
    yield ctx
    ctx.close()


# ---- the wave64 emulator (tests/emu, TEST INFRASTRUCTURE): DEMI_EMU=1 runs the `gpu` tests that need no torch device against the
# kernel sources compiled for the CPU (tests/test_emu_suite_cpu.py drives that from the CPU suite).  The product never sees it.
if os.environ.get("DEMI_EMU") == "1":
    from tests.emu import build as _emu_build
    _emu = _emu_build.build()
    os.environ["DEMI_NO_TORCH"] = "1"
    os.environ["DEMI_HIPRTC_LIB"] = _emu["hiprtc"]
    from demi_amd import _native as _n
    _n.LIB_PATH = _emu["lib"]


    # what cannot run there: the tests about the real library file, a torch device, RCCL, or bench.py itself
    _EMU_SKIP = {"test_native_library_is_the_one_running", "test_full_size_properties_1m", "test_launches_of_one_ctx_on_two_streams_are_ordered",
                 "test_rccl_communicator_world_of_one", "test_bench_py_two_ranks_on_one_gpu", "test_replay_launches_of_one_ctx_on_two_streams"}

    def pytest_collection_modifyitems(config, items):
        for it in items:
            if it.name.split("[")[0] in _EMU_SKIP:
                it.add_marker(pytest.mark.skip(reason="needs the GPU itself (DEMI_EMU=1 run)"))
