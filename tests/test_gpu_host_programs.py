"""The two host-side test programs of tests/host_san/ on the GPU box, uninstrumented and linked against the product library like any
client (`make host-plain`, built by __graft_entry__.build(); the binaries travel with the tree).  Their instrumented builds
(AddressSanitizer + UBSan on the host code) run in the build container only -- the GPU pool runs no sanitizer builds
(tests/test_sanitized_host.py, tests/host_san/san.mk)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "tests", "host_san")
LIB = os.path.join(ROOT, "detex_amd", "lib", "libdetexhip.so")


def _built(name):
    exe = os.path.join(SAN, name)
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", ROOT, "host-plain"], stderr=subprocess.DEVNULL)
    return exe


@pytest.mark.gpu
def test_entry_points_and_host_tier_program_with_the_device():
    """tests/host_san/api_san_main.cpp: every entry point called with arguments that must be refused, then -- with a device -- decodes
    through the host tier: a launch per call, the resident service (requests, format switches, idle exits and restarts, release with an
    instance lingering), staged textures, every answer compared with the launch path's."""
    r = subprocess.run([_built("api_plain")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "0 problems" in r.stdout and "device part ran" in r.stdout, (r.stdout[-3000:], r.stderr[-4000:])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["threads", "exit", "dlclose"])
def test_thread_and_process_teardown_with_resident_kernels_alive(mode):
    """tests/host_san/teardown_san_main.cpp: threads decode through the host tier -- their resident service kernels alive, idle time
    200 ms -- and exit WITHOUT detexhipReleaseThreadResources(); the process then returns from main() with the main thread's resident
    kernel lingering (`threads`), leaves through exit() from a worker thread (`exit`), or dlclose()s the PRODUCT library, opens and
    uses it again (`dlclose`).  No crash, no hang, every decode right."""
    r = subprocess.run([_built("teardown_plain"), mode] + ([LIB] if mode == "dlclose" else []), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "teardown_san: ok" in r.stdout, (r.returncode, r.stdout[-3000:], r.stderr[-4000:])
