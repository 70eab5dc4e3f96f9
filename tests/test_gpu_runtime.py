"""Runtime helpers around the driver's background wipe of freed device memory (host/runtime.c, DESIGN.md section 3 "SpMV
placement"): the driver's "VRAM in use" figure follows allocations, keeps counting memory that was freed until it is wiped,
and DflWaitDeviceMemoryQuiet returns once it has stopped falling.  Build-defined (the reference has no counterpart)."""
import ctypes as C
import time

import pytest

pytestmark = pytest.mark.gpu
GB = 1 << 30


@pytest.fixture(scope="module")
def api():
    from dedflow_amd import api as A
    A.lib()  # raises if the HIP library is missing: no fallback
    return A


def test_memory_in_use_follows_allocations_and_the_wait_outlasts_the_wipe(api):
    L, H = api.lib(), api.hip()
    L.Init(0, None)
    base = L.DflDeviceMemoryInUse()
    if base < 0:
        pytest.skip("rocm_smi not available on this box")
    assert L.DflWaitDeviceMemoryQuiet(30.0) >= 0.0          # whatever earlier tests freed is wiped after this
    base = L.DflDeviceMemoryInUse()
    p = C.c_void_p()
    nbytes = 48 * GB
    api._chk(H.hipMalloc(C.byref(p), nbytes))
    api._chk(H.hipMemset(p, 1, nbytes))
    api._chk(H.hipDeviceSynchronize())
    held = L.DflDeviceMemoryInUse()
    assert held - base > 40 * GB, (base, held)
    api._chk(H.hipFree(p))
    t0 = time.perf_counter()
    waited = L.DflWaitDeviceMemoryQuiet(30.0)
    wall = time.perf_counter() - t0
    after = L.DflDeviceMemoryInUse()
    assert waited >= 0.0 and wall < 31.0
    assert abs(after - base) < 2 * GB, (base, held, after)   # back to where it was once the wait is over
    # a second call right away finds (next to) nothing left to wait for
    t0 = time.perf_counter()
    assert 0.0 <= L.DflWaitDeviceMemoryQuiet(30.0) < 2.0
    assert time.perf_counter() - t0 < 2.5
