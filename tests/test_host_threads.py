"""CPU: the worker-thread copy layer behind the host-pointer entry points (csrc/hostpipe.cpp), driven through its test
hooks from several Python threads at once (ctypes releases the GIL): contents must be exact and nothing may crash -- the
completion latch of a parallel copy lives on the caller's stack, which is the kind of code that fails rarely and badly."""
import ctypes as C
import threading

import numpy as np
import pytest


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    from voiceconversion_jl_amd import _lib
    l = _lib.lib
    l.vcmi_debug_host_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    l.vcmi_debug_host_copy_rows.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int64]
    return l


def test_parallel_copy_exact_from_many_threads(lib):
    rng = np.random.default_rng(0)
    errors = []

    def worker(seed):
        r = np.random.default_rng(seed)
        for it in range(60):
            n = int(r.integers(1, 6_000_000))
            src = r.integers(0, 255, n, dtype=np.uint8)
            dst = np.zeros(n + 16, dtype=np.uint8)
            lib.vcmi_debug_host_copy(dst.ctypes.data + 8, src.ctypes.data, n)
            if not (np.array_equal(dst[8:8 + n], src) and not dst[:8].any() and not dst[8 + n:].any()):
                errors.append((seed, it, n))

    threads = [threading.Thread(target=worker, args=(int(rng.integers(1 << 30)),)) for _ in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors


def test_strided_rows(lib):
    rng = np.random.default_rng(1)
    for rows, row_bytes, ss, ds in [(1, 8, 8, 8), (5000, 320, 328, 320), (70001, 328, 328, 400), (3, 2_000_000, 2_000_008, 2_000_000)]:
        src = rng.integers(0, 255, rows * ss, dtype=np.uint8)
        dst = np.full(rows * ds, 7, dtype=np.uint8)
        lib.vcmi_debug_host_copy_rows(dst.ctypes.data, ds, src.ctypes.data, ss, row_bytes, rows)
        s2, d2 = src.reshape(rows, ss), dst.reshape(rows, ds)
        assert np.array_equal(d2[:, :row_bytes], s2[:, :row_bytes])
        assert np.all(d2[:, row_bytes:] == 7)


def test_set_devices_without_a_gpu_reports_no_device(lib):
    import torch
    import voiceconversion_jl_amd as vc
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(vc.VCMIError, match="no HIP device"):
        vc.set_devices([0])
    vc.set_devices([])                      # removing a (non-existent) group is always fine
    assert vc.get_devices() == []
