"""mi355_copy_batch_dev (include/mi355_h264_frame.h): n independent byte copies in one launch — device to device-visible
host memory is what the bridge's dispatcher uses it for; here device to device, checked by reading both back."""
import ctypes as C

import numpy as np


class CopyJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("bytes", C.c_uint64)]


def run(lib, sizes=(16, 4096, 65536 + 48, 1 << 20, 0, 333 * 16)):
    lib.mi355_malloc.restype = C.c_void_p
    lib.mi355_malloc.argtypes = [C.c_size_t]
    rng = np.random.default_rng(7)
    srcs, dsts, hosts = [], [], []
    jobs = (CopyJob * len(sizes))()
    for i, n in enumerate(sizes):
        a = rng.integers(0, 256, max(n, 16), dtype=np.uint8)
        s, d = lib.mi355_malloc(a.nbytes + 32), lib.mi355_malloc(a.nbytes + 32)
        assert s and d
        guard = np.full(a.nbytes + 32, 0xEE, np.uint8)
        assert lib.mi355_memcpy_h2d(C.c_void_p(d), C.c_void_p(guard.ctypes.data), C.c_size_t(guard.nbytes)) == 0
        assert lib.mi355_memcpy_h2d(C.c_void_p(s), C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes)) == 0
        jobs[i] = CopyJob(s, d, n)
        srcs.append(s); dsts.append(d); hosts.append(a)
    d_jobs = lib.mi355_malloc(C.sizeof(jobs))
    assert lib.mi355_memcpy_h2d(C.c_void_p(d_jobs), C.c_void_p(C.addressof(jobs)), C.c_size_t(C.sizeof(jobs))) == 0
    fn = lib.mi355_copy_batch_dev
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
    assert fn(d_jobs, len(sizes), max(sizes), None) == 0
    assert lib.mi355_sync(None) == 0
    for n, a, d in zip(sizes, hosts, dsts):
        got = np.zeros(a.nbytes + 32, np.uint8)
        assert lib.mi355_memcpy_d2h(C.c_void_p(got.ctypes.data), C.c_void_p(d), C.c_size_t(got.nbytes)) == 0
        assert np.array_equal(got[:n], a[:n]) and (got[n:] == 0xEE).all(), n       # exactly n bytes, nothing past them
    assert fn(None, 1, 16, None) != 0 and fn(d_jobs, 0, 16, None) != 0                # argument checks, no launch
    for p in srcs + dsts + [d_jobs]:
        lib.mi355_free(C.c_void_p(p))
    return len(sizes)
