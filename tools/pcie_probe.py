"""What the link and the host memory system give a packed call: pinned H2D / D2H, pageable D2H (hipMemcpy stages it), and the parallel
memcpy between a pinned staging buffer and pageable memory that the host layer's record assembly amounts to.
usage: python tools/pcie_probe.py [MB]"""
import ctypes
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rabe_amd import Engine          # noqa: E402


def main():
    mb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    n = mb << 20
    eng = Engine(0)
    dev = eng.alloc(n)
    pin = eng.host_alloc(n)
    ctypes.memset(pin, 1, n)
    page = np.ones(n, dtype=np.uint8)
    out = {"MB": mb, "cores": os.cpu_count()}

    def best(fn, reps=5):
        b = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            b = min(b, time.perf_counter() - t0)
        return round(n / b / 1e9, 2)

    def h2d():
        eng.upload_async(dev, pin, n)
        eng.sync()

    def d2h():
        eng.download_async(pin, dev, n)
        eng.sync()

    def d2h_page():
        eng._check(eng.lib.rhip_download(eng.ctx, page.ctypes.data_as(ctypes.c_void_p), dev.ptr, ctypes.c_size_t(n)))

    def h2d_page():
        eng._check(eng.lib.rhip_upload(eng.ctx, dev.ptr, page.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n)))

    out["pinned_h2d_GBps"] = best(h2d)
    out["pinned_d2h_GBps"] = best(d2h)
    out["pageable_d2h_GBps"] = best(d2h_page)
    out["pageable_h2d_GBps"] = best(h2d_page)
    src = np.ctypeslib.as_array((ctypes.c_uint8 * n).from_address(pin.value))
    out["numpy_copy_pinned_to_pageable_1thread_GBps"] = best(lambda: np.copyto(page, src))
    print(json.dumps(out))
    eng.host_free(pin)
    eng.close()


main()
