#!/usr/bin/env python3
"""Integer-VALU issue-rate calibration on the target GPU (the roofline denominator).
Prints one JSON line per instruction variant."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rabe_amd import Engine  # noqa: E402

NAMES = {0: "v_mad_u64_u32", 1: "v_mul_lo_u32", 2: "v_add_u32", 3: "v_fma_f64", 4: "v_lshl_add_u64",
         5: "fp_mont_mul(8x32 CIOS)", 6: "v_mul_hi_u32"}


def main():
    eng = Engine(0)
    n_cu, name = eng.device_info()
    for v in (2, 0, 1, 6, 4, 3, 5):
        iters = 2000 if v != 5 else 400
        ms, ops = eng.calibrate(v, iters)
        rate = ops / (ms * 1e-3)
        per_clk_cu = rate / (n_cu * 2.4e9)
        print(json.dumps({"variant": NAMES[v], "ms": round(ms, 3), "ops": ops, "Gops_per_s": round(rate / 1e9, 1),
                          "lane_ops_per_clk_per_CU@2.4GHz": round(per_clk_cu, 2), "n_cu": n_cu, "device": name}))
    eng.close()


if __name__ == "__main__":
    main()
