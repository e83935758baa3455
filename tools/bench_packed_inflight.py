"""AC17 packed encrypt + checked decrypt (50 attributes, 16 policies) with several packed calls IN FLIGHT: one host handle and one caller thread per
call (bench.py: packed_inflight_leg), for a few (handles, items per call) settings.   usage: python tools/bench_packed_inflight.py [handles,n ...]"""
import json
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                # noqa: E402
from rabe_amd import hostlib as hl          # noqa: E402
from rabe_amd import hostprep as hp         # noqa: E402
from rabe_amd.schemes import ac17           # noqa: E402


def main():
    legs = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(1, 65536), (2, 32768), (2, 65536), (3, 32768), (4, 16384)]
    host = hl.Host(0)
    attrs = ["a%d" % (i + 1) for i in range(50)]
    pk, msk = ac17.setup(host)
    sk = ac17.cp_keygen(host, msk, attrs)
    prnd = random.Random(2)
    pols = [hp.to_json(hp.random_binary_tree(attrs, prnd)) for _ in range(16)]
    for handles, n in legs:
        r = bench.packed_inflight_leg(hl, ac17, pk, sk, pols, n, handles, 4)
        r.pop("note", None)
        print(json.dumps(r), flush=True)
    host.close()


main()
