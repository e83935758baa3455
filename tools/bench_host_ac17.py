import random, time, sys
sys.path.insert(0, '.')
from rabe_amd import hostlib as hl, hostprep as hp
from rabe_amd.schemes import ac17
host = hl.Host(0)
rnd = random.Random(22)
attrs = ["a%d" % (i + 1) for i in range(50)]
pk, msk = ac17.setup(host)
sk = ac17.cp_keygen(host, msk, attrs)
policies = [hp.to_json(hp.random_binary_tree(attrs, rnd)) for _ in range(16)]
PT = b"x" * 55
for n in (256, 4096):
    pts = [PT] * n
    pol = [policies[i % 16] for i in range(n)]
    ac17.cp_encrypt_batch(host, pk, pol[:2], pts[:2], hl.JSON_POLICY)
    t0 = time.perf_counter(); cts = ac17.cp_encrypt_batch(host, pk, pol, pts, hl.JSON_POLICY); t1 = time.perf_counter()
    out = ac17.cp_decrypt_batch(host, [sk] * n, cts); t2 = time.perf_counter()
    assert out == pts
    print(n, "encrypt %.3f s  decrypt %.3f s  -> %.0f ops/s" % (t1 - t0, t2 - t1, n / (t2 - t0)))
