"""bench.py's printed line (benchkit/lib.py: compact_line / emit_line): the driver parses the LAST stdout line out of a bounded tail, and
round 5's 20 KB object did not survive it (BENCH_r05.parsed = null).  The compact form must stay small, strict JSON and complete."""
import json
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchkit import lib  # noqa: E402

FIXTURE = os.path.join(ROOT, "tests", "fixtures", "bench_result_r05.json")          # a full result object as round 5's bench.py built it


def canned():
    return json.load(open(FIXTURE))


def strict(text):
    def no_const(name):
        raise ValueError("non-finite constant %s in the line" % name)
    return json.loads(text, parse_constant=no_const)


def test_compact_line_is_small_strict_and_complete():
    c = lib.compact_line(canned())
    text = json.dumps(c, allow_nan=False, separators=(",", ":"))
    assert len(text) < lib.COMPACT_LIMIT == 4096, len(text)
    assert "\n" not in text
    d = strict(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["config"]["workload"] and "model" not in d["config"]
    for k in ("bound", "kernel", "kernel_ms", "items_per_launch", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-3
    for k in ("value", "unit", "cores", "kind"):
        assert k in d["cpu_baseline"], k
    assert d["value_lone_batch"] and d["value_end_to_end"]
    for k in ("3", "4", "5"):
        e = d["configs"][k]
        assert e["value"] > 0 and e["kernel_ms"] > 0 and 0 < e["frac"] < 1 and e["cpu_baseline"]["value"] > 0
    assert len(d["dtype"]) <= 24


def test_every_source_path_in_the_line_exists():
    """a `*_source` key names committed evidence: the file has to be in the tree (round 5's line cited two deleted ones)"""
    import bench
    for kernel in ("k_miller_multi_rr", "k_final_exp_rr", "k_ac17_enc_rows"):
        traffic, src = bench.pmc_traffic(kernel)
        if src is not None:
            assert os.path.exists(os.path.join(ROOT, src)), src
            assert traffic > 0
        vi = bench.pmc_valu_issue(kernel)
        if vi is not None:
            assert os.path.exists(os.path.join(ROOT, vi["source"])), vi["source"]
    assert bench.pmc_traffic("k_miller_multi_rr")[1] is not None, "no committed PMC traffic summary names the dominant kernel"


def test_non_finite_numbers_become_null():
    r = canned()
    r["roofline"]["frac"] = float("nan")
    r["value_lone_batch"] = float("inf")
    c = lib.compact_line(r)
    strict(json.dumps(c, allow_nan=False))
    assert c["roofline"].get("frac") is None and "value_lone_batch" not in c


def test_oversized_inputs_are_cut_not_overflowed():
    r = canned()
    r["config"]["workload"] = "w" * 5000
    r["cpu_baseline"]["sample"] = "s" * 5000
    r["dtype"] = "d" * 500
    for i in range(40):
        r["configs"]["9_extra%d" % i] = dict(r["configs"]["3_ragged"])
    c = lib.compact_line(r)
    text = json.dumps(c, allow_nan=False, separators=(",", ":"))
    assert len(text) <= lib.COMPACT_LIMIT
    assert c["roofline"]["kernel"] and c["cpu_baseline"]["value"]
    assert all(k in c.get("configs", {}) for k in ("3", "4", "5")), "the BASELINE configs go last"


def test_multi_gpu_line_keeps_the_gather_block():
    r = canned()
    r["n_gpus"] = 8
    r.pop("cpu_baseline")
    r.pop("configs")
    r["gather"] = {"backend": "nccl", "collective": "all_gather_into_tensor", "ms": 1.2, "bytes": 8 * 4096 * 384, "matches_unsharded_order": True,
                   "note": "x" * 900}
    c = lib.compact_line(r)
    assert c["gather"]["backend"] == "nccl" and c["gather"]["matches_unsharded_order"] is True and "note" not in c["gather"]
    assert c["gather"]["collective"] == "all_gather_into_tensor"
    assert len(json.dumps(c, separators=(",", ":"))) < lib.COMPACT_LIMIT


def test_emit_line_prints_one_json_line_on_stdout_and_writes_the_detail(tmp_path):
    """stdout = exactly one line, the compact object; the whole object goes to bench_detail.json and to stderr"""
    code = ("import json, sys; sys.path.insert(0, %r); from benchkit import lib; "
            "lib.write_detail.__defaults__ = (%r,); lib.emit_line(json.load(open(%r)))" % (ROOT, str(tmp_path), FIXTURE))
    pr = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert pr.returncode == 0, pr.stderr.decode()[-2000:]
    lines = pr.stdout.decode().splitlines()
    assert len(lines) == 1 and len(lines[0]) < 4096
    d = strict(lines[0])
    assert d["value"] == canned()["value"] and d["detail"] == "bench_detail.json"
    full = strict(open(tmp_path / "bench_detail.json").read())
    assert full["object_api"] and full["single_batch"] and full["configs"]["3"]["workload"]
    err = pr.stderr.decode()
    assert err.startswith("bench_detail: ") and strict(err[len("bench_detail: "):].splitlines()[0])["value"] == d["value"]


def test_sub_runs_hand_the_whole_object_over_the_pipe():
    code = ("import json, sys; sys.path.insert(0, %r); from benchkit import lib; lib.emit_line(json.load(open(%r)))" % (ROOT, FIXTURE))
    env = dict(os.environ, RABE_BENCH_FULL_LINE="1")
    pr = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120, env=env)
    d = strict(pr.stdout.decode().splitlines()[-1])
    assert d["timed_regions"]["count"] and d["roofline"]["kernels_ms"]
    assert not math.isnan(d["value"])
