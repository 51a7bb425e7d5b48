"""The bench line's contract (task statement + SURVEY.md 8d), checked on the committed line of the last profiled run:
one JSON object with the driver's keys, the roofline and cpu_baseline objects, and the parity gates beside the number."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def latest_line():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json")))
    assert files, "no committed bench line under profiles/"
    lines = [l for l in open(files[-1]).read().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "bench.py must print exactly one JSON line"
    return json.loads(lines[0])


def test_driver_keys_and_types():
    d = latest_line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "closest-hit Mrays/s" and d["unit"] == "Mrays/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["scaling"] in ("weak", "strong") and d["vs_baseline"] is None       # nothing published
    assert d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = rays of the whole job / wall time of the timed steps
    assert abs(d["value"] - d["config"]["rays_per_step"] / d["ms_per_step"] / 1e3) / d["value"] < 1e-3


def test_roofline_and_cpu_baseline_objects():
    d = latest_line()
    for name in ("roofline", "roofline_hbm"):
        r = d[name]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in r, (name, k)
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-3
    assert d["roofline"]["bound"] == "valu" and d["roofline"]["unit"] == "TFLOP/s" and d["roofline"]["peak"] == 157.3
    assert d["roofline_hbm"]["bound"] == "hbm" and d["roofline_hbm"]["unit"] == "GB/s" and d["roofline_hbm"]["peak"] == 8000.0
    assert d["roofline"]["traffic"] is None or d["roofline"]["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "Mrays/s" and c["sample"]
    # correctness gates ride beside the throughput number (north_star: mismatch rate stated)
    assert d["parity"]["prim_id_mismatches"] == 0 and d["parity"]["hit_records_bit_exact"] is True
    assert d["parity"]["bvh_walk_prim_id_mismatches"] == 0
