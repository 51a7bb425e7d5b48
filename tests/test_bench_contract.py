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


def test_self_launch_logic():
    """`python bench.py --gpus N` without a launcher starts the N ranks itself (torch.distributed.run as a child process,
    before the GPU is touched); under a launcher (WORLD_SIZE set) or with one GPU it runs in-process."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    a2 = bench.parse_args(["--gpus", "2", "--steps", "5", "--warmup", "1"])
    assert bench.needs_self_launch(a2, {}) is True
    assert bench.needs_self_launch(a2, {"WORLD_SIZE": "2", "RANK": "0"}) is False
    assert bench.needs_self_launch(bench.parse_args([]), {}) is False
    cmd = bench.launcher_command(a2, ["--gpus", "2", "--steps", "5", "--warmup", "1"], 29517)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29517"
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "2", "--steps", "5", "--warmup", "1"]          # the ranks see the same flags
    # the decision is taken before torch / the library are imported in bench.py
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.index("sys.exit(self_launch(") < src.index("import torch")


def test_self_launch_propagates_the_exit_code(tmp_path, monkeypatch):
    """A rank that fails makes the self-launched bench fail (rc != 0), nothing is swallowed."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    bad = tmp_path / "rank.py"
    bad.write_text("import os, sys\nsys.exit(3 if os.environ.get('RANK') == '1' else 0)\n")
    monkeypatch.setattr(bench, "launcher_command",
                        lambda args, argv, port: [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                                                  "--master-addr", "127.0.0.1", "--master-port", str(port), str(bad)])
    assert bench.self_launch(bench.parse_args(["--gpus", "2"]), []) != 0
    bad.write_text("print('ok')\n")
    assert bench.self_launch(bench.parse_args(["--gpus", "2"]), []) == 0
