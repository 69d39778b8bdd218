"""The bench contract line: ONE JSON line on stdout, <= bench.CONTRACT_MAX_BYTES, that carries what the driver
parses -- the contract's keys, `config`, `roofline`, `cpu_baseline`, `comm`, the compact `modes` map -- for the N = 1 and the N > 1
launch forms; everything else goes to bench_full.json + stderr.  CPU: the emitter on canned records (last round's 25-30 KB ones
and an inflated one).  GPU: the real line of the default command, parsed."""
import copy
import glob
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config", "roofline")
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_avg_ms", "algorithmic_bytes_per_path_step",
                 "path_steps_per_launch")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample")


def canned_records():
    out = []
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[45]_bench_*.json"))):
        txt = open(fn).read().strip()
        try:
            out.append((os.path.basename(fn), json.loads(txt.splitlines()[-1])))
        except ValueError:
            pass
    assert out, "no canned bench records under profiles/"
    return out


def check_line(rec, line, full):
    assert len(line.encode()) <= 6144 and "\n" not in line
    assert json.loads(line) == rec
    for k in CONTRACT_KEYS:
        assert k in rec, k
    assert rec["value"] == full["value"] and rec["ms_per_step"] == full["ms_per_step"]   # the driver's clock check: full precision
    assert rec["config"]["workload"] and rec["config"]["mode"] == full["config"]["mode"]
    for k in ROOFLINE_KEYS:
        assert k in rec["roofline"], k
    r = rec["roofline"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and 0 < r["frac"] < 1
    # consistency-relevant: bytes x units / duration == achieved
    assert abs(r["algorithmic_bytes_per_path_step"] * r["path_steps_per_launch"] / (r["kernel_avg_ms"] * 1e-3) / 1e9 - r["achieved"]) < 1e-3 * r["achieved"] \
        or r["bound"] == "mfma"

    def strings(x):
        if isinstance(x, str):
            yield x
        elif isinstance(x, dict):
            for v in x.values():
                yield from strings(v)
        elif isinstance(x, list):
            for v in x:
                yield from strings(v)
    assert max(len(s) for s in strings(rec)) <= 118


def test_contract_line_of_canned_records_fits_and_keeps_the_head():
    import bench
    for name, full in canned_records():
        if "other_modes" in full:
            assert len(json.dumps(full)) > 19000, name   # these are the lines that outgrew the driver
        rec, line = bench.contract_record(copy.deepcopy(full))
        check_line(rec, line, full)
        if "cpu_baseline" in full:
            for k in CPU_KEYS:
                assert k in rec["cpu_baseline"], k
        if "comm" in full:
            assert rec["comm"]["backend"] == full["comm"]["backend"]
        if "modes" in full:
            assert rec["modes"] == json.loads(json.dumps(bench._num(full["modes"])))
        assert "other_modes" not in rec and "smoothing" not in rec and "box" not in rec


def test_contract_line_of_an_n_gpu_record():
    """the N > 1 forms: per-GPU times, the SURVEY-C4 shard record, the communicator's report of every rank"""
    import bench
    _, full = canned_records()[-1]
    full = copy.deepcopy(full)
    for k in ("other_modes", "smoothing", "box", "cpu_baseline", "modes", "sustained"):
        full.pop(k, None)
    full["n_gpus"] = 8
    full["per_gpu_ms_per_step"] = [1.38 + 0.001 * k for k in range(8)]
    full["comm"] = bench.comm_record([{"nranks": 8, "rccl_nranks": 8, "rank": k, "rccl_rank": k, "rccl_version": 22606} for k in range(8)],
                                     None, "rccl", "bhip_comm_init_rank, one process per GPU")
    full["survey_c4"] = {"chains_per_gpu": 32768, "value": 1.3e12, "unit": "path-steps/s", "ms_per_step": 0.2, "scaling": "weak",
                         "per_gpu_ms_per_step": [0.19] * 8, "roofline": copy.deepcopy(full["roofline"]), "host_issue_us_per_step": 40.0, "note": "x" * 500}
    rec, line = bench.contract_record(full)
    check_line(rec, line, full)
    assert rec["n_gpus"] == 8 and len(rec["per_gpu_ms_per_step"]) == 8 and rec["comm"]["ranks_seen"] == list(range(8)) and rec["comm"]["consistent"]
    assert rec["survey_c4"]["chains_per_gpu"] == 32768 and "frac" in rec["survey_c4"]["roofline"]


def test_contract_line_never_exceeds_the_cap():
    """an inflated record: the compact map is what goes first (loudly), never the head; beyond that the emitter refuses"""
    import bench
    _, full = canned_records()[-1]
    big = copy.deepcopy(full)
    big["modes"] = {f"mode_{k}": [1.0, 0.5, "hbm"] for k in range(400)}
    rec, line = bench.contract_record(big)
    assert len(line.encode()) <= bench.CONTRACT_MAX_BYTES and rec["modes"] == {"dropped": "see bench_full.json"}
    assert rec["roofline"]["kernel"] == full["roofline"]["kernel"]
    huge = copy.deepcopy(full)
    huge["per_gpu_ms_per_step"] = [1.0] * 2000
    with pytest.raises(AssertionError, match="cap"):
        bench.contract_record(huge)


def test_emit_writes_the_full_record_beside_the_line(tmp_path, monkeypatch, capfd):
    import bench
    _, full = canned_records()[-1]
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench.emit_json(copy.deepcopy(full))
    cap = capfd.readouterr()
    lines = [l for l in cap.out.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0].encode()) <= bench.CONTRACT_MAX_BYTES
    assert json.loads(lines[0])["full_record"] == "bench_full.json"
    on_disk = json.load(open(tmp_path / "bench_full.json"))
    assert on_disk["other_modes"] == full["other_modes"] and on_disk["smoothing"] == full["smoothing"]
    assert '"other_modes"' in cap.err                     # and on stderr


@pytest.mark.gpu
def test_real_default_line_parses_with_roofline_and_cpu_baseline():
    """the driver's own N = 1 command: the line on stdout is the only one, parses, fits, and holds roofline + cpu_baseline + modes"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-live-traffic"],
                       capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines[:3]
    assert len(lines[0].encode()) <= 6144
    j = json.loads(lines[0])
    check_line(j, lines[0], j)
    assert j["steps"] == 20 and j["warmup"] == 5 and j["n_gpus"] == 1 and j["dtype"] == "f64" and j["unit"] == "path-steps/s"
    assert abs(j["value"] - j["config"]["path_steps_per_step"] / (j["ms_per_step"] * 1e-3)) < 1e-6 * j["value"]
    for k in CPU_KEYS:
        assert k in j["cpu_baseline"], k
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["value"] > 0
    assert j["comm"]["backend"] == "rccl" and j["comm"]["consistent"]
    assert {"mcmc", "c2", "proposals", "linpro32", "smooth_shared"} <= set(j["modes"])
    assert len(j["modes"]["linpro32"]) == 4 and 0 < j["modes"]["linpro32"][3] < j["modes"]["linpro32"][1]   # the matrix-pipe share beside the flop fraction
    assert j["headline_v2noise"]["frac"] > 0
    full = json.load(open(os.path.join(ROOT, "bench_full.json")))
    assert full["value"] == j["value"] and len(full["other_modes"]) >= 10
