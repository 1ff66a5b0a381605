"""bench.py's contract with the driver, without a GPU: the last thing on stdout is ONE compact JSON line (< 4 KB) that
carries the contract's keys plus `roofline` and `cpu_baseline` however much the sections put into the result, and
`python bench.py --gpus N` as typed is dispatched to the one-process group path instead of exiting."""
import io
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def canned(n_gpus=1):
    long = "x" * 5000
    return {
        "metric": bench.METRIC, "value": 412345.6, "unit": "Msamples/s", "n_gpus": n_gpus, "steps": 200, "warmup": 10,
        "ms_per_step": 0.1987, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int16 data, int32 accumulate",
        "data": long, "parity_gate": long,
        "config": {"workload": long, "frames_per_gpu_per_step": 128, "samples_per_step": 81920000, "parallelism": long,
                   "also_measured": {"k%d" % i: long for i in range(50)}},
        "roofline": {"bound": "hbm", "kernel": "hvk_k_direct<1, 1, 1, 0, 1, 1>", "achieved": 1650.1, "peak": 8000.0, "unit": "GB/s",
                     "frac": 0.2063, "traffic": 733605888, "algorithmic_bytes_per_launch": 327680000, "avg_launch_ms": 0.1986,
                     "launches_timed": 200, "path_frac": 0.2061, "other": {"nested": long}},
        "cpu_baseline": {"value": 78.9, "unit": "Msamples/s", "cores": 3, "kind": "reference", "sample": long, "runs": [1, 2, 3]},
        "multi_gpu": {"gather_backend": "rccl 2.x", "nested": {"a": long}, "note": long},
        "also": {"scalar_%d" % i: 1234.5 + i for i in range(400)},
        "baseline_configs": {"1": {"note": long}}, "moving_pictures": {"note": long}, "secam_l": {"note": long}, "5_one_hour": {"note": long},
    }


def test_the_line_stays_under_4_kb_and_keeps_the_contract():
    line = bench.compact_line(canned())
    assert len(line.encode()) <= bench.LINE_LIMIT and "\n" not in line
    d = json.loads(line)
    for k in CONTRACT:
        assert k in d, k
    assert d["vs_baseline"] is None and d["value"] == 412345.6 and d["config"]["frames_per_gpu_per_step"] == 128
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"]
    assert d["roofline"]["frac"] == 0.2063 and "other" not in d["roofline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"]
    assert len(d["cpu_baseline"]["sample"]) <= 120
    assert "baseline_configs" not in d and "moving_pictures" not in d and "nested" not in d["multi_gpu"]


def test_the_line_is_the_last_thing_on_stdout_and_the_detail_goes_to_the_sidecar(tmp_path, capsys):
    side = str(tmp_path / "detail.json")
    logs = []
    bench.emit(canned(), side, logs.append)
    out = capsys.readouterr().out
    assert out.endswith("\n") and out.count("\n") == 1          # one line, nothing behind it
    d = json.loads(out)
    assert d["roofline"]["kernel"].startswith("hvk_k_direct") and d["detail"] == side
    full = json.load(open(side))
    assert "secam_l" in full and "5_one_hour" in full and len(full["also"]) == 400


@pytest.mark.parametrize("argv, want", [(["--gpus", "2"], ("group", [0, 1])), (["--gpus", "4", "--devices", "0,0,1,1"], ("group", [0, 0, 1, 1])),
                                         (["--gpus", "1"], ("one", None)), ([], ("one", None))])
def test_gpus_n_as_typed_takes_the_group_path_and_does_not_exit(monkeypatch, capsys, argv, want):
    import bench_multi
    seen = []
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.setattr(bench_multi, "headline_group", lambda args, devices, log: seen.append(("group", devices)) or canned(len(devices)))
    monkeypatch.setattr(bench, "headline_one", lambda args, log: seen.append(("one", None)) or canned(1))
    bench.main(argv + ["--detail-out", ""])
    assert seen == [want]
    d = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert d["n_gpus"] == (len(want[1]) if want[1] else 1)


def test_under_torchrun_the_ranks_path_is_taken(monkeypatch, capsys):
    import bench_multi
    seen = []
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "1")
    monkeypatch.setattr(bench_multi, "headline_ranks", lambda args, log: seen.append("ranks") or None)
    bench.main(["--gpus", "2", "--detail-out", ""])
    assert seen == ["ranks"] and capsys.readouterr().out == ""      # ranks other than 0 print nothing
