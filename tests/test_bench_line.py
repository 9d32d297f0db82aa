"""The line the driver parses: strict JSON, the contract's keys with `roofline` and `cpu_baseline`, scalars only,
strings the driver does not have to cut, and small — round 5's line had grown to 25 KB and the driver recorded `parsed: null`."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline", "cpu_baseline")


def _strict(text):
    def refuse(name):
        raise ValueError(f"{name} is not JSON")
    return json.loads(text, parse_constant=refuse)


def _stored_full_records():
    out = []
    for name in sorted(os.listdir(os.path.join(ROOT, "profiles"))):
        if name.startswith("r0") and "_bench_v" in name and name.endswith(".json"):
            rec = json.load(open(os.path.join(ROOT, "profiles", name)))
            if isinstance(rec, dict) and "metric" in rec and "e2e_8h" in rec:
                out.append((name, rec))
    return out


@pytest.mark.parametrize("name,full", _stored_full_records()[-4:])
def test_result_line_is_small_strict_and_complete(name, full):
    full = dict(full)
    full["e2e_8h_batch"] = dict(full.get("e2e_8h_batch") or {}, x8={"audio_hours_per_s": float("nan")})     # a NaN from a leg must not reach the line
    text = json.dumps(bench.result_line_of(full), allow_nan=False)
    assert len(text) < bench.RESULT_LINE_MAX_BYTES <= 4096, len(text)
    line = _strict(text)
    for k in CONTRACT:
        assert k in line, k
    assert line["config"]["workload"] and len(line["config"]) <= 13
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(line["roofline"])
    assert abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-4
    for section in (line, line["config"], line["roofline"], line["cpu_baseline"] or {}):
        for k, v in section.items():
            assert not isinstance(v, (list,)) and (not isinstance(v, dict) or section is line), (k, v)    # scalars only below the top level
            if isinstance(v, str):
                assert len(v) <= 120, (k, len(v))
    summary = json.dumps({"summary": bench.summary_of(full)}, allow_nan=False)
    assert len(summary) < 3500 and _strict(summary)["summary"]["errors"] == sorted(k for k, v in full.items() if isinstance(v, dict) and "error" in v)
    assert len(summary) + len(text) < 8000          # both fit the 8 KB tail the driver keeps


def test_legs_are_printed_one_per_line(capsys):
    rec = bench.LegRecord(rank=0)
    rec.update({"metric": "m", "value": 1.0})
    rec["mel"] = {"audio_hours_per_s": 2.0}
    rec["vs_baseline"] = None
    other = bench.LegRecord(rank=1)
    other["mel"] = {"audio_hours_per_s": 3.0}
    lines = capsys.readouterr().out.splitlines()
    assert [_strict(l) for l in lines] == [{"leg": "mel", "result": {"audio_hours_per_s": 2.0}}]
    assert rec["mel"]["audio_hours_per_s"] == 2.0 and other["mel"]["audio_hours_per_s"] == 3.0


def test_multi_rank_record_has_no_cpu_baseline_and_still_parses():
    name, full = _stored_full_records()[-1]
    full = dict(full, n_gpus=8)
    full.pop("cpu_baseline")
    line = _strict(json.dumps(bench.result_line_of(full), allow_nan=False))
    assert line["n_gpus"] == 8 and line["cpu_baseline"] is None
