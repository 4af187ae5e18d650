"""The driver parses ONE stdout line of bench.py; round 3's 21 KB line did not reach its record (BENCH_r03.parsed = null).
The final line is built by bench.compact_line from the detail object: strict JSON, < 3 KB, whatever the probes returned."""
import json
import math
import os

import bench

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _canned():
    with open(os.path.join(REPO, "profiles", "r3d_bench_line.json")) as f:
        return json.load(f)


def _strict(text):
    def bad(c):
        raise ValueError(c)
    return json.loads(text, parse_constant=bad)


def test_compact_line_is_short_strict_json_with_the_contract_keys():
    line = _canned()
    assert len(json.dumps(line)) > 10000                      # the canned detail object is the 21 KB one
    text = bench.compact_line(line)
    assert "\n" not in text and len(text) < 3072
    got = _strict(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in got, k
    assert "workload" in got["config"] and "model" not in got["config"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in got["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in got["cpu_baseline"], k
    assert abs(got["value"] - line["value"]) <= 1e-5 * line["value"]
    assert abs(got["roofline"]["frac"] - got["roofline"]["achieved"] / got["roofline"]["peak"]) < 1e-4


def test_compact_line_survives_non_finite_values_and_failed_probes():
    line = _canned()
    line["roofline"]["traffic"] = float("nan")
    line["psnr"]["value_db"] = float("inf")
    line["train"] = {"error": "RuntimeError: " + "x" * 5000}
    line["config"]["workload"] = "w" * 5000
    line["cpu_baseline"]["sample"] = "s" * 5000
    line["stages_ms_per_step"] = {f"stage{i}": float(i) for i in range(400)}
    text = bench.compact_line(line)
    assert len(text) < 3072
    got = _strict(text)
    assert got["roofline"]["traffic"] is None and got["psnr"]["value_db"] is None
    assert math.isfinite(got["value"])


def test_emit_prints_exactly_one_stdout_line(capsys, tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    bench.emit(_canned())
    cap = capsys.readouterr()
    lines = cap.out.splitlines()
    assert len(lines) == 1 and len(lines[0]) < 3072
    assert _strict(lines[0])["metric"].startswith("shaded surface points/s")
    assert cap.err.startswith("BENCH_DETAIL ")
    _strict(cap.err[len("BENCH_DETAIL "):])
    _strict(open(tmp_path / "gpurun_out" / "bench_detail.json").read())


def test_compact_line_carries_both_tied_kernels():
    """Round 4: the inner-light kernel and the traversal take the same time within noise; the line holds the figures of both -- the
    one `roofline` describes and the other under `roofline_other` -- and names the longest stage, still under the limit."""
    with open(os.path.join(REPO, "profiles", "r4q_bench_detail.json")) as f:
        line = json.load(f)
    assert line["roofline"]["kernel"] == "inner_light3_kernel" and "bvh_trace_kernel" in line["roofline_other"]
    got = _strict(bench.compact_line(line))
    assert got["roofline"]["kernel"] == "inner_light3_kernel" and got["longest_stage"] in ("inner_light", "bvh_trace")
    other = got["roofline_other"]["bvh_trace_kernel"]
    assert other["bound"] == "hbm" and 0 < other["frac"] < 1 and other["avg_launch_ms"] > 0
    assert got["config"]["inner_light_operands"].startswith("f16x2")
    assert len(bench.compact_line(line)) < 3072


def test_compact_line_carries_the_stricter_readings_of_the_headline():
    """Round 5: `dtype` names the narrowest operand format per net; the fp32-grade and the with-aux-maps values, both forms of the
    dominant kernel's roofline, the hit-fraction probes and the psnr leg under both operand modes ride in the ONE line."""
    with open(os.path.join(REPO, "profiles", "r4q_bench_detail.json")) as f:
        line = json.load(f)
    line["dtype"] = "f32 ABI; f16x3 flow/point nets; f16x2 inner light"
    line["value_fp32_grade"] = 4.98e6
    line["value_with_aux"] = 4.40e6
    line["inner_light_f16x3"]["roofline"] = dict(kernel="inner_light3_kernel<.,3> (64-ray form)", bound="mfma", achieved=445.0, peak=2500.0, unit="TFLOP/s",
                                                 frac=0.178, avg_launch_ms=21.8, executed_tflops=1400.0, frac_executed=0.56)
    line["hit_fraction_probes"] = {"headline": [0.1475, 5.78e6], "scene_points": [0.136, 5.98e6], "fat_torus_scene": [0.25, 4.6e6], "note": "n" * 300}
    line["psnr"]["inner_light_modes"] = {"f16x2": [0.9892, 1.79e-3], "f16x3": [0.9898, 1.8e-3], "same_outlier_points": False, "outliers_in_one_mode_only": 6,
                                         "note": "n" * 300}
    text = bench.compact_line(line)
    assert len(text) < 3072
    got = _strict(text)
    assert got["dtype"].endswith("f16x2 inner light") and got["value_fp32_grade"] == 4.98e6 and got["value_with_aux"] == 4.40e6
    assert got["roofline_f16x3"]["frac"] == 0.178 and got["roofline"]["frac"] > got["roofline_f16x3"]["frac"]
    assert got["hit_fraction_probes"]["fat_torus_scene"] == [0.25, 4.6e6] and "note" not in got["hit_fraction_probes"]
    assert got["psnr"]["inner_light_modes"]["outliers_in_one_mode_only"] == 6 and "note" not in got["psnr"]["inner_light_modes"]
