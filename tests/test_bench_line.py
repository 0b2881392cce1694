"""The driver keeps a 2000-character tail of bench.py's stdout and parses its LAST line: the headline must stay one
short JSON object whatever the secondary blocks grow to (round 4's 22 KB line came back as `parsed: null`)."""
import importlib.util
import io
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def test_compact_line_is_short_and_complete(tmp_path, capsys):
    bench = _bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench.json")))      # a real 22 KB record
    # worst case: every optional figure present at full float width, N > 1
    full["n_gpus"] = 8
    full["scaling_efficiency_vs_rank0_alone"] = 0.98765432101234
    full["from_host_pcm_ms_per_step"] = 1234.56789012345
    full["parity"]["per_frame_ll_from_pcm"] = {"max_rel": 1.23456789e-5, "frames": 10 ** 7}
    for k in ("sustained_mfma", "sustained_mfma_streamed"):
        full["roofline"][k] = {"executed_tflops": 1712.3456789, "clock_mhz": 1634.56789, "frac_executed_of_sustained": 0.87654321}
    out = tmp_path / "blocks.json"
    bench.emit(full, str(out))
    cap = capsys.readouterr()
    lines = cap.out.strip().splitlines()
    assert len(lines) == 1
    assert len(lines[0]) <= bench.COMPACT_LIMIT < 2000
    line = json.loads(lines[0])
    for k in REQUIRED:
        assert k in line, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    assert "workload" in line["config"] and "model" not in line["config"]
    assert line["clock"]["mfma_streamed_tflops"] == 1712.0 and line["clock"]["kernel_over_streamed"] == 0.877
    assert line["vs_baseline"] is None and line["roofline"]["bound"] in ("hbm", "mfma")
    # nothing is lost: the full record is in the blocks file (and on stderr)
    assert json.load(open(out))["configs"].keys() == full["configs"].keys()
    assert json.loads(cap.err.strip().splitlines()[-1])["value"] == full["value"]


def test_compact_line_survives_missing_blocks():
    bench = _bench()
    text = bench.compact({"metric": "m", "value": 1.0, "unit": "frames/s", "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": 1.0,
                          "scaling": "weak", "cpu_baseline": {"error": "x" * 5000}, "roofline": {"kernel": "k" * 5000}})
    assert len(text) <= bench.COMPACT_LIMIT
    json.loads(text)


def test_compact_line_names_the_run_it_describes():
    """workload / dtype come from the record, not from literals: a run at another size is not labelled as configs[2] at its stated size"""
    bench = _bench()
    rec = {"metric": "frames/sec scored (MFCC+GMM)", "value": 1.0, "unit": "frames/s", "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": 1.0,
           "scaling": "weak", "dtype": "f32 (split fp16 ...)", "data": "synthetic",
           "config": {"frames_per_gpu": 10_000_000, "models": 201, "mixtures": 512, "dim": 39}}
    line = json.loads(bench.compact(rec))
    assert line["dtype"] == "f32" and line["config"]["workload"].startswith("configs[2]: ")
    assert "512-mix UBM + 200 MAP speakers, 10.0 M frames/GPU" in line["config"]["workload"]
    rec["config"]["frames_per_gpu"] = 200_000
    line = json.loads(bench.compact(rec))
    assert "non-default size" in line["config"]["workload"] and "0.2 M frames/GPU" in line["config"]["workload"]
    # strings far wider than anything the script writes are cut, the line stays parseable and short
    rec["metric"] = "m" * 4000
    rec["unit"] = "u" * 4000
    text = bench.compact(rec)
    assert len(text) <= bench.COMPACT_LIMIT
    json.loads(text)
