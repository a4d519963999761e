"""GPU (-m gpu): every path of bench.py at a small batch -- one JSON line with the contract's fields, `roofline`, the path's own
parity check green; the default path also reports the model order (the fused quant_conv op) beside the headline."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--spinup", "2", "--batch", "16",
                        "--no-cpu-baseline"] + list(extra), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("path", ["model", "model2", "tokens_model", "tokens", "tokens_fold", "model_fold"])
def test_bench_path_runs_and_checks_itself(path):
    d = _run("--path", path)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["config"]["path"] == path and d["steps"] == 6 and d["n_gpus"] == 1 and d["value"] > 0
    assert d["parity_checked"] is True, d["parity"]
    assert d["parity"]["code_mismatches"] == 0 and d["parity"]["slots_checked"] == 3
    assert d["config"]["repeats"] >= 1 and d["ms_per_step_min"] <= d["ms_per_step"] <= d["ms_per_step_max"]
    if path in ("model", "model2", "tokens_model", "tokens_fold", "model_fold"):
        assert d["parity"]["h_max_err_over_bound"] < 1.0 and d["parity"]["codes_match_rate_vs_fp64_conv"] > 0.995
    if path in ("model", "tokens_model"):
        assert d["parity"]["rerun_with_h_buf_mismatches"] == 0
        assert "1, true, false>" in d["roofline"]["kernel"]
    if path in ("tokens_fold", "model_fold"):
        assert "2, false, true>" in d["roofline"]["kernel"] and d["parity"]["tokens_resolved_with_a_conv"] > 0
        assert d["parity"].get("zq_mismatches", 0) == 0


def test_default_bench_reports_the_model_order_too():
    d = _run()
    assert d["config"]["path"] == "routed" and d["parity_checked"] is True
    m = d["model_order"]
    assert m["path"] == "model" and m["ms_per_step"] > 0 and m["parity_checked"] is True and m["parity"]["code_mismatches"] == 0
    for k, path in (("fold", "model_fold"), ("tokens_fold", "tokens_fold")):      # the loss-free forms with the conv folded into the codebook
        f = m[k]
        assert f["path"] == path and f["ms_per_step"] > 0 and f["parity_checked"] is True and f["parity"]["code_mismatches"] == 0
        assert f["parity"]["tokens_resolved_with_a_conv"] > 0
    assert "model_order" not in _run("--no-model-order", "--no-parity")
    # every BASELINE config in the reference's own op order, measured and checked in the same line
    c = d["configs"]
    assert c["all_parity_ok"] is True, c
    for key in ("cfg0", "cfg1", "cfg2", "cfg3", "cfg3_per_rank_of_8", "cfg4"):
        e = c[key]
        assert e["ms"] > 0 and e["code_mismatches"] == 0 and e["checked_images"] > 0, (key, e)
        assert not any(e.get("mismatches", {}).values()), (key, e)
    assert c["cfg2"]["entropy_max_abs_err"] <= 1e-5 and set(c["cfg2"]["stage_ms"]) == {"entropy_map", "assign_op", "pass1"}
    assert set(c["cfg1"]["stage_ms"]) == {"router_gate", "assign_op", "pass1"} and c["cfg1"]["loss_rel_err"] <= 1e-5
    assert c["cfg4"]["modes_bit_identical"] is True
    assert c["next_rows"]["permuter_forward_ms"] > 0 and c["next_rows"]["train_forward_backward_ms"] > 0, c["next_rows"]
    # the pre-flight record of a single-GPU run
    cfg = d["config"]
    assert cfg["world"] == 1 and cfg["distinct_devices"] == 1 and cfg["backend"] is None and len(cfg["ranks"]) == 1
    assert cfg["ranks"][0]["device_index"] == 0 and cfg["ranks"][0]["device_name"] and "no exchange" in cfg["parallelism"]
    assert "exchange_ms_per_step" not in d
