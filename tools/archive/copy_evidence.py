#!/usr/bin/env python3
"""Copy what tools/r04_profiles.sh wrote under gpurun_out/r04 (argument: another round tag) into profiles/ (tracked), fixing the csv paths of the rocprof
sidecars, and say whether the sidecars were made from the sources in the tree."""
import glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
TAG = sys.argv[1] if len(sys.argv) > 1 else "r04"
O, P = os.path.join(ROOT, "gpurun_out", TAG), os.path.join(ROOT, "profiles")
cp = lambda a, b: os.path.exists(os.path.join(O, a)) and shutil.copy(os.path.join(O, a), os.path.join(P, b))
for f in glob.glob(os.path.join(O, "bench*.json")):
    n = os.path.basename(f)
    if ".meta." not in n:
        shutil.copy(f, os.path.join(P, TAG + "_" + n))
for n in ("bench_kernel_stats.csv", "bench_model_kernel_stats.csv", "bench_tokens_kernel_stats.csv", "bench_tokens_fold_kernel_stats.csv",
          "bench_model_fold_kernel_stats.csv", "exchange_probe.json"):
    cp(TAG + "_" + n, TAG + "_" + n)
for a, b in (("pmc_traffic.json", "pmc_traffic.json"), ("pmc_traffic.json", TAG + "_pmc_traffic.json"), ("other_configs.json", TAG + "_other_configs.json"),
             ("bound_audit.json", TAG + "_bound_audit.json"), ("bound_audit_production.json", TAG + "_bound_audit_production_kernel.json"),
             ("stream_power_probe.json", TAG + "_stream_power_probe.json"), ("roofline_table.json", TAG + "_roofline_table.json"),
             ("pytest_gpu.log", TAG + "_pytest_gpu.log"), ("trace_bench_streams1.json", TAG + "_bench_under_rocprof_streams1.json"),
             ("train_step.jsonl", TAG + "_train_step.jsonl"), ("ema_probe.txt", TAG + "_ema_probe.txt"), ("conv_fused_probe.json", TAG + "_conv_fused_probe.json")):
    cp(a, b)
for a, b in (("sq_counters.json", TAG + "_sq_counters.json"), ("bound_audit_fold.json", TAG + "_bound_audit_fold.json")):
    cp(a, b)
for src, dst, csv in (("bench_kernel_stats.meta.json", "bench_kernel_stats.meta.json", "profiles/%s_bench_kernel_stats.csv" % TAG),
                      ("bench_kernel_stats.model.meta.json", "bench_kernel_stats.model.meta.json", "profiles/%s_bench_model_kernel_stats.csv" % TAG),
                      ("bench_kernel_stats.tokens_fold.meta.json", "bench_kernel_stats.tokens_fold.meta.json", "profiles/%s_bench_tokens_fold_kernel_stats.csv" % TAG),
                      ("bench_kernel_stats.model_fold.meta.json", "bench_kernel_stats.model_fold.meta.json", "profiles/%s_bench_model_fold_kernel_stats.csv" % TAG)):
    if os.path.exists(os.path.join(O, src)):
        m = json.load(open(os.path.join(O, src)))
        m["csv"] = csv
        json.dump(m, open(os.path.join(P, dst), "w"), indent=1)
        print(dst, m["source_sha16"], "tree", bench.source_sha16(), "OK" if m["source_sha16"] == bench.source_sha16() else "STALE", m["dominant_kernel_avg_ms"])
