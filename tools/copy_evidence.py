#!/usr/bin/env python3
"""Copy what tools/r03_profiles.sh wrote under gpurun_out/r03 into profiles/ (tracked), fixing the csv paths of the rocprof
sidecars, and say whether the sidecars were made from the sources in the tree."""
import glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
O, P = os.path.join(ROOT, "gpurun_out", "r03"), os.path.join(ROOT, "profiles")
cp = lambda a, b: os.path.exists(os.path.join(O, a)) and shutil.copy(os.path.join(O, a), os.path.join(P, b))
for f in glob.glob(os.path.join(O, "bench*.json")):
    n = os.path.basename(f)
    if ".meta." not in n:
        shutil.copy(f, os.path.join(P, "r03_" + n))
for n in ("r03_bench_kernel_stats.csv", "r03_bench_model_kernel_stats.csv", "r03_bench_tokens_kernel_stats.csv", "r03_exchange_probe.json"):
    cp(n, n)
for a, b in (("pmc_traffic.json", "pmc_traffic.json"), ("pmc_traffic.json", "r03_pmc_traffic.json"), ("other_configs.json", "r03_other_configs.json"),
             ("bound_audit.json", "r03_bound_audit.json"), ("bound_audit_production.json", "r03_bound_audit_production_kernel.json"),
             ("stream_power_probe.json", "r03_stream_power_probe.json"), ("roofline_table.json", "r03_roofline_table.json"),
             ("pytest_gpu.log", "r03_pytest_gpu.log"), ("trace_bench_streams1.json", "r03_bench_under_rocprof_streams1.json"),
             ("train_step.jsonl", "r03_train_step.jsonl"), ("ema_probe.txt", "r03_ema_probe.txt"), ("conv_fused_probe.json", "r03_conv_fused_probe.json")):
    cp(a, b)
for src, dst, csv in (("bench_kernel_stats.meta.json", "bench_kernel_stats.meta.json", "profiles/r03_bench_kernel_stats.csv"),
                      ("bench_kernel_stats.model.meta.json", "bench_kernel_stats.model.meta.json", "profiles/r03_bench_model_kernel_stats.csv")):
    if os.path.exists(os.path.join(O, src)):
        m = json.load(open(os.path.join(O, src)))
        m["csv"] = csv
        json.dump(m, open(os.path.join(P, dst), "w"), indent=1)
        print(dst, m["source_sha16"], "tree", bench.source_sha16(), "OK" if m["source_sha16"] == bench.source_sha16() else "STALE", m["dominant_kernel_avg_ms"])
