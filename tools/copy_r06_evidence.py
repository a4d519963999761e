"""Copy what tools/r06_profiles.sh left under gpurun_out/r06/ into profiles/ (tracked), under round-6 names, and place the
rocprofv3 sidecars where bench.py looks for them (profiles/bench_kernel_stats*.meta.json, csv paths rewritten)."""
import json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(ROOT, "gpurun_out", "r06"), os.path.join(ROOT, "profiles")
plain = {"bench.json": "r06_bench.json", "bench_driver_flags.json": "r06_bench_driver_flags.json", "bench_streams1.json": "r06_bench_streams1.json",
         "bench_model.json": "r06_bench_model.json", "bench_model_fold.json": "r06_bench_model_fold.json", "bench_tokens.json": "r06_bench_tokens.json",
         "bench_tokens_fold.json": "r06_bench_tokens_fold.json", "bench_strong_n1.json": "r06_bench_strong_n1.json",
         "bench_strong_b128_rank_size.json": "r06_bench_strong_b128_rank_size.json", "bench_exchange_world1.json": "r06_bench_exchange_world1.json",
         "r06_bench_kernel_stats.csv": "r06_bench_kernel_stats.csv", "r06_bench_model_kernel_stats.csv": "r06_bench_model_kernel_stats.csv",
         "r06_configs_kernel_stats.csv": "r06_configs_kernel_stats.csv", "r06_entropy_kernel_stats.csv": "r06_entropy_kernel_stats.csv",
         "gate_trace.txt": "r06_gate_trace.txt", "pmc_traffic.json": "r06_pmc_traffic.json", "roofline_table.json": "r06_roofline_table.json",
         "entropy_time.json": "r06_entropy_time.json", "bound_audit.json": "r06_bound_audit.json", "train_step.json": "r06_train_step.json",
         "entropy_pmc/summary.json": "r06_entropy_sq_counters.json"}
for src, dst in plain.items():
    s = os.path.join(O, src)
    if os.path.exists(s):
        shutil.copy(s, os.path.join(P, dst))
    else:
        print("missing", src, file=sys.stderr)
shutil.copy(os.path.join(P, "r06_pmc_traffic.json"), os.path.join(P, "pmc_traffic.json"))     # what bench.py reads
for m in ("bench_kernel_stats.meta.json", "bench_kernel_stats.model.meta.json"):
    s = os.path.join(O, m)
    if os.path.exists(s):
        d = json.load(open(s))
        d["csv"] = "profiles/" + os.path.basename(d["csv"])
        json.dump(d, open(os.path.join(P, m), "w"), indent=1)
        print(m, d["source_sha16"], d["dominant_kernel_avg_ms"])
for stale in ("bench_kernel_stats.model_fold.meta.json", "bench_kernel_stats.tokens_fold.meta.json"):     # round-5 sidecars of paths not re-traced
    p = os.path.join(P, stale)
    if os.path.exists(p):
        os.remove(p)
