"""headline fields of a bench.py JSON line read from stdin"""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d["roofline"]
print(json.dumps({"value": round(d["value"]), "ms_per_step": round(d["ms_per_step"], 4), "serial_ms": round(d["serial_ms_per_step"], 4),
                  "kernel_ms": round(r["kernel_ms"], 4), "bracketed": round(r["kernel_ms_bracketed"], 4), "frac": round(r["frac"], 4)}))
