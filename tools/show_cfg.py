"""print one entry of the configs block of a bench.py JSON line read from stdin: python tools/show_cfg.py cfg0"""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps(d["configs"][sys.argv[1]]))
