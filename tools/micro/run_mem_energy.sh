#!/bin/bash
# every case of tools/micro/mem_energy (built beforehand) for 3 s, rocm-smi's socket power sampled beside it; one JSON line per case
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
out=gpurun_out/mem_energy.jsonl
: > $out
idle=$(/opt/rocm/bin/rocm-smi --showpower 2>/dev/null | grep -oE "Power \(W\): [0-9.]+" | grep -oE "[0-9.]+$")
echo "{\"case\": \"idle\", \"watts\": [$idle]}" >> $out
for c in read_dword read_dword_plain read_x4 read_x4_plain read_dma read_dma_nt read_dma4 write_dword write_x4 copy_dword copy_x4; do
  timeout 60 tools/micro/mem_energy $c 3.0 > /tmp/me_$c.json 2>/dev/null &
  pid=$!
  w=""
  sleep 1.0
  while kill -0 $pid 2>/dev/null; do
    p=$(/opt/rocm/bin/rocm-smi --showpower 2>/dev/null | grep -oE "Power \(W\): [0-9.]+" | grep -oE "[0-9.]+$")
    w="$w$p,"
    sleep 0.3
  done
  wait $pid
  python3 - "$c" "$w" <<'PY' >> $out
import json,sys
c,w=sys.argv[1],sys.argv[2]
d=json.load(open('/tmp/me_%s.json'%c))
ws=[float(x) for x in w.split(',') if x]
ws=ws[:-1] if len(ws)>2 else ws          # the last sample may fall after the run
d['watts']=ws
print(json.dumps(d))
PY
done
cat $out
