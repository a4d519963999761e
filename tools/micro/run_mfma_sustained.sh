#!/bin/bash
# runs tools/micro/mfma_sustained (built beforehand) and samples rocm-smi's socket power beside it
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 120 tools/micro/mfma_sustained 3.0 > gpurun_out/mfma_sustained.json 2> gpurun_out/mfma_sustained.err &
pid=$!
: > gpurun_out/mfma_sustained_power.txt
while kill -0 $pid 2>/dev/null; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr '\n' ' ' >> gpurun_out/mfma_sustained_power.txt
  echo >> gpurun_out/mfma_sustained_power.txt
  sleep 0.4
done
wait $pid
cat gpurun_out/mfma_sustained.json
cat gpurun_out/mfma_sustained_power.txt | sed 's/  */ /g' | head -60
