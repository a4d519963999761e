#!/bin/bash
# bench.py at 1 / 2 / 3 / 4 stream slots on one box (twice), then strong scaling N=1, then the 2-rank launcher on one device
for rep in ; do
for s in 1 2 3 4; do
  python bench.py --no-cpu-baseline --no-parity --steps 400 --streams $s | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('weak streams $s', round(d['ms_per_step']*1000,1), 'us/step', round(d['value']), 'img/s', 'op', round(d['roofline']['whole_op_ms']*1000,1), 'p1', round(d['roofline']['kernel_ms']*1000,1))"
done; done
for s in ; do
  python bench.py --no-cpu-baseline --no-parity --steps 200 --scaling strong --streams $s | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('strong streams $s', round(d['ms_per_step']*1000,1), 'us/step', round(d['value']), 'img/s')"
done
python bench.py --steps 200 --streams 3 --cpu-seconds 2 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('parity', d['parity_checked'], d['code_mismatches'], d['parity'])"
DVQ_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 3 --batch 64 --streams 3 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('2 ranks', d['n_gpus'], d['parity_checked'], d['parity'])"
cat /sys/class/drm/card*/device/hwmon/hwmon*/power1_cap /sys/class/drm/card*/device/hwmon/hwmon*/power1_cap_max 2>/dev/null | head -4
/opt/rocm/bin/rocm-smi --showmaxpower 2>/dev/null | grep -i power
