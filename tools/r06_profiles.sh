#!/bin/bash
# Round-6 evidence on the GPU box.  Outputs under gpurun_out/r06/ (copy what is to be judged into profiles/).
# Usage: tools/r06_profiles.sh [bench] [rocprof] [pmc] [misc]   (default: all)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
WHAT=${*:-bench rocprof pmc misc}
TUNE=$R/dynamicvectorquantization_amd/csrc/libdvq_tuning.so
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has bench; then
  timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
  timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2>> $O/bench.err; cut -c1-200 $O/bench_driver_flags.json
  timeout 300 python bench.py --streams 1 --no-cpu-baseline --no-model-order --configs off > $O/bench_streams1.json 2>> $O/bench.err
  for p in model model_fold tokens tokens_fold; do
    timeout 300 python bench.py --path $p --no-cpu-baseline > $O/bench_$p.json 2>> $O/bench.err; cut -c1-160 $O/bench_$p.json
  done
  timeout 300 python bench.py --scaling strong --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_strong_n1.json 2>> $O/bench.err
  timeout 300 python bench.py --scaling strong --batch 128 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_strong_b128_rank_size.json 2>> $O/bench.err
  DVQ_BENCH_FORCE_EXCHANGE=1 timeout 300 python bench.py --no-cpu-baseline --no-model-order --configs off > $O/bench_exchange_world1.json 2>> $O/bench.err
fi
if has rocprof; then
  cd /tmp && export TMPDIR=/tmp
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_s1 -o t -- python3 $R/bench.py --streams 1 --no-cpu-baseline --no-parity --no-model-order --configs off > $O/trace_bench_streams1.json 2> $O/trace.err
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_model -o t -- python3 $R/bench.py --path model --streams 1 --no-cpu-baseline --no-parity > $O/trace_bench_model.json 2>> $O/trace.err
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_configs -o t -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/trace_bench_configs.json 2>> $O/trace.err
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_entropy -o t -- python3 $R/tools/entropy_time.py > /dev/null 2>> $O/trace.err
  cd $R
  python3 tools/rocprof_meta.py $O/trace_s1 $O/r06_bench_kernel_stats.csv routed weak "vq_assign_filter_kernel<256, 2, false, false>" | tee $O/rocprof_meta.log
  python3 tools/rocprof_meta.py $O/trace_model $O/r06_bench_model_kernel_stats.csv model weak "vq_assign_filter_kernel<256, 1, true, false>" bench_kernel_stats.model.meta.json | tee -a $O/rocprof_meta.log
  for p in configs entropy; do f=$(ls $O/trace_$p/*kernel_stats.csv $O/trace_$p/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/r06_${p}_kernel_stats.csv; done
  bash tools/gate_trace.sh gpurun_out/r06/gate_trace > $O/gate_trace.txt 2>&1
fi
if has pmc; then
  cd /tmp && export TMPDIR=/tmp
  DVQ_LIBRARY=$TUNE timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 $R/tools/pmc_workload.py > $O/pmc_fetch.log 2>&1
  DVQ_LIBRARY=$TUNE timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 $R/tools/pmc_workload.py > $O/pmc_write.log 2>&1
  cd $R
  python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1; tail -2 $O/pmc_traffic.log
  bash tools/entropy_pmc.sh gpurun_out/r06/entropy_pmc > $O/entropy_pmc.log 2>&1
fi
if has misc; then
  timeout 400 python tools/roofline_table.py > $O/roofline_table.json 2>> $O/bench.err
  timeout 300 python tools/entropy_time.py > $O/entropy_time.json 2>> $O/bench.err
  timeout 400 python tools/bound_audit.py 256 > $O/bound_audit.json 2>> $O/bench.err; tail -1 $O/bound_audit.json
  timeout 300 python tools/train_step_probe.py > $O/train_step.json 2>> $O/bench.err
fi
ls $O | head -60
