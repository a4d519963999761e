#!/bin/bash
# Round-5 evidence on the GPU box: GPU tests, bench lines per path, rocprofv3 kernel stats of the bench command (with the
# sidecar that ties them to the sources), the PMC passes (HBM traffic incl. the fetch split, SQ counters of the SEL = 2 / CONV /
# FOLD kernels), bound audits (filter, production kernel, fold), per-config timings.  Outputs under gpurun_out/r05/.
# Usage: tools/r05_profiles.sh [tests] [bench] [rocprof] [pmc] [sq] [misc]   (default: all)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
WHAT=${*:-tests bench rocprof pmc sq misc}
TUNE=$R/dynamicvectorquantization_amd/csrc/libdvq_tuning.so
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has tests; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
fi
if has bench; then
  timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-400 $O/bench.json
  timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2>> $O/bench.err; cut -c1-300 $O/bench_driver_flags.json
  timeout 300 python bench.py --streams 1 --no-cpu-baseline --no-model-order > $O/bench_streams1.json 2>> $O/bench.err; cut -c1-200 $O/bench_streams1.json
  for p in model model2 tokens tokens_model tokens_fold model_fold select; do
    timeout 300 python bench.py --path $p --no-cpu-baseline > $O/bench_$p.json 2>> $O/bench.err; cut -c1-200 $O/bench_$p.json
  done
  timeout 300 python bench.py --scaling strong --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_strong_n1.json 2>> $O/bench.err; cut -c1-200 $O/bench_strong_n1.json
  timeout 300 python bench.py --scaling strong --batch 128 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_strong_b128_rank_size.json 2>> $O/bench.err; cut -c1-200 $O/bench_strong_b128_rank_size.json
fi
if has rocprof; then
  cd /tmp && export TMPDIR=/tmp
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_s1 -o t -- python3 $R/bench.py --streams 1 --no-cpu-baseline --no-parity --no-model-order > $O/trace_bench_streams1.json 2> $O/trace.err
  for p in model tokens tokens_fold model_fold; do
    timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$p -o t -- python3 $R/bench.py --path $p --streams 1 --no-cpu-baseline --no-parity > $O/trace_bench_$p.json 2>> $O/trace.err
  done
  cd $R
  python3 tools/rocprof_meta.py $O/trace_s1 $O/r05_bench_kernel_stats.csv routed weak "vq_assign_filter_kernel<256, 2, false, false>" | tee $O/rocprof_meta.log
  python3 tools/rocprof_meta.py $O/trace_model $O/r05_bench_model_kernel_stats.csv model weak "vq_assign_filter_kernel<256, 1, true, false>" bench_kernel_stats.model.meta.json | tee -a $O/rocprof_meta.log
  python3 tools/rocprof_meta.py $O/trace_tokens_fold $O/r05_bench_tokens_fold_kernel_stats.csv tokens_fold weak "vq_assign_filter_kernel<256, 2, false, true>" bench_kernel_stats.tokens_fold.meta.json | tee -a $O/rocprof_meta.log
  python3 tools/rocprof_meta.py $O/trace_model_fold $O/r05_bench_model_fold_kernel_stats.csv model_fold weak "vq_assign_filter_kernel<256, 2, false, true>" bench_kernel_stats.model_fold.meta.json | tee -a $O/rocprof_meta.log
  for p in tokens; do f=$(ls $O/trace_$p/*kernel_stats.csv $O/trace_$p/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/r05_bench_${p}_kernel_stats.csv; done
fi
if has pmc; then
  cd /tmp && export TMPDIR=/tmp
  DVQ_LIBRARY=$TUNE timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 $R/tools/pmc_workload.py > $O/pmc_fetch.log 2>&1
  DVQ_LIBRARY=$TUNE timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 $R/tools/pmc_workload.py > $O/pmc_write.log 2>&1
  cd $R
  python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1; tail -2 $O/pmc_traffic.log
fi
if has sq; then
  bash tools/pmc_sq.sh gpurun_out/r05/pmc_sq > $O/pmc_sq.log 2>&1; tail -60 $O/pmc_sq.log
  cp $O/pmc_sq/summary.json $O/sq_counters.json 2>/dev/null
fi
if has misc; then
  timeout 400 python tools/bench_configs.py > $O/other_configs.json 2>> $O/bench.err
  timeout 400 python tools/bound_audit.py 256 > $O/bound_audit.json 2>> $O/bench.err; tail -1 $O/bound_audit.json
  timeout 400 python tools/bound_audit.py 128 --fold > $O/bound_audit_fold.json 2>> $O/bench.err; tail -1 $O/bound_audit_fold.json
  DVQ_LIBRARY=$TUNE timeout 400 python tools/bound_audit.py 256 --production > $O/bound_audit_production.json 2>> $O/bench.err; tail -1 $O/bound_audit_production.json
  timeout 300 python tools/roofline_table.py > $O/roofline_table.json 2>> $O/bench.err
fi
ls $O | head -80
