#!/bin/bash
# Round-3 evidence on the GPU box: GPU tests, bench lines per path, rocprofv3 kernel stats of the bench command (with the
# sidecar that ties them to the sources), the two PMC passes for HBM traffic, the exchange probe, the pass-1 A/B with
# in-kernel clock stamps (tuning build), per-config timings, bound audit, power probe.  Outputs under gpurun_out/r03/.
# Usage: tools/r03_profiles.sh [tests] [bench] [rocprof] [pmc] [xch] [ab] [misc]   (default: all)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
WHAT=${*:-tests bench rocprof pmc xch ab misc}
TUNE=$R/dynamicvectorquantization_amd/csrc/libdvq_tuning.so
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has tests; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
fi
if has bench; then
  timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-400 $O/bench.json
  timeout 300 python bench.py --streams 1 --no-cpu-baseline --no-model-order > $O/bench_streams1.json 2>> $O/bench.err; cut -c1-200 $O/bench_streams1.json
  timeout 300 python bench.py --path model --no-cpu-baseline > $O/bench_model.json 2>> $O/bench.err; cut -c1-200 $O/bench_model.json
  timeout 300 python bench.py --path model2 --no-cpu-baseline > $O/bench_model2.json 2>> $O/bench.err; cut -c1-200 $O/bench_model2.json
  timeout 300 python bench.py --path model --streams 1 --no-cpu-baseline --no-parity > $O/bench_model_streams1.json 2>> $O/bench.err; cut -c1-200 $O/bench_model_streams1.json
  timeout 300 python bench.py --path tokens --no-cpu-baseline > $O/bench_tokens.json 2>> $O/bench.err; cut -c1-200 $O/bench_tokens.json
  timeout 300 python bench.py --path tokens_model --no-cpu-baseline > $O/bench_tokens_model.json 2>> $O/bench.err; cut -c1-200 $O/bench_tokens_model.json
  timeout 300 python bench.py --path select --no-cpu-baseline > $O/bench_select_path.json 2>> $O/bench.err; cut -c1-200 $O/bench_select_path.json
  timeout 300 python bench.py --scaling strong --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_strong_n1.json 2>> $O/bench.err; cut -c1-200 $O/bench_strong_n1.json
  timeout 300 python bench.py --scaling strong --batch 128 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_strong_b128_rank_size.json 2>> $O/bench.err; cut -c1-200 $O/bench_strong_b128_rank_size.json
fi
if has rocprof; then
  cd /tmp && export TMPDIR=/tmp
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_s1 -o t -- python3 $R/bench.py --streams 1 --no-cpu-baseline --no-parity --no-model-order > $O/trace_bench_streams1.json 2> $O/trace.err
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_model -o t -- python3 $R/bench.py --path model --streams 1 --no-cpu-baseline --no-parity > $O/trace_bench_model.json 2>> $O/trace.err
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_tokens -o t -- python3 $R/bench.py --path tokens --streams 1 --no-cpu-baseline --no-parity > $O/trace_bench_tokens.json 2>> $O/trace.err
  cd $R
  python3 tools/rocprof_meta.py $O/trace_s1 $O/r03_bench_kernel_stats.csv routed weak "vq_assign_filter_kernel<256, 2, false, false>" | tee $O/rocprof_meta.log
  cp $O/bench_kernel_stats.meta.json $O/r03_bench_kernel_stats.meta.json 2>/dev/null
  python3 tools/rocprof_meta.py $O/trace_model $O/r03_bench_model_kernel_stats.csv model weak "vq_assign_filter_kernel<256, 1, true, false>" bench_kernel_stats.model.meta.json | tee -a $O/rocprof_meta.log
  for p in tokens; do f=$(ls $O/trace_$p/*kernel_stats.csv $O/trace_$p/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/r03_bench_${p}_kernel_stats.csv; done
fi
if has pmc; then
  cd /tmp && export TMPDIR=/tmp
  DVQ_LIBRARY=$TUNE timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 $R/tools/pmc_workload.py > $O/pmc_fetch.log 2>&1
  DVQ_LIBRARY=$TUNE timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 $R/tools/pmc_workload.py > $O/pmc_write.log 2>&1
  cd $R
  python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1; tail -2 $O/pmc_traffic.log
fi
if has xch; then
  # what the exchange (pack kernel + RCCL all-gather + unpack kernel) adds to a step, on the 1-rank RCCL group a 1-GPU box allows
  for S in 1 3; do
    timeout 300 python bench.py --streams $S --no-cpu-baseline --no-parity --no-model-order > $O/xch_plain_s$S.json 2>> $O/bench.err
    DVQ_BENCH_FORCE_EXCHANGE=1 timeout 300 python bench.py --streams $S --no-cpu-baseline --no-parity > $O/xch_rccl1_s$S.json 2>> $O/bench.err
  done
  cd /tmp && export TMPDIR=/tmp
  DVQ_BENCH_FORCE_EXCHANGE=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_xch -o t -- python3 $R/bench.py --streams 3 --no-cpu-baseline --no-parity > $O/trace_bench_xch.json 2>> $O/trace.err
  cd $R
  python3 tools/exchange_probe.py $O > $O/r03_exchange_probe.json; cut -c1-600 $O/r03_exchange_probe.json
fi
if has ab; then
  DVQ_LIBRARY=$TUNE timeout 500 python tools/p1_ab.py $O/r03_pass1_antiphase_ab.json > $O/p1_ab.log 2>&1; tail -2 $O/p1_ab.log
fi
if has misc; then
  timeout 400 python tools/bench_configs.py > $O/other_configs.json 2>> $O/bench.err
  timeout 400 python tools/bound_audit.py 256 > $O/bound_audit.json 2>> $O/bench.err; tail -1 $O/bound_audit.json
  DVQ_LIBRARY=$TUNE timeout 400 python tools/bound_audit.py 256 --production > $O/bound_audit_production.json 2>> $O/bench.err; tail -1 $O/bound_audit_production.json
  timeout 300 python tools/stream_power_probe.py > $O/stream_power_probe.json 2>> $O/bench.err
  timeout 300 python tools/roofline_table.py > $O/roofline_table.json 2>> $O/bench.err
  timeout 300 python tools/train_step_probe.py > $O/train_step.jsonl 2>> $O/bench.err; cat $O/train_step.jsonl
  timeout 300 python tools/ema_probe.py > $O/ema_probe.txt 2>> $O/bench.err
  timeout 300 python tools/conv_fused_probe.py > $O/conv_fused_probe.json 2>> $O/bench.err; cut -c1-300 $O/conv_fused_probe.json
fi
ls $O | head -80
