#!/bin/bash
# Regenerates the judged evidence on the GPU box (round 2): GPU tests, bench lines (weak N=1, strong N=1, select
# path), rocprofv3 kernel stats of the same bench command, the two PMC passes for HBM traffic, per-config timings,
# bound audit and the A/B tables.  Outputs under gpurun_out/final/.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
python bench.py --streams 1 --no-cpu-baseline > $O/bench_streams1.json 2>> $O/bench.err; cut -c1-200 $O/bench_streams1.json
python bench.py --path select --no-cpu-baseline > $O/bench_select_path.json 2>> $O/bench.err
python bench.py --scaling strong --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_strong_n1.json 2>> $O/bench.err; cut -c1-200 $O/bench_strong_n1.json
python bench.py --scaling strong --batch 128 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_strong_b128_rank_size.json 2>> $O/bench.err; cut -c1-200 $O/bench_strong_b128_rank_size.json
python bench.py --mode exact --steps 20 --no-cpu-baseline --path select > $O/bench_exact.json 2>> $O/bench.err
python tools/bench_configs.py > $O/other_configs.json 2>> $O/bench.err
python tools/ab_step.py > $O/ab_step.json 2>> $O/bench.err
python tools/bound_audit.py 256 > $O/bound_audit.json 2>> $O/bench.err; tail -1 $O/bound_audit.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/bench.py --no-cpu-baseline --no-parity > $O/trace_bench.json 2> $O/trace.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_s1 -o t -- python3 $R/bench.py --streams 1 --no-cpu-baseline --no-parity > $O/trace_bench_streams1.json 2>> $O/trace.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_strong -o t -- python3 $R/bench.py --scaling strong --batch 128 --steps 200 --streams 1 --no-cpu-baseline --no-parity > $O/trace_bench_strong_b128.json 2>> $O/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 $R/tools/pmc_workload.py > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 $R/tools/pmc_workload.py > $O/pmc_write.log 2>&1
cd $R
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1; tail -2 $O/pmc_traffic.log
ls $O $O/trace | head -40
