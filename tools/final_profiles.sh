#!/bin/bash
# Regenerates the judged evidence on the GPU box: GPU tests, bench line, rocprofv3 kernel stats of the
# same bench command, and the two PMC passes for HBM traffic.  Outputs under gpurun_out/final/.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
python bench.py --mode exact --steps 20 --no-cpu-baseline > $O/bench_exact.json 2>> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/bench.py --no-cpu-baseline > $O/trace_bench.json 2> $O/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 $R/tools/pmc_workload.py > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 $R/tools/pmc_workload.py > $O/pmc_write.log 2>&1
cd $R
ls $O $O/trace | head -30
