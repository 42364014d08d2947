#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r04f; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
( cd $R && timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
timeout 300 python $R/tools/probes/cli_overhead.py > $O/cli_overhead.json 2> $O/cli_overhead.err
timeout 600 python $R/bench.py --no_cpu_baseline --cli_steps 1100 > $O/bench.json 2> $O/bench.err
ls -la $O
