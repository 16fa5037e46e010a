#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_15
mkdir -p $O
cd $R
timeout 1700 python -m pytest tests -m gpu -q > $O/tests.log 2>&1
echo "tests rc=$?" > $O/status.txt; tail -2 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 2400 bash tools/profile_round.sh r02 > $O/profile_round.log 2>&1
tail -3 $O/profile_round.log | cut -c1-300
