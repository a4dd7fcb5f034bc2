#!/bin/bash
# round-3 final: full GPU suite, smoke, default bench, profiles of the final binary
O=gpurun_out/r03_final; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu --tb=short > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -1 $O/smoke.log
S=$(date +%s); timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $? elapsed $(( $(date +%s) - S )) s"
bash tools/scripts/r03_numbers.sh 2>&1 | tail -12
bash tools/scripts/r03_prof.sh > $O/prof.log 2>&1; tail -3 $O/prof.log
bash tools/scripts/r03_run15.sh > $O/trainprof.log 2>&1; tail -2 $O/trainprof.log
