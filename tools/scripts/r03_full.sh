#!/bin/bash
O=gpurun_out/r03_full; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x --tb=short > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -6 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -4 $O/smoke.log
