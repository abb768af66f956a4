#!/bin/bash
# tools/r5_visit11.sh: hybrid parity on the default (rows dealt out from 8 slices on), the whole GPU suite, smoke
t=${1:-r5_v11}; mkdir -p gpurun_out/$t
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/$t/tests.log 2>&1; tail -4 gpurun_out/$t/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
