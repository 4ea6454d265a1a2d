#!/bin/bash
# the round-end checks as the driver runs them: the -m gpu suite and smoke() on the final build
O=gpurun_out
mkdir -p $O
timeout -k 10 800 python3 -m pytest tests -x -q -m gpu > $O/r04_final_suite.txt 2>&1 || { tail -30 $O/r04_final_suite.txt; exit 1; }
tail -3 $O/r04_final_suite.txt
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r04_final_smoke.txt 2>&1 || { tail -20 $O/r04_final_smoke.txt; exit 1; }
tail -2 $O/r04_final_smoke.txt
