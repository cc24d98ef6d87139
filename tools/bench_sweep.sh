# bench.py value (ms per fit+predict, exact evaluation budget) under planner / concurrency toggles, same box
run() { echo "$1: $(env $2 timeout -k 10 300 python bench.py --no-cpu-baseline $3 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["config"]["evals_issued_rank0_per_step"])')"; }
run "default          " "A=1" ""
run "MACRO=6          " "MFGP_MACRO=6" ""
run "MACRO=5          " "MFGP_MACRO=5" ""
run "XCD_ORDER=0      " "MFGP_XCD_ORDER=0" ""
run "MERGE_COLS=0     " "MFGP_MERGE_COLS=0" ""
run "SKINNY=0         " "MFGP_SKINNY=0" ""
