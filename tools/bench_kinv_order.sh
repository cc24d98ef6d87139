for cfg in "1 8" "2 16" "4 8" "1 8" "2 16" "4 8"; do
  set -- $cfg
  echo "BI=$1 BJ=$2: $(MFGP_KINV_BI=$1 MFGP_KINV_BJ=$2 timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["roofline"]["avg_launch_ms"], d["roofline"]["uncontended"]["achieved"])')"
done
