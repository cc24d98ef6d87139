#!/bin/bash
set -o pipefail
out=gpurun_out/r04r; mkdir -p $out
python tools/power_watch.py --period 0.2 -- sleep 1 > $out/power_debug.txt 2>&1; tail -3 $out/power_debug.txt | cut -c1-400
ls /sys/class/drm/card*/device/hwmon/hwmon*/ 2>/dev/null | head -30 > $out/hwmon_ls.txt; head -5 $out/hwmon_ls.txt
timeout -k 10 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_models.py tests/test_gpu_c_abi.py -m gpu -q -x > $out/tests.log 2>&1; rc=$?; tail -5 $out/tests.log; [ $rc = 0 ] || exit $rc
timeout -k 10 300 python bench.py --no-cpu-baseline --power > $out/bench_default.json 2> $out/bench.err; python -c "
import json; d=json.loads([l for l in open('$out/bench_default.json') if l.startswith('{')][-1]); print('default bench:', d['value'], d['config']['restarts_run_as'][:60], d.get('power'))"
