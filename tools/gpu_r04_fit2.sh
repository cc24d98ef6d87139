#!/bin/bash
out=gpurun_out/r04j; mkdir -p $out
for args in "--lanes 1" "--lanes 2" "--lanes 2 --width 6" "--lanes 1 --width 7" "--lockstep 0"; do
python bench.py --no-cpu-baseline --no-power $args > $out/b.json 2>> $out/err.txt; python -c "
import json; d=json.load(open('$out/b.json')); print('$args:', d['value'], d['roofline']['achieved'], 'fit', d['serial_floor']['fit_ms_this_run'], d.get('lockstep_last_step'))"
done
