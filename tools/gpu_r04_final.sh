#!/bin/bash
# round 4, second session: acceptance (suite + smoke + bench) and the records that depend on the host layer
set -o pipefail
out=gpurun_out/${1:-r04fin}; mkdir -p $out
bash tools/gpu_r04_full.sh $(basename $out) || exit 1
python tools/midsize_fit.py 64 128 256 512 1024 2048 4096 > $out/midsize_fit.txt 2>&1; cat $out/midsize_fit.txt | cut -c1-400
python tools/midsize_fit.py --evals 20 2048 4096 >> $out/midsize_fit.txt 2>&1; tail -3 $out/midsize_fit.txt | cut -c1-400
timeout -k 10 500 python3 tools/run_configs.py > $out/configs.txt 2>&1; cat $out/configs.txt
python tools/small_n_latency.py > $out/small_n.txt 2>&1; tail -12 $out/small_n.txt
