#!/bin/bash
out=gpurun_out/r04g; mkdir -p $out
for n in 1024 2048 4096 8192; do python tools/lanes_batch.py $n "1:1 1:2 1:3 1:4 1:6 2:1 2:2 2:3 3:1 3:2 4:1"; done > $out/lanes.txt 2>&1; cat $out/lanes.txt
