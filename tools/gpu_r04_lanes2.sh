#!/bin/bash
out=gpurun_out/r04h; mkdir -p $out
for m in 2 3 4 6 8; do echo "MFGP_MACRO=$m"; MFGP_MACRO=$m python tools/lanes_batch.py 4096 "1:1 1:4 2:2 2:3 3:2"; done > $out/lanes_macro.txt 2>&1
for m in 3 4 6; do echo "MFGP_MACRO=$m"; MFGP_MACRO=$m python tools/lanes_batch.py 2048 "1:1 1:4 2:2 2:3 3:2"; done >> $out/lanes_macro.txt 2>&1
for m in 4 6 8; do echo "MFGP_MACRO=$m"; MFGP_MACRO=$m python tools/lanes_batch.py 8192 "1:1 2:2 2:3"; done >> $out/lanes_macro.txt 2>&1
cat $out/lanes_macro.txt
