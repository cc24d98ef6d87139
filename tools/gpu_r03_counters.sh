#!/bin/bash
# Round 3: settle the fp64 MFMA ceiling with counters.  rocprofv3 --pmc passes (with --kernel-trace
# only; the program directly after "--") over the bare probes and over one evaluation at N = 8192, plus the kernel trace
# the per-launch accounting (plan_flops / chain_account) is made from.  Output: gpurun_out/<tag>/.
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r03a}; out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
echo "== counters offered by this rocprofv3"
timeout -k 10 120 rocprofv3 -L > $out/counters_list.txt 2>&1
grep -c . $out/counters_list.txt
pick() { python3 $R/tools/pmc_pick.py $out/counters_list.txt "$@"; }
SQ_A=$(pick SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_F64 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE)
SQ_B=$(pick SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE)
SQ_C=$(pick SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MFMA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_SALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE)
echo "pass A: $SQ_A"; echo "pass B: $SQ_B"; echo "pass C: $SQ_C"
echo "== probes, un-profiled"
timeout -k 10 300 python3 $R/tools/probes/probes.py > $out/probes.txt 2>&1; tail -c 2500 $out/probes.txt
echo "== PMC pass A over the probes"
timeout -k 10 300 rocprofv3 --pmc $SQ_A --kernel-trace --output-format csv -d $out/pmcA_probes -- python3 $R/tools/probes/probes.py shapes > $out/pmcA_probes.log 2>&1
f=$(find $out/pmcA_probes -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $out/pmcA_probes.csv
f=$(find $out/pmcA_probes -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp $f $out/pmcA_probes_trace.csv
tail -3 $out/pmcA_probes.log
echo "== PMC passes A, B, C over one evaluation at N = 8192"
for p in A B C; do
  eval "cs=\$SQ_$p"
  timeout -k 10 300 rocprofv3 --pmc $cs --kernel-trace --output-format csv -d $out/pmc${p}_eval -- python3 $R/tools/time_eval.py 8192 > $out/pmc${p}_eval.log 2>&1
  f=$(find $out/pmc${p}_eval -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $out/pmc${p}_eval.csv
  f=$(find $out/pmc${p}_eval -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp $f $out/pmc${p}_eval_trace.csv
  tail -2 $out/pmc${p}_eval.log
done
(cd $R && python3 tools/mfma_counters.py $out > $out/mfma_counters.txt 2>&1; tail -60 $out/mfma_counters.txt)
echo "== kernel traces (un-countered) of one evaluation"
for n in 4096 8192; do
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/trace_$n -- python3 $R/tools/time_eval.py $n > $out/trace_$n.log 2>&1
  (cd $R && python3 tools/chain_account.py $out/trace_$n > $out/chain_account_$n.txt 2>&1; python3 tools/trace_summary.py $out/trace_$n > $out/trace_summary_$n.txt 2>&1; python3 tools/trace_timeline.py $out/trace_$n 0 400 > $out/timeline_$n.txt 2>&1; python3 tools/plan_flops.py $((n/128)) $out/timeline_$n.txt > $out/plan_flops_$n.txt 2>&1)
  cat $out/chain_account_$n.txt; tail -3 $out/plan_flops_$n.txt; tail -1 $out/trace_$n.log
done
find $out -name "*.csv" -size +20M -delete
find $out -type d -name "pmc*_*" -prune -exec rm -rf {} \; 2>/dev/null
find $out -type d -name "trace_*" -prune -exec rm -rf {} \; 2>/dev/null
echo "== bench (un-profiled)"
cd $R && timeout -k 10 500 python3 bench.py > $out/bench.json 2> $out/bench.err; tail -c 600 $out/bench.json
