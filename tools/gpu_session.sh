#!/bin/bash
# One parameterised GPU session script (round 5: replaces the per-experiment gpu_r0*.sh of rounds 3-4).
#   usage (on the GPU box, through gpurun):  bash tools/gpu_session.sh <session> [outdir-name]
# Every session writes under gpurun_out/<outdir-name>/ and chains its steps with && -- a step that fails or times out ends the session.
set -o pipefail
session=${1:?session name}; out=gpurun_out/${2:-r05_$session}; mkdir -p $out
export MFGP_HW_QUEUES=${MFGP_HW_QUEUES:-2}
last_json() { python - "$1" <<'EOF'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
c = d.get("config", {})
print(sys.argv[1], "value", d["value"], "ms; n_gpus", d.get("n_gpus"), "ranks", c.get("ranks"), "rccl_ranks", c.get("rccl_ranks"),
      "frac", d.get("roofline", {}).get("frac"), "checksum", d.get("result_checksum"))
EOF
}
case $session in
deferred_k)
  # K^-1 as ONE deep-K stand-alone product per pass (MFGP_KINV_STREAM=0) against the streamed K = 128*MB chunks, in
  # the BATCHED regime (tools/batch_eval.py) and in the bench, alternating within this one call (box-to-box spread is 2-4 %)
  for rep in 1 2; do
    for ks in 1 0; do
      MFGP_KINV_STREAM=$ks BATCHES="3 4 6" timeout -k 10 300 python tools/batch_eval.py 4096 8192 > $out/batch_kinv${ks}_rep$rep.txt 2>&1 || { tail $out/batch_kinv${ks}_rep$rep.txt; exit 1; }
      echo "== MFGP_KINV_STREAM=$ks rep $rep"; grep '^N=' $out/batch_kinv${ks}_rep$rep.txt
    done
  done &&
  for rep in 1 2; do
    for ks in 1 0; do
      MFGP_KINV_STREAM=$ks timeout -k 10 300 python bench.py --no-cpu-baseline --steps 6 --warmup 1 > $out/bench_kinv${ks}_rep$rep.json 2> $out/bench_kinv${ks}_rep$rep.err || { tail $out/bench_kinv${ks}_rep$rep.err; exit 1; }
      echo "== bench MFGP_KINV_STREAM=$ks rep $rep"; last_json $out/bench_kinv${ks}_rep$rep.json
    done
  done
  ;;
rehearsal)
  # the N-rank bench at FULL size on the one-GPU box (every rank on GPU 0, RCCL over its socket transport: sharding.rehearsal_env).
  # 6 ranks is the most the box allows on its card (process guard); the 8-rank layout is rehearsed on the CPU (tests/test_bench_launcher.py)
  for n in ${RANKS:-6}; do
    timeout -k 10 900 python bench.py --gpus $n --single-device --no-cpu-baseline --steps 1 --warmup 1 > $out/bench_n$n.json 2> $out/bench_n$n.err || { tail -30 $out/bench_n$n.err; exit 1; }
    last_json $out/bench_n$n.json
  done
  ;;
pick)
  # selected tests, each group under its own timeout; PICK="file::test ..." (space separated pytest node ids / -k expressions are not split)
  for t in ${PICK:?}; do
    name=$(echo "$t" | tr '/:[]' '____')
    timeout -k 10 ${PICK_TIMEOUT:-600} python -m pytest "$t" -m gpu -x -q -s > $out/$name.log 2>&1; rc=$?
    tail -6 $out/$name.log
    [ $rc -eq 0 ] || exit $rc
  done
  ;;
tests)
  timeout -k 10 1100 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS:-} > $out/tests.log 2>&1; rc=$?; tail -15 $out/tests.log; exit $rc
  ;;
bench)
  timeout -k 10 600 python bench.py ${BENCH_ARGS:-} > $out/bench.json 2> $out/bench.err || { tail -30 $out/bench.err; exit 1; }
  last_json $out/bench.json
  ;;
*) echo "unknown session $session"; exit 2 ;;
esac
