#!/bin/bash
# re-entry check of the restored tree (round 5, session 4): the GPU suite, the same suite with guard bands behind every
# device buffer (FLUHIP_CANARY=1: an out-of-bounds device write aborts on release), and the bench line
export TMPDIR=/tmp; out=gpurun_out/r05s4; mkdir -p $out
python -m pytest tests -x -q -m gpu > $out/gputest.log 2>&1; echo "gpu suite rc=$?" | tee -a $out/summary.txt
tail -3 $out/gputest.log | tee -a $out/summary.txt
FLUHIP_CANARY=1 python -m pytest tests -x -q -m gpu > $out/gputest_canary.log 2>&1; echo "canary suite rc=$?" | tee -a $out/summary.txt
tail -3 $out/gputest_canary.log | tee -a $out/summary.txt
python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" | tee -a $out/summary.txt
cat $out/bench.json | head -c 1500
