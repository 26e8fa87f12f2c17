#!/bin/bash
# does rocprofv3 survive a graph-mode bench step at this batch size?  usage: tools/prof_probe.sh <batch> [extra rocprofv3 flags]
R=$GRAFT_REPO_ROOT; b=$1; shift
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pp_$b
F="--steps 1 --warmup 1 --concurrent 1 --no-cpu-baseline --no-ref-layout --no-parity --no-fast-modes --no-roofline --batch $b"
rocprofv3 --kernel-trace "$@" --output-format csv -d /tmp/pp_$b -- python3 -X faulthandler $R/bench.py $F > /tmp/pp_$b.json 2> /tmp/pp_$b.err
echo "batch $b: rc $? json bytes $(stat -c %s /tmp/pp_$b.json) $(grep -c SIGSEGV /tmp/pp_$b.err) SIGSEGV lines"; cut -c1-120 /tmp/pp_$b.json
