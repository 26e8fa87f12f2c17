#!/bin/bash
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/mb; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mb -- python3 $R/tools/membound_x3.py 72 > /dev/null 2>&1
f=$(ls /tmp/mb/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY' | tee $R/gpurun_out/r6_membound_kernel_stats.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} min_us {float(r['MinNs'])/1e3:9.1f} max_us {float(r['MaxNs'])/1e3:9.1f}")
PY
