#!/bin/bash
# usage (GPU box): tools/kernel_ab.sh "<label>" <one_kernel args...>   (environment = the variant under test)
# rocprofv3 --kernel-trace of tools/one_kernel.py: median / min launch duration of every igemm / attention / reduce kernel of the run + their sum per call
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
label=$1; shift
rm -rf /tmp/kab
rocprofv3 --kernel-trace --output-format csv -d /tmp/kab -- python3 $R/tools/one_kernel.py "$@" > /dev/null 2>&1
python3 - "$label" "$*" <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(list)
for f in glob.glob("/tmp/kab/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if any(t in k for t in ("igemm", "attn", "gn_", "layernorm", "split_pair", "conv3x3")):
            d[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0.0
for k, v in d.items():
    v.sort()
    # one_kernel.py warms up with 10 calls: launches of the tuner's other candidates show up as extra kernels with few launches
    print(f"{sys.argv[1]:22s} {sys.argv[2]:28s} {k[:86]:86s} median {v[len(v)//2]:8.1f} us  min {v[0]:8.1f}  n={len(v)}", flush=True)
PY
