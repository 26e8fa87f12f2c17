#!/bin/bash
# usage: tools/pmc.sh <tag> <one_kernel args...>   -> gpurun_out/pmc_<tag>.txt  (three separate --pmc passes)
tag=$1; shift
cd /tmp; export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/p1 -- python3 $GRAFT_REPO_ROOT/tools/one_kernel.py "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $out/p2 -- python3 $GRAFT_REPO_ROOT/tools/one_kernel.py "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $out/p3 -- python3 $GRAFT_REPO_ROOT/tools/one_kernel.py "$@" > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 - "$out" > gpurun_out/pmc_$tag.txt <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "igemm" in k or "attn" in k or "gn_" in k or "layernorm" in k or "split_pair" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/p1/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, cs in agg.items():
    print(k[:90])
    if k in dur:
        d = sorted(dur[k]); print(f"   duration_us median {d[len(d)//2]:.1f}  n={len(d)}")
    for c, v in sorted(cs.items()):
        v = sorted(v); print(f"   {c:28s} median {v[len(v)//2]:.4g}")
PY
rm -rf $out
cat gpurun_out/pmc_$tag.txt
