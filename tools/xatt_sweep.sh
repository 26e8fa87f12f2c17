#!/bin/bash
# A/B of the text cross-attention kernel under rocprofv3 --kernel-trace (median launch duration): previous library vs this one, waves per workgroup x grid size
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() {   # label, env..., -- one_kernel args
  label=$1; shift
  rm -rf /tmp/xs; 
  rocprofv3 --kernel-trace --output-format csv -d /tmp/xs -- python3 $R/tools/one_kernel.py "$@" > /dev/null 2>&1
  python3 - "$label" "$*" <<'PY'
import csv, glob, sys
d = []
for f in glob.glob("/tmp/xs/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "xattn" in r["Kernel_Name"] or "attn_x3_kernel" in r["Kernel_Name"]:
            d.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
d.sort()
print(f"{sys.argv[1]:28s} {sys.argv[2]:24s} median {d[len(d)//2]:8.1f} us  min {d[0]:8.1f}  n={len(d)}", flush=True)
PY
}
export ONE_MODE=x3 ONE_B=${ONE_B:-24}
for shp in "4096 320 5 1" "4096 320 5 2" "1024 640 10 1" "1024 640 10 2" "256 1280 20 1" "64 1280 20 1"; do
  FREEFINE_HIP_LIB=$R/build/native/libfreefine_hip_prev.so run "previous" xattn $shp
  for cfg in "4 8" "8 8"; do
    set -- $cfg
    FFN_XATT_NW=$1 FFN_XATT_WPC=$2 run "prefetch nw=$1 wpc=$2" xattn $shp
  done
done
