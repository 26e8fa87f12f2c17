#!/bin/bash
# GPU box: MFMA-busy / clock of attn_x3p_kernel vs attn_x3w_kernel on the harness shapes (one --pmc pass; durations from the same pass)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "16 4096 5 2" "16 4096 5 1" "16 1024 10 2"; do
  tag=$(echo $cfg | tr ' ' '_')
  rm -rf /tmp/pmc_$tag
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d /tmp/pmc_$tag -- $R/build/native/x3w_test time $cfg 3 > /dev/null 2>&1
  python3 - /tmp/pmc_$tag "$cfg" <<'PY'
import csv, glob, sys, collections
out, cfg = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, cs in agg.items():
    if "attn_x3" not in k: continue
    med = {c: sorted(v)[len(v) // 2] for c, v in cs.items()}
    d = sorted(dur[k])[len(dur[k]) // 2]
    busy = med["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (med["GRBM_GUI_ACTIVE"] / 8)
    clk = med["GRBM_GUI_ACTIVE"] / 8 / d / 1e3
    print(f"[{cfg}] {k[:40]:40s} {d:8.1f} us  MFMA busy {100 * busy:5.1f} %  clock {clk:.2f} GHz  " + "  ".join(f"{c}={med[c]:.3g}" for c in sorted(med)))
PY
done
