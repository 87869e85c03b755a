#!/bin/bash
# SQ counters of the fused 3-D smoothing walks (forward: smooth_fused3, backward: smooth_fused3_adj) at a BASELINE shape
# usage: tools/pmc_smooth_fused3.sh "B H T W C"
export TMPDIR=/tmp
SHAPE=${1:-"512 128 48 128 3"}
mkdir -p gpurun_out/pmc_f3
for dir in fwd bwd; do
  script=tools/bench_smooth.py; [ $dir = bwd ] && script=tools/bench_smooth_bwd.py
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_f3/$dir -- python3 $script $SHAPE conv3d > gpurun_out/pmc_f3/$dir.log 2>&1
  python3 - $dir <<'PY'
import csv, glob, sys, collections
d = "gpurun_out/pmc_f3/%s" % sys.argv[1]
cc = glob.glob(d + "/*/*counter_collection.csv")[0]
kt = glob.glob(d + "/*/*kernel_trace.csv")[0]
dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(kt))}
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(cc)):
    name = r["Kernel_Name"].split("(")[0].replace("void kccot::", "")
    if "fused3" not in name: continue
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        acc[name]["ns"] += dur[r["Dispatch_Id"]]; n[name] += 1
for k, m in sorted(acc.items()):
    cyc = m["GRBM_GUI_ACTIVE"] / 8
    print("%s %-30s launches %d  %.3f ms  clock %.2f GHz  valu_active/wave_cycle %.3f  wait_any/wave_cycle %.3f  lds_active/cyc/CU %.3f  lds_conflict/lds_active %.3f  VALU insts/launch %.3g  wave_cycles/(cyc*1024) %.2f" % (
        sys.argv[1], k[:30], n[k], m["ns"] / n[k] / 1e6, cyc / m["ns"], m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"], m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"],
        m["SQ_LDS_ACTIVE"] / (cyc * 256) if cyc else 0, m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_LDS_ACTIVE"], 1), m["SQ_INSTS_VALU"] / n[k], m["SQ_WAVE_CYCLES"] / (cyc * 1024)))
PY
done
