#!/bin/bash
# SQ counters of the B = 512 video gradient: apply_q256 (option 1) vs apply_coeffs_x3_m256n128 (option 0)
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_aq
for o in 0 1; do
  KCCOT_OPTIONS="apply_q256=$o" timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_aq/o$o -- python3 tools/bench_apply.py 512 128 48 128 3 > gpurun_out/pmc_aq/o$o.log 2>&1
  python3 - $o <<'PY'
import csv, glob, sys, collections
o = sys.argv[1]
d = "gpurun_out/pmc_aq/o%s" % o
cc = glob.glob(d + "/*/*counter_collection.csv")[0]
kt = glob.glob(d + "/*/*kernel_trace.csv")[0]
dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(kt))}
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(cc)):
    name = r["Kernel_Name"].split("(")[0].replace("void kccot::", "")
    if "apply" not in name: continue
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        acc[name]["ns"] += dur[r["Dispatch_Id"]]; n[name] += 1
for k, m in sorted(acc.items()):
    cyc = m["GRBM_GUI_ACTIVE"] / 8
    print("apply_q256=%s %-28s launches %d  %.2f ms  clock %.2f GHz  mfma_busy %.3f  lds_active/cyc %.3f  lds_conflict/lds_active %.3f  wait_inst/wave %.2f  valu_active/wave %.2f" % (
        o, k[:28], n[k], m["ns"] / n[k] / 1e6, cyc / m["ns"], m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024) if cyc else 0,
        m["SQ_LDS_ACTIVE"] / (cyc * 256) if cyc else 0, m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_LDS_ACTIVE"], 1),
        m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"]))
PY
done
