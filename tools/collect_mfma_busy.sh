#!/bin/bash
# MFMA-busy cycles of every kernel of the default bench command (one rocprofv3 --pmc pass, no trace domains) — run on the GPU box:
#   bash tools/collect_mfma_busy.sh gpurun_out/mfma
# Writes $1/mfma_busy.json: per kernel family the summed SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE of its launches and
#   busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs)
# (the counter is summed over the 1024 SIMDs of the 256 CUs; GRBM_GUI_ACTIVE is the kernel's active cycles as summed over the 8
# XCDs by rocprofv3, hence the division by 8 below).  The algorithmic figure next to it is FLOPs / (1024 FLOP/cycle/SIMD).
set -e
OUT=${1:-gpurun_out/mfma}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_m
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_m -- python3 "$REPO/bench.py" --no-cpu-baseline --no-roofline > "$OUT/pass_mfma.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, collections, json, sys
out = sys.argv[1]
f = glob.glob("/tmp/pmc_m/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    fam = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].strip()
    fam = {"gemm_tn_wide_kernel": "gemm_tn_kernel"}.get(fam, fam)
    agg[fam][r["Counter_Name"]] += float(r["Counter_Value"])
    n[fam].add(r["Dispatch_Id"])
res = {}
for fam, d in agg.items():
    busy, act = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), d.get("GRBM_GUI_ACTIVE", 0.0)
    if busy <= 0 or act <= 0:
        continue
    res[fam] = {"launches": len(n[fam]), "SQ_VALU_MFMA_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE": act,
                "mfma_busy_fraction_xcd8": busy / (act / 8.0 * 1024.0), "mfma_busy_fraction_raw": busy / (act * 1024.0)}
res = dict(sorted(res.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"]))
json.dump({"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --no-cpu-baseline --no-roofline",
           "families": res}, open(out + "/mfma_busy.json", "w"), indent=1)
for fam, v in list(res.items())[:12]:
    print(f"{fam:34s} launches {v['launches']:5d}  busy/(active/8*1024) = {v['mfma_busy_fraction_xcd8']:.3f}   raw {v['mfma_busy_fraction_raw']:.4f}")
PY
