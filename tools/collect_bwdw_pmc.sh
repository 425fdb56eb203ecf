#!/bin/bash
# Kernel trace + PMC passes of the stage-1 CNBlock backward kernels:  bash tools/collect_bwdw_pmc.sh gpurun_out/bwdw_pmc
set -e
OUT=${1:-gpurun_out/bwdw_pmc}; REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bwkt
# Every pass records its return code in $OUT/passes.rc; the collection STOPS at the first pass that fails (a fault or an abort under the
# profiler must not be papered over, and a GPU step that failed is not run again) and the summary below says which passes completed.
: > "$OUT/passes.rc"
rc=0
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bwkt -- python3 "$REPO/tools/bwdw_pmc.py" > "$OUT/trace.log" 2>&1 || rc=$?
echo "trace rc=$rc" >> "$OUT/passes.rc"
if [ "$rc" -ne 0 ]; then echo "kernel-trace pass failed (rc=$rc): see $OUT/trace.log"; tail -5 "$OUT/trace.log"; exit "$rc"; fi
cp /tmp/bwkt/*/*kernel_stats.csv "$OUT/kernel_stats.csv"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VMEM SQ_INSTS_SALU" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM"; do
  rm -rf /tmp/pm$i
  rc=0
  BWDW_OLD=${BWDW_OLD:-1} rocprofv3 --pmc $set --output-format csv -d /tmp/pm$i -- python3 "$REPO/tools/bwdw_pmc.py" > "$OUT/pass$i.log" 2>&1 || rc=$?
  echo "pass$i rc=$rc ($set)" >> "$OUT/passes.rc"
  if [ "$rc" -ne 0 ]; then echo "PMC pass $i failed (rc=$rc): see $OUT/pass$i.log - stopping, later passes not run"; tail -5 "$OUT/pass$i.log"; break; fi
  cp /tmp/pm$i/*/*counter_collection.csv "$OUT/pass$i.csv"
  i=$((i+1))
done
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
print("== passes (return codes): " + "; ".join(l.strip() for l in open(out + "/passes.rc")))
print("== kernel trace (us per launch)")
try:
    for r in csv.DictReader(open(out + "/kernel_stats.csv")):
        if any(k in r["Name"] for k in ("cnblock", "gemm_tn")):
            print(f"  {r['Name'][:70]:70s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:10.1f} us")
except Exception as e:
    print("  (no kernel stats:", e, ")")
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for f in sorted(glob.glob(out + "/pass*.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].strip()
        if "cnblock" not in k and "gemm_tn" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"    {c:28s} {v/max(len(cnt[(k, c)]), 1):.4g}   (per launch, {len(cnt[(k, c)])} launches)")
PY
