#!/bin/bash
# FETCH_SIZE of the depthwise kernels' access shapes against known byte counts (tools/micro/fetch_calib.hip) - run on the GPU box:
#   bash tools/collect_fetch_calib.sh gpurun_out/fetch_calib.txt
OUT=${1:-gpurun_out/fetch_calib.txt}; REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$(dirname "$OUT")"; OUT=$(cd "$(dirname "$OUT")" && pwd)/$(basename "$OUT")
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fc
rc=0
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/fc -- "$REPO/tools/micro/fetch_calib" > /tmp/fc.log 2>&1 || rc=$?
echo "rocprofv3 --pmc FETCH_SIZE -- tools/micro/fetch_calib   (rc=$rc)" > "$OUT"
grep "true bytes" /tmp/fc.log >> "$OUT"
[ "$rc" -ne 0 ] && { tail -5 /tmp/fc.log >> "$OUT"; cat "$OUT"; exit "$rc"; }
python3 - /tmp/fc /tmp/fc.log >> "$OUT" <<'PY'
import csv, glob, collections, re, sys
true = dict(zip(("k_wide", "k_seg64x16", "k_seg64x4", "k_all3x16"), map(int, re.findall(r"(\d{6,})", [l for l in open(sys.argv[2]) if "true bytes" in l][0]))))
agg = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0])):
    if r["Counter_Name"] == "FETCH_SIZE":
        agg[r["Kernel_Name"].split("(")[0].strip()].append(float(r["Counter_Value"]) * 1024)
print("kernel        true bytes      FETCH_SIZE (KiB -> bytes)   FETCH_SIZE / true   (second launch of two)")
for k, v in agg.items():
    if k in true:
        print(f"{k:12s} {true[k]:14d}  {v[-1]:16.0f}   {v[-1] / true[k]:.3f}")
PY
cat "$OUT"
