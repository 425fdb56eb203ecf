#!/bin/bash
# PMC split of one NT GEMM shape (run on the GPU box):  bash tools/collect_gemm_pmc.sh gpurun_out/gemm_pmc 8192 8192 8192 none
# (PMC_PROG=tools/tn8_one.py bash tools/collect_gemm_pmc.sh gpurun_out/tn8_pmc 262144 512 2048: the same passes around another one-shape driver)
OUT=${1:-gpurun_out/gemm_pmc}; shift; REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_ANY SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"; do
  rm -rf /tmp/gp$i
  rocprofv3 --pmc $set --output-format csv -d /tmp/gp$i -- python3 "$REPO/${PMC_PROG:-tools/gemm_one.py}" "$@" > "$OUT/pass$i.log" 2>&1 || true
  cp /tmp/gp$i/*/*counter_collection.csv "$OUT/pass$i.csv" 2>/dev/null || true
  i=$((i+1))
done
python3 - "$OUT" "$@" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in sorted(glob.glob(out + "/pass*.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].strip()
        if "gemm" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
print("shape", sys.argv[2:])
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"    {c:28s} {v / n[k][c]:.4g}")
PY
