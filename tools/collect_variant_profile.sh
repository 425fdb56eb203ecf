#!/bin/bash
# Per-kernel time of a bench variant — run on the GPU box:  bash tools/collect_variant_profile.sh gpurun_out/prof r01_vit --variant vit_b16
# Writes $1/${2}_kernel_stats.md (copy to profiles/).
set -e
OUT=$1; TAG=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktx
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktx -- python3 "$REPO/bench.py" "$@" --no-cpu-baseline --no-roofline --steps 1 --warmup 1 > "$OUT/${TAG}_profiled_bench.log" 2>&1
python3 - "$OUT" "$TAG" "$*" <<'PY'
import csv, glob, json, sys
out, tag, flags = sys.argv[1], sys.argv[2], sys.argv[3]
rows = list(csv.DictReader(open(glob.glob("/tmp/ktx/*/*kernel_stats.csv")[0])))
b = json.loads([l for l in open(f"{out}/{tag}_profiled_bench.log") if l.startswith("{")][-1])
md = [f"# {tag} — `rocprofv3 --kernel-trace --stats -- python3 bench.py {flags} --steps 1 --warmup 1` (1x MI355X, {b['config']['workload']})", "",
      f"Bench line of the profiled run: {b['value']:.1f} {b['unit']}, {b['ms_per_step']:.0f} ms/step (2 steps in the trace).", "",
      "| kernel | calls | total ms | avg us | % of GPU time |", "|---|---|---|---|---|"]
for r in rows[:16]:
    md.append(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.1f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.1f} |")
open(f"{out}/{tag}_kernel_stats.md", "w").write("\n".join(md) + "\n")
print("\n".join(md))
PY
