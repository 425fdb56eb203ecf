#!/bin/bash
# Per-kernel time of the default bench command — run on the GPU box:  bash tools/collect_profile.sh gpurun_out/prof rNN
# Writes $1/${2}_bench_default_kernel_stats.csv and $1/${2}_bench_default.md (copy both to profiles/).
set -e
OUT=${1:-gpurun_out/prof}; TAG=${2:-r01}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 "$REPO/bench.py" --no-cpu-baseline > "$OUT/${TAG}_profiled_bench.log" 2>&1
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, sys
out, tag = sys.argv[1], sys.argv[2]
f = glob.glob("/tmp/kt/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
open(f"{out}/{tag}_bench_default_kernel_stats.csv", "w").write(open(f).read())
line = [l for l in open(f"{out}/{tag}_profiled_bench.log") if l.startswith("{")][-1]
b = json.loads(line)
tot = sum(float(r["TotalDurationNs"]) for r in rows)
steps = b["steps"] + b["warmup"]
md = [f"# {tag} — `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline` (1x MI355X, {b['config']['workload']}, {b['warmup']} warm-up + {b['steps']} timed steps)", "",
      f"Full per-kernel table: `{tag}_bench_default_kernel_stats.csv`; bench line of the profiled run: {b['value']:.1f} {b['unit']}, {b['ms_per_step']:.1f} ms/step.", "",
      "| kernel | calls | total ms | avg us | % of GPU time |", "|---|---|---|---|---|"]
for r in rows[:24]:
    md.append(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.1f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.1f} |")
md += ["", f"Sum of kernel time {tot/1e6:.0f} ms over {steps} steps = {tot/1e6/steps:.0f} ms/step vs {b['ms_per_step']:.0f} ms/step wall under the profiler.", ""]
for rl in [b["roofline"]] + b.get("roofline_other_kernels", []):
    fam = rl["kernel"]
    # (the weight-gradient family is two kernels: gemm_tn_kernel and the wide-tile gemm_tn_wide_kernel; bench.py times both as one)
    base = lambda n: {"gemm_tn_wide_kernel": "gemm_tn_kernel"}.get(n, n)
    sel = [r for r in rows if base(r["Name"].replace("void ", "").split("<")[0].split("(")[0].strip()) == fam]
    if not sel:
        continue
    n = sum(int(r["Calls"]) for r in sel); t = sum(float(r["TotalDurationNs"]) for r in sel)
    md.append(f"`{fam}` (all instantiations, all {steps} steps): {n} launches, {t/1e6:.1f} ms, average {t/1e3/n:.1f} us per launch "
              f"(bench.py's HIP-event figure for the {b['steps']} timed steps of the same run: {rl['launches']} launches, {rl['avg_launch_us']} us average).")
open(f"{out}/{tag}_bench_default.md", "w").write("\n".join(md) + "\n")
print(md[-1])
PY
