#!/bin/bash
# Per-kernel time of the default bench command — run on the GPU box:  bash tools/collect_profile.sh gpurun_out/prof rNN
# Writes $1/${2}_bench_default_kernel_stats.csv and $1/${2}_bench_default.md (copy both to profiles/).
# The traced command is `bench.py --no-cpu-baseline --one-stream`: with the text tower on its side stream (the default of the timed
# region) the trace stretches BERT's small kernels, which then run beside the image tower's full-grid ones - a kernel's OWN duration,
# which is what bench.py's roofline leg reports from its extra one-stream steps, needs one stream.  `$3` = extra bench flags.
set -e
OUT=${1:-gpurun_out/prof}; TAG=${2:-r01}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 "$REPO/bench.py" --no-cpu-baseline --one-stream --detail-out "$OUT/${TAG}_profiled_bench_detail.json" $3 > "$OUT/${TAG}_profiled_bench.log" 2>&1
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, sys
out, tag = sys.argv[1], sys.argv[2]
f = glob.glob("/tmp/kt/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
open(f"{out}/{tag}_bench_default_kernel_stats.csv", "w").write(open(f).read())
b = json.load(open(f"{out}/{tag}_profiled_bench_detail.json"))     # the full record; stdout carries only the compact contract line
tot = sum(float(r["TotalDurationNs"]) for r in rows)
steps = b["steps"] + b["warmup"] + (b["roofline_method"]["profiled_steps"] + 1 if "roofline_method" in b else 0)
md = [f"# {tag} — `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --one-stream` (1x MI355X, {b['config']['workload']}, {b['warmup']} warm-up + {b['steps']} timed steps)", "",
      f"Full per-kernel table: `{tag}_bench_default_kernel_stats.csv`; bench line of the profiled run: {b['value']:.1f} {b['unit']}, {b['ms_per_step']:.1f} ms/step.", "",
      "| kernel | calls | total ms | avg us | % of GPU time |", "|---|---|---|---|---|"]
for r in rows[:24]:
    md.append(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.1f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.1f} |")
md += ["", f"Sum of kernel time {tot/1e6:.0f} ms over {steps} steps = {tot/1e6/steps:.0f} ms/step vs {b['ms_per_step']:.0f} ms/step wall under the profiler.", ""]
def norm(name):            # "void gemm_nt_kernel<256, 256, 64, 4, 2, 0>(GemmNT)" -> "gemm_nt_kernel<256, 256, 64, 4, 2, 0>"
    name = name.replace("void ", "")
    depth, cut = 0, len(name)
    for i, ch in enumerate(name):
        depth += ch == "<"
        depth -= ch == ">"
        if ch == "(" and depth == 0:
            cut = i
            break
    return name[:cut].strip()
md += ["## bench.py's HIP-event figures (its extra one-stream steps) beside this trace, kernel by kernel", "",
       "| kernel (bench line) | bench launches / step | bench avg us | rocprof avg us | bench / rocprof |", "|---|---|---|---|---|"]
worst = 0.0
for rl in [b["roofline"]] + b.get("roofline_other_kernels", []):
    name = rl["kernel"]
    if "*" in name:            # one entry point, several launches (cnblock_bwdw: two kernels per call): sum of their averages
        pre = name.split("<")[0]
        sel = [r for r in rows if norm(r["Name"]).startswith(pre + "<") and "pack" not in r["Name"]]
        avg = sum(float(r["TotalDurationNs"]) / int(r["Calls"]) for r in sel) / 1e3 if sel else 0.0
    else:
        sel = [r for r in rows if norm(r["Name"]) == name] or [r for r in rows if norm(r["Name"]).split("<")[0] == name]
        n = sum(int(r["Calls"]) for r in sel)
        avg = sum(float(r["TotalDurationNs"]) for r in sel) / 1e3 / n if n else 0.0
    if not avg:
        continue
    ratio = rl["avg_launch_us"] / avg
    if rl["ms_per_step"] >= 1.0:
        worst = max(worst, abs(ratio - 1.0))
    md.append(f"| `{name}` | {rl['launches'] / b['roofline_method']['profiled_steps']:.0f} | {rl['avg_launch_us']:.1f} | {avg:.1f} | {ratio:.3f} |")
md += ["", f"Largest deviation among kernels with >= 1 ms per step: {100 * worst:.1f} %.  Sum of the bench line's instrumented kernels: "
       f"{b['roofline_method']['instrumented_kernel_ms_per_step']} ms per step (one stream) against {b['ms_per_step']} ms per step of the timed region.", ""]
open(f"{out}/{tag}_bench_default.md", "w").write("\n".join(md) + "\n")
print(md[-1])
PY
