#!/bin/bash
# Where is the GPU idle inside a step?  rocprofv3 --kernel-trace of the default bench command (text tower on its side stream), then the union of
# the kernel intervals against the wall time of the traced window, the largest gaps and what ran on either side of them.
#   bash tools/trace_gaps.sh gpurun_out/gaps   (run on the GPU box; copy gaps.txt to profiles/)
set -e
OUT=${1:-gpurun_out/gaps}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kg
rocprofv3 --kernel-trace --output-format csv -d /tmp/kg -- python3 "$REPO/bench.py" --no-cpu-baseline --no-roofline --steps 4 --warmup 2 > "$OUT/bench.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob("/tmp/kg/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows)
# the last 4 steps: take the last 4/6 of the adamw launches as step boundaries
adam = [e for e in ev if e[2].startswith("adamw_kernel")]
per_step = len(adam) // 6
t0 = adam[-4 * per_step - 1][1] if len(adam) > 4 * per_step else ev[0][0]
t1 = adam[-1][1]
win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
busy, gaps, cur_s, cur_e, last = 0, [], None, None, None
for s, e, n, q in win:
    if cur_e is None:
        cur_s, cur_e, last = s, e, n
        continue
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, last, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    if e >= cur_e:
        last = n
busy += cur_e - cur_s
wall = t1 - t0
lines = [f"traced window: 4 steps, {wall / 4e6:.1f} ms per step; GPU busy (union of kernel intervals, all streams) {busy / 4e6:.1f} ms per step = {100 * busy / wall:.1f} %",
         f"idle {(wall - busy) / 4e6:.2f} ms per step in {len(gaps) // 4} gaps per step; sum of kernel durations {sum(e - s for s, e, _, _ in win) / 4e6:.1f} ms per step",
         "largest gaps (us, kernel before -> kernel after):"]
for g, a, b in sorted(gaps, reverse=True)[:25]:
    lines.append(f"  {g / 1e3:8.1f}  {a}  ->  {b}")
import collections
hist = collections.Counter()
for g, a, b in gaps:
    hist[min(int(g / 1e3) // 5 * 5, 100)] += g
lines.append("idle time by gap length (us bucket: ms per step): " + ", ".join(f"{k}+: {v / 4e6:.2f}" for k, v in sorted(hist.items())))
open(f"{out}/gaps.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
