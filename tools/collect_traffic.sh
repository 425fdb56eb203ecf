#!/bin/bash
# HBM traffic of every kernel of the default bench command, from rocprofv3 PMC counters collected in SEPARATE passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with trace domains) — run on the GPU box:
#   bash tools/collect_traffic.sh gpurun_out/traffic
# Writes $1/traffic.json (then copy it to profiles/rNN_traffic.json).  FETCH_SIZE is doubled (gfx950 counts 128-B
# requests as 64 B on wide coalesced reads, MI355X_MICROARCH.md §HBM); WRITE_SIZE is exact for 16-B-per-lane stores.
set -e
OUT=${1:-gpurun_out/traffic}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_f /tmp/pmc_w
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -- python3 "$REPO/bench.py" --no-cpu-baseline --no-roofline > "$OUT/pass_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -- python3 "$REPO/bench.py" --no-cpu-baseline --no-roofline > "$OUT/pass_write.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, collections, json, sys
out = sys.argv[1]
res = {}
for tag, d in (("FETCH_SIZE", "/tmp/pmc_f"), ("WRITE_SIZE", "/tmp/pmc_w")):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == tag:
            k = r["Kernel_Name"].replace("void ", "")
            depth = 0
            for i, ch in enumerate(k):          # cut the argument list, not the template arguments
                depth += ch == "<"
                depth -= ch == ">"
                if ch == "(" and depth == 0:
                    k = k[:i]
                    break
            k = k.strip()
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])     # KiB
    res[tag] = agg
kernels = {}
for k, (n, kb) in res["FETCH_SIZE"].items():
    w = res["WRITE_SIZE"].get(k, [0, 0.0])
    kernels[k] = {"launches": n, "fetch_bytes_per_launch": 2.0 * kb * 1024 / n, "write_bytes_per_launch": w[1] * 1024 / max(w[0], 1)}
    # round 4 (tools/micro/fetch_calib.hip, profiles/r04_fetch_calib.txt): the x2 holds for requests of whole 128-byte lines; a kernel whose
    # reads are 64-byte segments (a depthwise halo row of one 32-channel slab) is counted at x1 - its x2 figure is twice its traffic.  Both kept.
    if k.startswith("dwconv7"):
        kernels[k]["requests_64B"] = True
        kernels[k]["fetch_bytes_per_launch_x1"] = kb * 1024 / n
fam = collections.defaultdict(lambda: [0, 0.0])        # kernel family = symbol name without template arguments
for k, v in kernels.items():
    v["hbm_bytes_per_launch"] = v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]
    f = k.split("<")[0].strip()
    fam[f][0] += v["launches"]
    fam[f][1] += v["launches"] * (v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"])
families = {f: {"launches": n, "hbm_bytes_per_launch": tot / n} for f, (n, tot) in fam.items() if n}
summary = {"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --no-cpu-baseline --no-roofline",
           "correction": "FETCH_SIZE x2 (gfx950 wide-read undercount), WRITE_SIZE x1, KiB -> bytes; kernels flagged requests_64B read 64-byte segments, "
                         "which FETCH_SIZE counts in full (profiles/r04_fetch_calib.txt): their traffic is fetch_bytes_per_launch_x1",
           "families": families, "kernels": kernels}
json.dump(summary, open(out + "/traffic.json", "w"), indent=1)
print(json.dumps({f: families[f] for f in ("gemm_nt_kernel", "gemm_tn_wide_kernel", "cnblock_mlp_fwd_kernel", "cnblock_bwdw_kernel") if f in families}))
PY
