#!/bin/bash
# The other BASELINE.json configurations through the same bench.py step, back to back on one box:
#   bash tools/collect_variants.sh gpurun_out/variants.jsonl      (then copy to profiles/rNN_variants.jsonl)
# Every variant leaves a line: the bench line with its flags, or {"flags": ..., "error": rc} when bench.py failed / timed out
# (rc 124 / 137 = the 280 s limit); its stderr is kept in <out>.<n>.log.  A GPU step that was killed ends the collection.
# $2 = a (the C2-family and ViT lines), b (the ConvNeXt-B lines, C5 last) or all: two calls fit two 1200 s GPU slots.
OUT=${1:-gpurun_out/variants.jsonl}
PART=${2:-all}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
: > "$OUT"
N=0
run() {
    N=$((N + 1))
    local log="$OUT.$N.log" line rc
    echo "# bench.py $*" >&2
    # ENVV="NAME=value ..." in front of a run line sets knobs for that line only (recorded in its flags)
    env $ENVV timeout -k 10 280 python3 "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --detail-out "$OUT.$N.detail.json" "$@" > "$log.out" 2> "$log"
    rc=$?
    line=$(grep '^{' "$log.out" | tail -1)
    if [ "$rc" -ne 0 ] || [ -z "$line" ]; then
        python3 -c "import json, sys; print(json.dumps({'flags': sys.argv[1], 'error': int(sys.argv[2]), 'log': sys.argv[3]}))" "$*" "$rc" "$log" >> "$OUT"
        echo "# FAILED rc=$rc: bench.py $* (see $log)" >&2
        if [ "$rc" -eq 124 ] || [ "$rc" -eq 137 ]; then echo "# a GPU step was killed at its limit: stopping here" >&2; wc -l "$OUT"; exit 1; fi
        return
    fi
    printf '%s' "$line" | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); d['flags'] = (sys.argv[2] + ' ' if sys.argv[2] else '') + sys.argv[1]; print(json.dumps(d))" "$*" "$ENVV" >> "$OUT"
}
if [ "$PART" != b ]; then
run --seq-len 256
run --variant faithful
run --variant faithful --seq-len 256
run --checkpoint
run --variant vit_b16
run --variant vit_b16 --checkpoint
fi
if [ "$PART" != a ]; then
run --variant base
run --variant base --fp8
run --variant base --checkpoint
ENVV="MMG_FP8_BWD=0" run --variant base --checkpoint --fp8      # e4m3 forward GEMMs only (rounds 1 - 3)
run --variant base --checkpoint --fp8                           # BASELINE config C5: 8-bit GEMMs in both directions
fi
wc -l "$OUT"
