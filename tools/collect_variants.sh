#!/bin/bash
# The other BASELINE.json configurations through the same bench.py step, back to back on one box:
#   bash tools/collect_variants.sh gpurun_out/variants.jsonl      (then copy to profiles/rNN_variants.jsonl)
OUT=${1:-gpurun_out/variants.jsonl}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
: > "$OUT"
run() { echo "# bench.py $*" >&2; timeout -k 10 280 python3 "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | tail -1 | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); d['flags'] = '$*'; print(json.dumps(d))" >> "$OUT"; }
run --seq-len 256
run --variant faithful
run --variant faithful --seq-len 256
run --checkpoint
run --variant vit_b16
run --variant vit_b16 --checkpoint
run --variant base
run --variant base --fp8
run --variant base --checkpoint
run --variant base --checkpoint --fp8
wc -l "$OUT"
