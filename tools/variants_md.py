#!/usr/bin/env python3
"""profiles/rNN_variants.jsonl (tools/collect_variants.sh) -> markdown table on stdout:  python tools/variants_md.py profiles/r03_variants.jsonl r03"""
import json
import sys
path, tag = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "rNN")
rows = [json.loads(l) for l in open(path) if l.strip().startswith("{")]
print(f"# {tag} — the other BASELINE.json configurations through the same `bench.py` step (1x MI355X, batch 256 per GPU, 1 warm-up + 3 timed steps)\n")
print(f"Full JSON lines: `{path.split('/')[-1]}` (`tools/collect_variants.sh`), one box, back-to-back runs; a line whose bench.py failed or timed out reads `error`.\n")
print("| `bench.py` flags | workload | pairs/s | ms/step | peak HBM GiB | dominant kernel (share of a step) |")
print("|---|---|---|---|---|---|")
for d in rows:
    if "error" in d:
        print(f"| `{d['flags']}` | - | error rc={d['error']} | - | - | see {d.get('log', '')} |")
        continue
    r = d.get("roofline", {})
    dom = f"`{r.get('kernel', '?')}` {r.get('bound', '?')} {r.get('frac', 0):.3f} ({r.get('ms_per_step', 0):.1f} ms)" if r else "-"
    print(f"| `{d['flags']}` | {d['config']['workload'].split(':')[0]} / {d['dtype'][:40]} | {d['value']} | {d['ms_per_step']} | {d['config']['peak_hbm_gib']} | {dom} |")
