#!/usr/bin/env python3
"""tools/sweep_table.py [profiles/r05_sweep.jsonl] -- the markdown table of DESIGN 9 (fraction of 8 TB/s on algorithmic bytes per STEP, per class and N)."""
import json, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r05_sweep.jsonl")
recs = [json.loads(l) for l in open(path) if l.startswith("{")]
Ns = [8, 16, 32, 64, 128, 256, 512]
rows = {}
for r in recs:
    key = (r["matrix"], (r.get("options", "") + (" row-major operands" if r.get("layout") == "rm" else "")).strip())
    rows.setdefault(key, {})[r["N"]] = r
short = lambda k: k.replace("spmm_csr_", "").replace("spmm_", "")
print("| Class | kernels (N = 8 / 16 / 128) | " + " | ".join(f"N={n}" if n == 8 else str(n) for n in Ns) + " |")
print("|---|---|" + "---|" * len(Ns))
fixtures = {}
for (m, o), d in rows.items():
    if not m.startswith("synth:") and m != "nasa4704":
        for n, r in d.items():
            fixtures.setdefault(n, []).append(r)
        continue
    ks = " / ".join(short(d[n]["kernel"]) if n in d else "—" for n in (8, 16, 128))
    cells = " | ".join(f"{d[n]['roofline_frac']:.3f}" if n in d else "—" for n in Ns)
    print(f"| `{m.replace('synth:', '')}`{' ' + o if o else ''} | {ks} | {cells} |")
fx = [r for v in fixtures.values() for r in v] + [r for (m, o), d in rows.items() if m == "nasa4704" for r in d.values()]
print(f"\n{len(recs)} records; .mtx files (--check): {len(set(r['matrix'] for r in fx))} files x {len(fixtures)} N, "
      f"bit_identical = {sorted(set(str(r.get('bit_identical')) for r in fx))}, passed = {sorted(set(str(r.get('passed')) for r in fx))}")
