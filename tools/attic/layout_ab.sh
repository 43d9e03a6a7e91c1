#!/bin/bash
# tools/layout_ab.sh -- same-box comparison of the column-major and the row-major entry points on the long-row classes (KKT, power law).
for m in synth:kkt:2000000:4 synth:powerlaw:1000000:6:120:400000; do for l in "" "--rm"; do python -m sextans_amd.sweep --rp 20 --n 8,16,32,64,128 $l $m 2>/dev/null | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(\"$m ${l:-cm}\", r[\"N\"], r[\"kernel\"], r[\"ms\"], r[\"roofline_frac\"])
"; done; done
