for o in "" "refine_sweeps=16" "refine_sweeps=32" "refine_rows=48" "refine_rows=56" "cluster_top=4"; do
  python tools/holdout_r05.py --rm --skip-rcm $o 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if not l.startswith('{'): continue
    r=json.loads(l)
    if r['N']==16: print('opts=[$o]', r['matrix'], r['kernel_us'], r['roofline_frac_kernel'], 'panel_rows', r.get('panel_rows_clustered'), r.get('panel_rows_natural'), 'blocks', r.get('panel_blocks_clustered'), 'plan_s', r.get('plan_build_s'))
"
done
