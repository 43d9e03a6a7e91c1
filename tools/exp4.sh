python -m pytest tests/test_wide_gpu.py -m gpu -x -q 2>&1 | tail -3
python tools/wide_exp.py big 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['N'], d['kernel'], d['opts'], 'kernel_us', d['kernel_us'], 'frac', d['roofline_frac_kernel'])
"
python tools/wide_exp.py small 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['N'], d['kernel'], d['opts'], 'step_us', d['us_per_step'], 'kernel_us', d['kernel_us'])
"
