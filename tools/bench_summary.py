import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "NO JSON", e); continue
    r = d["roofline"]
    print(f"{f}: {d['value']} fps  {d['ms_per_step']} ms/step  dominant {r['achieved']} TF  all-conv {r['all_conv_kernels']['achieved']} TF")
    for k, v in r["per_kernel"].items():
        print(f"    {k[31:-2]:42s} {v['ms_per_step']:8.4f} ms  {v['algorithmic_tflops']:7.2f} TF alg  frac {v['frac']:.3f}  clk {v.get('clock_ghz')}")
