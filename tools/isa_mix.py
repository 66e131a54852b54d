"""Kernel-tuning aid: static instruction mix per basic block of the conv kernels (hipcc -S of dcx_conv_mfma.hip).
usage: python tools/isa_mix.py [substring of the mangled kernel name ...]"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(tempfile.gettempdir(), "dcx_conv.s")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-mllvm", "-pragma-unroll-threshold=200000", "-S",
                os.path.join(ROOT, "deepcharuco_amd/csrc/dcx_conv_mfma.hip"), "-o", out], check=True, capture_output=True)
s = open(out).read()
pats = sys.argv[1:] or ["wino"]
for m in re.finditer(r"^(_Z2[0-9]dcx_conv_\w+):[^\n]*\n", s, re.M):
    sym = m.group(1)
    if not any(p in sym for p in pats):
        continue
    b = s[m.end():s.index(".Lfunc_end", m.end())]
    blocks = re.split(r"\n(\.LBB[0-9_]+):", "\n.LBBentry:" + b)
    print(sym)
    for k in range(1, len(blocks), 2):
        ins = [l.strip().split()[0] for l in blocks[k + 1].split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        c = collections.Counter()
        for x in ins:
            key = ("mfma" if x.startswith("v_mfma") else "acc_mov" if x.startswith("v_accvgpr") else "valu" if x.startswith("v_")
                   else "wait" if x.startswith("s_waitcnt") else "nop" if x.startswith("s_nop") else "barrier" if x.startswith("s_barrier")
                   else "salu" if x.startswith("s_") else "lds" if x.startswith("ds_") else "vmem" if x.startswith(("buffer_", "global_")) else "other")
            c[key] += 1
        if len(ins) > 30:
            print("   %-12s %5d  %s" % (blocks[k], len(ins), dict(c)))
