#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE rocprofv3 passes (mean per launch, per kernel).

HBM-side bytes per launch = FETCH_SIZE[KB] * 1024 / k + WRITE_SIZE[KB] * 1024, where k is CALIBRATED on a known byte
count in the kernels' own access pattern (tools/ubench/fetch_calib.hip, third argument = its rocprofv3 --pmc FETCH_SIZE
result): on gfx950 FETCH_SIZE reports 0.50 of the bytes of a coalesced 16 B/lane stream (the guide's "double it") but 0.91
of the bytes of the convolution kernels' input staging (288-B row segments starting 16 B before a 256-B boundary,
buffer_load_dwordx4) -- the blanket x2 of round 1 over-stated their reads by 1.8x.  WRITE_SIZE matched the exact output
bytes of the pooled conv kernels and is used as reported.  The file is stamped with a hash of the kernel sources it was
measured on; bench.py ignores it when they have changed.
usage: python tools/make_pmc_traffic.py <fetch.db> <write.db> <calib.db> > profiles/pmc_traffic.json
"""
import collections, json, re, sqlite3, sys


def per_kernel(path, counter):
    cur = sqlite3.connect(path).cursor()
    per = collections.defaultdict(float)
    for name, disp, c, v in cur.execute("select name, dispatch_id, counter_name, counter_value from pmc_events"):
        if c == counter:
            per[(name, disp)] += v
    agg = collections.defaultdict(list)
    for (name, _), v in per.items():
        agg[name].append(v)
    return {k: sum(v) / len(v) for k, v in agg.items()}


def lib_name(rocprof_name):
    m = re.search(r"DcxWino2pCfg<([^>]*)>", rocprof_name)
    if m:   # <TH, TW, EPI, G>
        f = [x.strip() for x in m.group(1).split(",")]
        epi = {"0": "DCX_EPI_BNRELU", "2": "DCX_EPI_HEAT"}[f[2] if len(f) > 2 else "0"]
        return "dcx_conv_wino2p_kernel<DcxWino2pCfg<" + f[0] + "," + f[1] + "," + epi + "," + (f[3] if len(f) > 3 else "1") + ">>"
    m = re.search(r"DcxWino2hCfg<([^>]*)>", rocprof_name)
    if m:   # <TH, TW, POOL, G>
        f = [x.strip() for x in m.group(1).split(",")]
        f[2] = "1" if f[2] == "true" else "0"
        return ("dcx_conv_wino2h_kernel<DcxWino2hCfg<" + ",".join(f[:3]) + "," + (f[3] if len(f) > 3 else "1")
                + (",1" if len(f) > 4 and f[4] == "1" else "") + ">>")
    m = re.search(r"DcxConvCfg<([^>]*)>", rocprof_name)
    if not m:
        return None
    f = [x.strip() for x in m.group(1).split(",")]
    f[7] = "1" if f[7] == "true" else "0"
    f[8] = {"0": "DCX_EPI_BNRELU", "1": "DCX_EPI_RAW", "2": "DCX_EPI_HEAT"}[f[8]]
    return "dcx_conv_mfma_kernel<DcxConvCfg<" + ",".join(f) + ">>"


KNOWN = {"calib_stream": 1258291200, "calib_tiles<32, 15>": 393154560, "calib_tiles<18, 0>": 1248473088}   # fetch_calib.hip


def main(fetch_db, write_db, calib_db):
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from deepcharuco_amd import _lib
    f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    cal = per_kernel(calib_db, "FETCH_SIZE")
    ratio = {}
    for name, v in cal.items():
        for key, known in KNOWN.items():
            if key in name:
                ratio[key] = v * 1024 / known
    k = ratio["calib_tiles<32, 15>"]
    out = {}
    for name in f:
        ln = lib_name(name)
        if ln is None or name not in w:
            continue
        out[ln] = int(f[name] * 1024 / k + w[name] * 1024)
    out["_detail"] = {lib_name(n): {"FETCH_SIZE_KB_reported": round(f[n], 1), "WRITE_SIZE_KB": round(w.get(n, 0), 1)}
                      for n in f if lib_name(n)}
    out["_calibration"] = {"FETCH_SIZE_reported_over_known_bytes": {kk: round(v, 4) for kk, v in ratio.items()},
                           "used": "calib_tiles<32, 15> (the conv kernels' staging pattern)"}
    out["_csrc_sha256"] = _lib.csrc_sha256()
    out["_note"] = ("bytes per launch = FETCH_SIZE / k + WRITE_SIZE (KB*1024), mean over the launches of separate --pmc passes; "
                    "FETCH_SIZE excludes what the 256 MB Infinity Cache serves, so a layer whose input was just written can "
                    "read less than its algorithmic input bytes")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])
