#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE rocprofv3 passes (mean per launch, per kernel).

HBM bytes per launch = 2 * FETCH_SIZE[KB] * 1024 + WRITE_SIZE[KB] * 1024.  The factor 2 on FETCH_SIZE is the gfx950
correction of /opt/skills/guides/MI355X_MICROARCH.md (section HBM): this rocprofv3 tallies the 128-B requests of a wide
coalesced stream at 64 B.  WRITE_SIZE matched the exact output bytes of the pooled conv kernels and is used as reported.
usage: python tools/make_pmc_traffic.py <fetch.db> <write.db> > profiles/pmc_traffic.json
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
    m = re.search(r"DcxWino2Cfg<([^>]*)>", rocprof_name)
    if m:   # <TH, TW, POOL, EPI>
        f = [x.strip() for x in m.group(1).split(",")]
        f[2] = "1" if f[2] == "true" else "0"
        epi = f[3] if len(f) > 3 else "0"
        return "dcx_conv_wino2_kernel<DcxWino2Cfg<" + ",".join(f[:3]) + (",DCX_EPI_HEAT" if epi == "2" else "") + ">>"
    m = re.search(r"DcxWinoCfg<([^>]*)>", rocprof_name)
    if m:   # <WM, WN, TH, TW, POOL, EPI> -> the name dcx_profile_kernel_name() reports
        f = [x.strip() for x in m.group(1).split(",")]
        f[4] = "1" if f[4] == "true" else "0"
        epi = f[5] if len(f) > 5 else "0"
        return "dcx_conv_wino_kernel<DcxWinoCfg<" + ",".join(f[:5]) + (",DCX_EPI_HEAT" if epi == "2" else "") + ">>"
    m = re.search(r"DcxConvCfg<([^>]*)>", rocprof_name)
    if not m:
        return None
    f = [x.strip() for x in m.group(1).split(",")]
    f[7] = "1" if f[7] == "true" else "0"
    f[8] = {"0": "DCX_EPI_BNRELU", "1": "DCX_EPI_RAW", "2": "DCX_EPI_HEAT"}[f[8]]
    return "dcx_conv_mfma_kernel<DcxConvCfg<" + ",".join(f) + ">>"


def main(fetch_db, write_db):
    f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    out = {}
    for name in f:
        ln = lib_name(name)
        if ln is None or name not in w:
            continue
        out[ln] = int(2 * f[name] * 1024 + w[name] * 1024)
    out["_detail"] = {lib_name(n): {"FETCH_SIZE_KB_reported": round(f[n], 1), "WRITE_SIZE_KB": round(w.get(n, 0), 1)}
                      for n in f if lib_name(n)}
    out["_note"] = "bytes per launch = 2*FETCH_SIZE + WRITE_SIZE (KB*1024), mean over the launches of separate --pmc passes"
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
