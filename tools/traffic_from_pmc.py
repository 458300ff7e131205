"""profiles/traffic_c2.json / traffic_c3.json from the committed PMC summaries (profiles/rNN_cX_pmc_summary.txt):
HBM bytes per launch (FETCH_SIZE KB x 1024 x 2 -- the gfx950 correction for wide coalesced reads, MI355X_MICROARCH.md
-- + WRITE_SIZE KB x 1024) and VALU wave-instructions per launch (SQ_INSTS_VALU).  bench.py reads both
(roofline.traffic, roofline.valu_bound).
    python tools/traffic_from_pmc.py [r02]"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"


def grab(path, sect):
    for blk in open(path).read().split("== "):
        if blk.startswith(sect + "\n"):
            return {m.group(1): float(m.group(2)) for m in
                    (re.match(r"\s+(\S+)\s+mean\s+(\S+)", l) for l in blk.split("\n")[1:]) if m}
    return {}


ALGO = 1382400000
for w in ("c2", "c3", "c3_adversarial", "c2_s16"):
    summ = os.path.join(ROOT, "profiles", "%s_%s_pmc_summary.txt" % (tag, w))
    if not os.path.exists(summ):
        continue
    out = os.path.join(ROOT, "profiles", "traffic_%s.json" % w)
    t = json.load(open(out)) if os.path.exists(out) else {}
    s, p = grab(summ, "scan"), (grab(summ, "tp") if not w.startswith("c2") else {})
    t["source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU (separate passes, counters only), profiles/%s" % os.path.basename(summ)
    if w.startswith("c2"):
        t["fetch_size_kb_raw"], t["write_size_kb"] = s["FETCH_SIZE"], s["WRITE_SIZE"]
    else:
        t["scan_fetch_size_kb_raw"], t["scan_write_size_kb"] = s["FETCH_SIZE"], s["WRITE_SIZE"]
        t["tp_fetch_size_kb_raw"], t["tp_write_size_kb"] = p["FETCH_SIZE"], p["WRITE_SIZE"]
    t["hbm_bytes_per_launch"] = int((s["FETCH_SIZE"] + p.get("FETCH_SIZE", 0.0)) * 2048 + (s["WRITE_SIZE"] + p.get("WRITE_SIZE", 0.0)) * 1024)
    t["algorithmic_bytes_per_launch"] = ALGO
    if w.endswith("_s16"):
        t["resident_bytes_per_launch"] = ALGO // 2  # (interleaved int16: what is actually there to read)
    t["valu_wave_instructions_per_launch"] = int(s["SQ_INSTS_VALU"] + p.get("SQ_INSTS_VALU", 0.0))
    json.dump(t, open(out, "w"), indent=1)
    print(w, t["hbm_bytes_per_launch"], "%.4f x algorithmic" % (t["hbm_bytes_per_launch"] / ALGO), t["valu_wave_instructions_per_launch"])
