"""Turns the per-kernel sums of one set of counter passes (scripts/pmc_passes.sh) into an entry of
profiles/r02_counters.json, the file bench.py reads counter-derived values from.
usage: python scripts/pmc_to_json.py <pmc outdir> <kernel substring> <entry key> <samples per dispatch | run:N> <json file> [note]
With run:N the counters of ALL matching dispatches of the run are summed and divided by the N samples the run rendered
(a kernel sequence that takes several dispatches per sample, such as the wavefront integrator's).
Derivations (per dispatch, then per sample):
  hbm bytes     (2 x FETCH_SIZE + WRITE_SIZE) x 1024: both counters are in KB; on gfx950 FETCH_SIZE tallies 128-byte read requests at
                64 bytes, so wide reads count at half their bytes (the raw value is kept next to it)
  valu_busy     4 x SQ_ACTIVE_INST_VALU / SIMD cycles, SIMD cycles = GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 x 1024 SIMDs
  lanes_active  SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)
  l1_hit        1 - TCP_TCC_READ_REQ / TCP_TOTAL_CACHE_ACCESSES;  l2_hit = TCC_HIT / (TCC_HIT + TCC_MISS)"""
import collections, csv, glob, json, os, re, sys

out, want, key, path = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[5]
whole_run = sys.argv[4].startswith("run:")
samples = float(sys.argv[4][4:]) if whole_run else float(sys.argv[4])
note = sys.argv[6] if len(sys.argv) > 6 else ""
tot = collections.defaultdict(float)
calls = collections.defaultdict(int)
names = set()
for f in glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if want not in r["Kernel_Name"]:
            continue
        names.add(r["Kernel_Name"])
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
        calls[r["Counter_Name"]] += 1
if not tot:
    raise SystemExit(f"no dispatch of a kernel matching '{want}' under {out}")
if whole_run:
    v = dict(tot)                                  # the run's totals stand for one "launch" of `samples` samples
    entry = {"kernel": want, "kernels_matched": sorted(names)[:8], "samples_per_dispatch": None, "samples_in_the_run": samples,
             "dispatches_per_pass": min(calls.values()), "note": note}
else:
    v = {c: tot[c] / calls[c] for c in tot}           # per dispatch
    entry = {"kernel": want, "kernels_matched": sorted(names)[:4], "samples_per_dispatch": samples, "dispatches_per_pass": min(calls.values()), "note": note}
if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
    entry["fetch_kb_raw_per_launch"] = v["FETCH_SIZE"]
    entry["write_kb_per_launch"] = v["WRITE_SIZE"]
    entry["hbm_bytes_per_launch"] = (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0
    entry["hbm_bytes_per_sample"] = entry["hbm_bytes_per_launch"] / samples
if "SQ_INSTS_VALU" in v:
    entry["valu_insts_per_launch"] = v["SQ_INSTS_VALU"]
    entry["valu_insts_per_sample"] = v["SQ_INSTS_VALU"] / samples
    entry["salu_per_valu"] = round(v.get("SQ_INSTS_SALU", 0.0) / v["SQ_INSTS_VALU"], 3)
if "SQ_ACTIVE_INST_VALU" in v and "GRBM_GUI_ACTIVE" in v:
    entry["valu_busy"] = round(4.0 * v["SQ_ACTIVE_INST_VALU"] / (v["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4)
if "SQ_THREAD_CYCLES_VALU" in v and v.get("SQ_ACTIVE_INST_VALU"):
    entry["lanes_active"] = round(v["SQ_THREAD_CYCLES_VALU"] / (64.0 * v["SQ_ACTIVE_INST_VALU"]), 4)
if v.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
    entry["l1_hit"] = round(1.0 - v.get("TCP_TCC_READ_REQ_sum", 0.0) / v["TCP_TOTAL_CACHE_ACCESSES_sum"], 4)
if v.get("TCC_HIT_sum") is not None and (v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0)) > 0:
    entry["l2_hit"] = round(v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]), 4)
if v.get("SQ_ACTIVE_INST_LDS"):
    entry["lds_bank_conflict_frac"] = round(v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_ACTIVE_INST_LDS"], 4)
data = json.load(open(path)) if os.path.exists(path) else {}
data[key] = entry
json.dump(data, open(path, "w"), indent=1, sort_keys=True)
print(key, json.dumps(entry))
