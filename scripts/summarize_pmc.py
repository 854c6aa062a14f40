"""Turns the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as the MI355X guide prescribes) of
`bench.py` into per-kernel HBM traffic per launch.  gfx950 corrections (MI355X_MICROARCH.md, HBM section): both
counters are in KiB; FETCH_SIZE reports exactly half of the bytes of wide coalesced reads -> doubled; WRITE_SIZE is
exact for 16-byte-per-lane streaming stores.  Usage: summarize_pmc.py <fetch_dir> <write_dir> <out.json>"""
import collections, csv, glob, json, os, sys

def load(d, counter):
    out = collections.defaultdict(list)
    files = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    for f in files[-1:]:          # the newest pass only (gpurun merges every call's output into the same directory)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                out[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return out

fetch = load(sys.argv[1], "FETCH_SIZE"); write = load(sys.argv[2], "WRITE_SIZE")
res = {}
for k in sorted(set(fetch) | set(write)):
    name = k.replace("void ", "").replace("ftr::(anonymous namespace)::", "").split("(")[0]
    f = sum(fetch.get(k, [0])) / max(len(fetch.get(k, [0])), 1)
    w = sum(write.get(k, [0])) / max(len(write.get(k, [0])), 1)
    res[name] = dict(launches=len(fetch.get(k, [])), fetch_bytes=int(2 * f * 1024), write_bytes=int(w * 1024),
                     hbm_bytes=int((2 * f + w) * 1024))
json.dump(res, open(sys.argv[3], "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes"])[:25]:
    print(f"{k[:70]:70s} fetch {v['fetch_bytes']/1e6:9.1f} MB  write {v['write_bytes']/1e6:9.1f} MB")
