"""Merge per-config PMC summaries (tools/pmc_summary.py output) into profiles/*pmc_traffic.json, the file bench.py reads
`roofline.traffic` from.  usage: pmc_traffic.py OUT.json CFG=summary.json [CFG=summary.json ...]"""
import json
import sys

METHOD = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes over `bench.py --config CFG --steps 4 --warmup 1 "
          "--no-cpu-baseline` (MI355X_MICROARCH.md §HBM / rocprofv3 PMC slots); per-kernel averages over the dispatches that did work; counters are "
          "KiB; traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE reports half of the fetched bytes of a coalesced stream (guide) — "
          "checked on k_eval, whose coalesced inputs are 1.85 MB and whose FETCH_SIZE reads 0.96 MB; Infinity-Cache hits are counted, so this is "
          "fabric-side traffic of the XCD L2s, an upper bound of HBM bytes for these cache-resident windows")
out = {"_method": METHOD}
for arg in sys.argv[2:]:
    cfg, path = arg.split("=", 1)
    d = json.load(open(path))
    out[cfg] = {}
    for kernel, c in d.items():
        name = kernel.split("<")[0].split("::")[-1]
        f, w = c.get("FETCH_SIZE"), c.get("WRITE_SIZE")
        if f is None or w is None:
            continue
        out[cfg][name] = {"fetch_size_kib": f, "write_size_kib": w, "traffic_bytes": (2 * f + w) * 1024, "dispatches": c.get("active_dispatches")}
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
print("wrote", sys.argv[1], {k: len(v) for k, v in out.items() if k != "_method"})
