"""Prints workload, trajectories/s, ms per call and roofline.frac of the bench JSON lines under a directory."""
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + "/bench*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(d["config"]["workload"], round(d["value"]), round(d["ms_per_step"], 1), (d.get("roofline") or {}).get("frac"))
