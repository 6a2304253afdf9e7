import sys,json
d=json.loads(sys.stdin.read()); r=d["roofline"]
print(d["value"], d["ms_per_step"], r["kernel"], r["frac"])
for k,v in r["families"].items(): print(f"{k:55s} {v}")
