import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
for k,v in d["roofline"]["per_kernel"].items(): print(k, round(v["ms_per_step"],3), v["launches_per_step"], round(v["achieved"]), round(v["frac"],3))
print(d.get("multi"))
