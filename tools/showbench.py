import json,sys
l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(l["value"], l["ms_per_step"], l["decode"]["ms_per_step"], l["roofline"]["frac"], l["decode"]["roofline"]["launch_us"])
print(json.dumps(l.get("neighbours"), indent=1))
