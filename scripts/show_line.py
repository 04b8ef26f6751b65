import json, sys
l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(l["value"], l["roofline"]["kernel"], l["roofline"]["frac"], l["roofline"]["frac_isolated"])
for k,v in (l["secondary"] or {}).items():
    print(k, {kk:vv for kk,vv in v.items() if kk in ("pages_per_s","error","f32s","bf16","us_per_page","us_per_page_grouped")} if isinstance(v,dict) else v)
print(l["cpu_baseline"]["value"] if l["cpu_baseline"] else None)
