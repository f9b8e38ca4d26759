import sys,json
for l in sys.stdin:
    if not l.startswith("{"): continue
    r=json.loads(l); print(r.get("tag",""), r["dims"], r["family_of_instances"], r["B"], r["order"], "LW",r["latency_waves"], "fills",r["stage_fills"], "->", r["kernel_ms"], r["family"][0], "s" if r["staged"] else "-", "t" if r["tail"] else "-")
