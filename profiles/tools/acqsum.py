import json,sys
for f in sys.argv[1:]:
    try:
        l=[x for x in open(f) if x.startswith('{"metric"')][-1]
        d=json.loads(l); a=d["extra"]["acquisition"] if "extra" in d and "acquisition" in d["extra"] else d.get("acquisition")
        print(f, {k:a[k] for k in a if k in ("ms_per_search","value","dwells_per_s")}, a.get("roofline",{}).get("frac"))
    except Exception as e:
        print(f, "ERR", e)
