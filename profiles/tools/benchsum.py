import json,sys
for f in sys.argv[1:]:
    try:
        l=[x for x in open(f) if x.startswith('{"metric"')][-1]
    except Exception as e:
        print(f, "no line", e); continue
    d=json.loads(l)
    ex=d.get("extra",d)
    def g(k):
        v=ex.get(k) or {}
        return v.get("ms_per_step")
    print(f, "ms/step %.4f"%d["ms_per_step"], "roofline", d["roofline"]["frac"], "shared", g("shared_stream"), "i16", g("int16_input"), "gal", g("galileo_e1_5tap"), "hyb", g("hybrid_gps_galileo_beidou"))
