import json,sys
for f in sys.argv[1:]:
    l=[x for x in open(f) if x.startswith('{"metric"')][-1]
    d=json.loads(l)
    print(f, "closed_loop ms", d['closed_loop']['ms'], "galileo data/pilot", d['closed_loop_galileo_e1']['ms_data_only'], d['closed_loop_galileo_e1']['ms_pilot'])
