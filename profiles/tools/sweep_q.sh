for q in ${QS:-96 90 74 66 50 135 170}; do
  GNSSCORR_ACQ_Q_MB=$q python bench.py --no-shared --no-cpu > gpurun_out/sw_$q.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/sw_$q.json").read().strip().splitlines()[-1]); print($q, d["acquisition"]["ms_per_search"])
PY
done
