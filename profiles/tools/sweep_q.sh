# search time against the inter-pass buffer budget ($GNSSCORR_ACQ_Q_MB: satellites per batch); run on the GPU box
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for q in 35 70 100 140 200 270 540; do
  GNSSCORR_ACQ_Q_MB=$q python3 $R/profiles/tools/acq_time.py --label "q$q" 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['label'], 'cold %.4f steady %.4f found %s' % (j['cold_ms'], j['steady_ms'], j['found']))"
done
