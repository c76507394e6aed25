set -o pipefail
R=$(pwd)
mkdir -p gpurun_out/ab
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --no-cpu-baseline --no-extra-points --no-roofline --repeats 1"
for v in 0 1; do
  export NBCI_ATTN_BIASGRAD=$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ab/trace$v -o t -- python3 $B --steps 20 --warmup 3 > $R/gpurun_out/ab/trace$v.log 2>&1 || exit 1
  python3 $R/tools/db_stats.py "$(find $R/gpurun_out/ab/trace$v -name '*.db' | head -1)" > $R/gpurun_out/ab/stats$v.csv
  rm -rf $R/gpurun_out/ab/trace$v
done
cd $R
python3 - <<'PY'
import csv
for v in (0,1):
    rows=list(csv.reader(open(f"gpurun_out/ab/stats{v}.csv")))[1:]
    tot=0
    for r in rows:
        if any(k in r[0] for k in ("attn_bwd","colsum","attn_fwd")): print(v, r[0][:60], r[1], r[3]); 
        if any(k in r[0] for k in ("attn_bwd","colsum")): tot+=int(r[2])
    print(v, "attn_bwd + colsum total us per step", tot/23/1e3)
PY
