# usage (GPU box, repo root): bash tools/dp_rehearsal.sh <outdir under gpurun_out/> — two ranks sharing ONE GPU over gloo (host-staged exchange: a
# rehearsal of the world > 1 code path, NOT an xGMI measurement): bench.py's dp block (bytes, buckets, step time with and without the exchange)
# for f32 and bf16 buckets, per-rank batch 64 (weak) and 8 (strong: global batch 16 over 2 ranks).
set -o pipefail
out=gpurun_out/$1; mkdir -p $out
for cd in fp32 bf16; do
  NBCI_DIST_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 2 --steps 5 --warmup 2 --repeats 3 --global-batch 16 --comm-dtype $cd > $out/dp_gloo_2ranks_$cd.json 2> $out/dp_gloo_2ranks_$cd.err || { tail -5 $out/dp_gloo_2ranks_$cd.err; exit 1; }
  python - <<PY
import json
d = [json.loads(l) for l in open("$out/dp_gloo_2ranks_$cd.json") if l.startswith("{")][-1]   # (gloo prints its own lines to stdout)
print("$cd", "n_gpus", d["n_gpus"], "weak ms/step", d["ms_per_step"], "other", d["other_scaling"]["ms_per_step"], d["dp"])
PY
done
