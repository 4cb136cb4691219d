# usage: bash tools/_ab.sh <outdir> "tag|spec flags|bench flags" ...   (A/B of plan-specialised step-kernel variants on one box)
O=gpurun_out/$1; shift
mkdir -p $O
for spec in "$@"; do
  tag=${spec%%|*}; rest=${spec#*|}; flags=${rest%%|*}; bflags=${rest#*|}
  MARO_AMD_SPEC_FLAGS="$flags" python bench.py --steps 200 --warmup 20 --no-cpu --secondary 0 --parity-envs 16 --no-episode $bflags > $O/$tag.json 2> $O/$tag.err
  python - <<P
import json
try:
    d=json.loads(open("$O/$tag.json").read().strip().splitlines()[-1])
    print("$tag", round(d["value"]/1e6,1), round(d["ms_per_step"]*1e3,2), "us parity", d.get("parity",{}).get("ok"), "spec", d["config"].get("specialized_kernels"), "kernel_us", round(d["roofline"].get("kernel_ms",0)*1e3,1))
except Exception as e:
    print("$tag FAILED", e); print(open("$O/$tag.err").read()[-1500:])
P
done
