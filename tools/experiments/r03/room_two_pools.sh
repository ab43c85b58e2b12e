O=gpurun_out/room2; mkdir -p $O
run() { tag=$1; shift; env "$@" python3 bench.py --cpu-seconds 0 --no-roofline-counters --workload room_4k_64spp_d16 --steps 2 --warmup 1 > $O/$tag.json 2>>$O/err.log || echo FAILED $tag; }
run prod DRT_NONE=1
run k1_p1024_t768 DRT_POOL_STACK_LDS=1 DRT_POOL_PATHS=1024 DRT_POOL_THREADS=768
run k1_p1024_t512 DRT_POOL_STACK_LDS=1 DRT_POOL_PATHS=1024 DRT_POOL_THREADS=512
run k1_auto DRT_POOL_STACK_LDS=1
run k2_p896_t768 DRT_POOL_STACK_LDS=2 DRT_POOL_PATHS=896 DRT_POOL_THREADS=768
python3 - $O <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get("roofline") or {}
        print("%-28s %10.1f Msamples/s  %9.4f ms/step  kernel alone %s ms  %s" % (os.path.basename(f), d["value"], d["ms_per_step"], r.get("kernel_ms"), r.get("kernel")))
    except Exception as e:
        print(f, "FAILED", e)
PY
