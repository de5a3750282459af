#!/bin/bash
# HBM traffic of the pair kernel per launch, per workload (separate rocprofv3 --pmc passes: FETCH_SIZE and WRITE_SIZE do not fit one
# pass; MI355X_MICROARCH.md, HBM section: FETCH_SIZE doubled on gfx950, WRITE_SIZE as reported).  -> gpurun_out/nb_traffic.json
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd $R
OUT=$R/gpurun_out/pmc_traffic; rm -rf $OUT; mkdir -p $OUT
run() { # tag, counter, bench args...
  local tag=$1 c=$2; shift 2
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $OUT/${tag}_$c -- python3 bench.py --cpu-seconds 0 --no-dd-leg "$@" > $OUT/${tag}_$c.log 2>&1 || { tail -3 $OUT/${tag}_$c.log; return 1; }
}
for C in FETCH_SIZE WRITE_SIZE; do
  run gw_200k $C --steps 20 --warmup 5 || exit 1
  run chr1_50k $C --workload chr1_50k || exit 1
  run gw_1m $C --workload gw_1m --steps 40 --warmup 10 || exit 1
done
python3 - <<PY
import csv, glob, json, collections
out = {}
for wl, cmd in (("gw_200k", "bench.py --steps 20 --warmup 5"), ("chr1_50k", "bench.py --workload chr1_50k"), ("gw_1m", "bench.py --workload gw_1m --steps 40 --warmup 10")):
    per = {}
    for C in ("FETCH_SIZE", "WRITE_SIZE"):
        fs = glob.glob("$OUT/%s_%s/**/*counter_collection.csv" % (wl, C), recursive=True)
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(fs[0])):
            k = r["Kernel_Name"]
            if "k_nb_" in k and "unsort" not in k and "fold" not in k:
                a = acc[k.split("(")[0].split("::")[-1].split("<")[0]]; a[0] += float(r["Counter_Value"]); a[1] += 1
        per[C] = {k: (v / n, n) for k, (v, n) in acc.items()}
    kern = max(per["FETCH_SIZE"], key=lambda k: per["FETCH_SIZE"][k][1])   # the pair kernel launched most often
    f_kb, n = per["FETCH_SIZE"][kern]; w_kb = per["WRITE_SIZE"][kern][0]
    out[wl] = {"kernel": kern, "command": cmd + " --cpu-seconds 0 --no-dd-leg", "launches_sampled": n,
               "FETCH_SIZE_KB_per_launch": f_kb, "WRITE_SIZE_KB_per_launch": w_kb, "fetch_correction": 2.0,
               "bytes_per_launch": int(round((2.0 * f_kb + w_kb) * 1024))}
    print(wl, out[wl])
out["_source"] = ("scripts/r5_traffic.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over the named commands; FETCH_SIZE doubled "
                  "(gfx950 reports half the bytes of wide streaming reads), WRITE_SIZE as reported (MI355X_MICROARCH.md, HBM section)")
json.dump(out, open("$R/gpurun_out/nb_traffic.json", "w"), indent=1)
PY
