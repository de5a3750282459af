#!/bin/bash
# SQ_INSTS_VALU / SALU / LDS of the half-shell kernel per diagnosis bit and state -> gpurun_out/n3_budget.txt
# (every k_nb_n3 dispatch of the process in dispatch order = the handle's n3_launches counter: n3_budget.py records each block's range)
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd $R
OUT=$R/gpurun_out/pmc_n3_budget; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 500 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT -- python3 scripts/n3_budget.py gw_200k 4 > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
python3 - <<PY | tee $R/gpurun_out/n3_budget.txt
import csv, glob, json, collections
seq = json.load(open("$R/gpurun_out/n3_budget_seq.json"))["seq"]
fs = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)
by = collections.defaultdict(dict)
for r in csv.DictReader(open(fs[0])):
    if "k_nb_n3<" in r["Kernel_Name"]:
        by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(by)
print(f"# {len(ids)} k_nb_n3 dispatches; per-dispatch means by (state, diag)")
for s in seq:
    blk = [by[i] for i in ids[s["first"]:s["first"] + s["launches"]]]
    mean = lambda k: sum(x.get(k, 0.0) for x in blk) / max(len(blk), 1)
    print(f"state {s['state']:4d} diag {s['diag']:2d}: {s['us']:7.1f} us  VALU {mean('SQ_INSTS_VALU'):.4e}  SALU {mean('SQ_INSTS_SALU'):.4e}  LDS {mean('SQ_INSTS_LDS'):.4e}  (n={len(blk)})")
PY
