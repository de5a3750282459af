"""Summary of scripts/profile_kernels.sh: per kernel the rocprofv3 average duration, the algorithmic bytes per launch
(SURVEY.md 8d per-unit figures x units per launch), the achieved algorithmic GB/s and its fraction of the 6.3 TB/s the
part sustains, and the HBM traffic the counters saw (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950,
WRITE_SIZE as reported; KB -> bytes)."""
import collections, csv, glob, sys
out, wl, nb = sys.argv[1], sys.argv[2], int(sys.argv[3])
sys.path.insert(0, '.')
from multimm_amd import synthetic_system
from multimm_amd.system import _PRESETS
n0, _, l0, _ = _PRESETS[wl]
n = nb or n0
loops = max(1, int(round(l0 * n / n0)))
ALG = {  # bytes per launch
    "k_backbone": 25.0 * n, "k_loops": 64.0 * loops, "k_confine": 25.0 * n, "k_history": 228.0 * n,
    "k_pack": (28.0 + 168.0) * n, "k_cell_fill": 28.0 * n, "k_cell_order": 56.0 * n, "k_order_items": 56.0 * n, "k_nb_n3_unsort": 28.0 * n,
    "k_nb_n3": 32.0 * n, "k_nb_clusters_j": 32.0 * n,
}
stats = {}
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        key = r["Name"].split("(")[0].split("<")[0].split("::")[-1].split(" ")[-1]
        s = stats.setdefault(key, [0, 0.0])
        s[0] += int(r["Calls"]); s[1] += float(r["TotalDurationNs"])
pmc = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(out + "/pmc_%s/**/*counter_collection.csv" % C, recursive=True):
        for r in csv.DictReader(open(f)):
            key = r["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1].split(" ")[-1]
            acc[key][0] += float(r["Counter_Value"]); acc[key][1] += 1
    pmc[C] = {k: v[0] / max(v[1], 1) for k, v in acc.items()}
print(f"# {wl} n_beads={n} loops={loops}; rocprofv3 --kernel-trace --stats averages; peak = 6300 GB/s achievable (8000 spec)")
print(f"{'kernel':<22}{'calls':>7}{'avg us':>10}{'alg MB':>10}{'alg GB/s':>10}{'of 6.3TB/s':>11}{'HBM MB/launch (2*FETCH+WRITE)':>32}")
for k, (calls, tot) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
    if k not in ALG:
        continue
    us = tot / calls / 1e3
    gbs = ALG[k] / (us * 1e-6) / 1e9
    tr = None
    if k in pmc.get("FETCH_SIZE", {}) or k in pmc.get("WRITE_SIZE", {}):
        tr = (2.0 * pmc["FETCH_SIZE"].get(k, 0.0) + pmc["WRITE_SIZE"].get(k, 0.0)) * 1024.0 / 1e6
    print(f"{k:<22}{calls:>7}{us:>10.1f}{ALG[k] / 1e6:>10.2f}{gbs:>10.0f}{gbs / 6300.0:>10.1%}{'' if tr is None else f'{tr:>32.1f}'}")
