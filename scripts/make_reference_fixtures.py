"""Reference-run fixtures for the parts of the hot path's surroundings that the reference CAN execute in the build
container (rows a2, a13, a14, f2, f3).  Runs HERE only (it reads /root/reference); what it writes
under tests/golden/ref_* is data -- inputs and the reference's outputs -- and is all that travels.

What is imported and how.  `import multimm` fails in this image (`multimm/__init__.py` pulls `bridge.py`, which imports
`openmm`: ModuleNotFoundError).  The three modules below are therefore loaded by file path under a private package name,
with EMPTY placeholder modules registered for the three import LINES that name packages the image lacks:

    utils.py:14                      import pyBigWig                               (used only by import_bw, :600)
    utils.py:17                      from openmm.unit import Quantity              (used only by save_args_to_txt, :737)
    initial_structure_tools.py:7     from hilbertcurve.hilbertcurve import HilbertCurve   (only generate_hilbert_curve, :158)

The placeholders have no behaviour: `Quantity` and `HilbertCurve` raise if anything touches them.  Functions CALLED here, none
of whose bodies reaches a placeholder:

    utils.import_mns_from_bedpe (utils.py:425-547)      utils.import_bed (utils.py:220-347)
    initial_structure_tools.build_init_mmcif (:292-358) for the deterministic curves circle / helix / spiral / knot
        (-> compute_init_struct :256-289 -> polymer_circle / helix_structure / spiral_structure / trefoil_knot_structure)
    initial_structure_tools.write_mmcif_chrom (:417-458)
    initial_structure_tools.compute_init_struct for those four curves (the raw float64 arrays)
    utils.get_coordinates_cif on the init files its own writer produced (the reader of model.py:1001,1083)
    utils.chrom_lengths_array / utils.chrom_strength (module-level tables, utils.py:67-137)

NOT pinned here (needs the absent packages): the Hilbert start (hilbertcurve 2.0.5) and every energy / force / minimizer
number (OpenMM 8.5.1).  Those rows stay "parity unpinned" ("Oracle").

Usage:  python scripts/make_reference_fixtures.py          (re-creates tests/golden/ref_*)
"""
from __future__ import annotations

import hashlib
import importlib.util
import json
import os
import shutil
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src/multimm"
REF_FIXTURE = "/root/reference/tests/fixtures/ENCFF045MJY_simple.bedpe"
OUT = os.path.join(ROOT, "tests", "golden")
PKG = "_multimm_reference_modules"


class _Untouchable:
    """Stands where a class of an absent package is NAMED by an import line; any use is an error."""

    def __init__(self, *a, **k):
        raise RuntimeError("placeholder for a package this image lacks was used: the fixture would not be a reference output")


def load_reference_modules():
    if not os.path.isdir(REF):
        raise SystemExit("this script runs in the build container only (/root/reference is absent here)")
    for name in ("pyBigWig", "openmm", "openmm.unit", "hilbertcurve", "hilbertcurve.hilbertcurve"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["openmm.unit"].Quantity = _Untouchable
    sys.modules["hilbertcurve.hilbertcurve"].HilbertCurve = _Untouchable
    import matplotlib
    matplotlib.use("Agg")
    pkg = types.ModuleType(PKG)
    pkg.__path__ = [REF]
    sys.modules[PKG] = pkg
    mods = {}
    for name in ("enums", "utils", "initial_structure_tools"):
        spec = importlib.util.spec_from_file_location(f"{PKG}.{name}", os.path.join(REF, name + ".py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[f"{PKG}.{name}"] = m
        spec.loader.exec_module(m)
        mods[name] = m
    return mods


# the synthetic .bed: whole-genome rows (coordinates inside the reference's chromosome table, every label class the parser knows + one it skips), written
# by this script, committed as tests/golden/ref_inputs/synthetic_subcompartments.bed
def synthetic_bed_rows(seed=11):
    rng = np.random.RandomState(seed)
    labels = ["A.1", "A.2", "B.1", "B.2", "A1", "A2", "B1", "B2", "A", "B", "Other", "A.1.x", "B.2.y"]    # not "NA": pandas reads it as NaN and utils.py:290 raises
    sys.path.insert(0, ROOT)
    from multimm_amd.system import CHROM_LENGTHS      # the reference's own table (utils.py:67-95), restated there
    sizes = [int(v) for v in CHROM_LENGTHS]
    names = [f"chr{i + 1}" for i in range(22)] + ["chrX", "chrY"]
    rows = []
    for name, size in zip(names, sizes):
        pos = int(rng.randint(10_000, 400_000))
        while pos < size - 1_500_000:
            length = int(rng.randint(150_000, 1_400_000))
            rows.append((name, pos, pos + length, labels[int(rng.randint(len(labels)))]))
            pos += length + int(rng.randint(0, 120_000))
    order = rng.permutation(len(rows))          # file order matters (later rows overwrite earlier ones)
    return [rows[i] for i in order]


BEDPE_CASES = {
    # name: kwargs of import_mns_from_bedpe (beyond the file); sizes are BASELINE.json's configs
    "gw_200k": dict(N_beads=200_000),
    "gw_1m": dict(N_beads=1_000_000),
    "chr1_50k": dict(N_beads=50_000, chrom="chr1", coords=[0, 248387328]),
    "chr6_region_5k": dict(N_beads=5_000, chrom="chr6", coords=[25_000_000, 60_000_000]),
    "gw_200k_shuffle_seed3": dict(N_beads=200_000, shuffle=True, seed=3),
    "gw_200k_shuffle_seed7": dict(N_beads=200_000, shuffle=True, seed=7),
    "gw_200k_down07_seed5": dict(N_beads=200_000, down_prob=0.7, seed=5),
    "gw_50k_threshold60": dict(N_beads=50_000, threshold=60),
    "gw_20k_mindist8": dict(N_beads=20_000, min_loop_dist=8),
}
BED_CASES = {
    "gw_200k": dict(N_beads=200_000),
    "gw_20k_shuffle_seed3": dict(N_beads=20_000, shuffle=True, seed=3),
    "chr1_50k": dict(N_beads=50_000, chrom="chr1", coords=[0, 248387328]),
    "chr6_region_5k": dict(N_beads=5_000, chrom="chr6", coords=[25_000_000, 60_000_000]),
    "gw_20k_flip_seed4": dict(N_beads=20_000, flip_prob=0.15, seed=4),
    "gw_20k_noise_seed9": dict(N_beads=20_000, noise_strength=0.6, seed=9),
    "gw_20k_noise_flip_shuffle_seed2": dict(N_beads=20_000, noise_strength=0.4, flip_prob=0.1, shuffle=True, seed=2),
}
CIF_CASES = [
    # (curve, n, chrom_ends)
    ("circle", 500, [0, 500]),
    ("helix", 257, [0, 257]),
    ("helix", 1000, [0, 400, 1000]),
    ("spiral", 640, [0, 100, 333, 640]),
    ("knot", 300, [0, 150, 300]),
]


def sha(path):
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def main():
    mods = load_reference_modules()
    utils, ist = mods["utils"], mods["initial_structure_tools"]
    inputs = os.path.join(OUT, "ref_inputs")
    os.makedirs(inputs, exist_ok=True)
    bedpe = os.path.join(inputs, "ENCFF045MJY_simple.bedpe")
    shutil.copyfile(REF_FIXTURE, bedpe)                 # a data file the reference's own tests hold
    os.chmod(bedpe, 0o644)
    bed = os.path.join(inputs, "synthetic_subcompartments.bed")
    with open(bed, "w") as f:
        for r in synthetic_bed_rows():
            f.write("\t".join(str(c) for c in r) + "\n")
    manifest = {"made_by": "scripts/make_reference_fixtures.py", "reference": "SFGLab/MultiMM v2.0.2 at /root/reference",
                "inputs": {"ENCFF045MJY_simple.bedpe": sha(bedpe), "synthetic_subcompartments.bed": sha(bed)},
                "bedpe": {}, "bed": {}, "cif": []}

    work = tempfile.mkdtemp(prefix="ref_fixtures_")
    os.makedirs(os.path.join(work, "metadata"))
    save = os.path.join(work, "")

    arrays = {}
    for name, kw in BEDPE_CASES.items():
        ms, ns, ds, ends, idxs = utils.import_mns_from_bedpe(bedpe_file=bedpe, path=save, **kw)
        arrays[f"bedpe/{name}/ms"] = np.asarray(ms, dtype=np.int64)
        arrays[f"bedpe/{name}/ns"] = np.asarray(ns, dtype=np.int64)
        arrays[f"bedpe/{name}/ds"] = np.asarray(ds, dtype=np.float64)
        arrays[f"bedpe/{name}/chr_ends"] = np.asarray(ends, dtype=np.int64)
        arrays[f"bedpe/{name}/chrom_idxs"] = np.asarray(idxs, dtype=np.int64)
        manifest["bedpe"][name] = {"kwargs": kw, "n_loops": int(len(ms))}
        print(f"bedpe {name}: {len(ms)} loops, chr_ends[:3] {list(ends[:3])}")
    for name, kw in BED_CASES.items():
        cs, ends, idxs = utils.import_bed(bed_file=bed, save_path=save, **kw)
        arrays[f"bed/{name}/Cs"] = np.asarray(cs, dtype=np.int8)
        arrays[f"bed/{name}/chr_ends"] = np.asarray(ends, dtype=np.int64)
        arrays[f"bed/{name}/chrom_idxs"] = np.asarray(idxs, dtype=np.int64)
        manifest["bed"][name] = {"kwargs": kw, "histogram": {int(v): int((cs == v).sum()) for v in (-2, -1, 0, 1, 2)}}
        print(f"bed {name}: histogram {manifest['bed'][name]['histogram']}")
    np.savez_compressed(os.path.join(OUT, "ref_ingest.npz"), **{k.replace("/", "__"): v for k, v in arrays.items()})

    cif_dir = os.path.join(OUT, "ref_cif")
    os.makedirs(cif_dir, exist_ok=True)
    curves = {}
    for curve, n, ends in CIF_CASES:
        tag = f"{curve}_{n}"
        ist.build_init_mmcif(n, np.array(ends), psf=False, path=save, curve=curve)
        dst = os.path.join(cif_dir, f"init_{tag}.cif")
        shutil.copyfile(save + "MultiMM_init.cif", dst)
        pts = np.asarray(ist.compute_init_struct(n, mode=curve), dtype=np.float64)
        curves[tag] = pts
        # per-chromosome writer on the same numbers, as save_chromosomes hands them over (model.py:899-905: 10 * nm)
        seg = pts[ends[0]:ends[1]]
        dstc = os.path.join(cif_dir, f"chrom_{tag}.cif")
        ist.write_mmcif_chrom(coords=seg, path=dstc)
        # the reference's reader (utils.py:168-205, what model.py:1001,1083 load structures with) on its own writer's file
        curves["read_" + tag] = np.asarray(utils.get_coordinates_cif(dst), dtype=np.float64)
        manifest["cif"].append({"curve": curve, "n": n, "chrom_ends": ends, "init": os.path.basename(dst),
                                "chrom": os.path.basename(dstc), "chrom_slice": [ends[0], ends[1]]})
        print(f"cif {tag}: {os.path.getsize(dst)} + {os.path.getsize(dstc)} bytes")
    # module-level tables of utils.py (67-137): the hg38 chromosome lengths and the central-force weights derived from them
    curves["chrom_lengths_array"] = np.asarray(utils.chrom_lengths_array)
    curves["chrom_strength"] = np.asarray(utils.chrom_strength, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "ref_curves.npz"), **curves)
    with open(os.path.join(OUT, "ref_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
        f.write("\n")
    shutil.rmtree(work)


if __name__ == "__main__":
    main()
