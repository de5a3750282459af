"""Reference-pinned, bit-exact: the ingest parsers, the mmCIF writers (a13 / f3) and the
deterministic start curves (a2, the non-Hilbert part) against OUTPUTS OF THE REFERENCE ITSELF.

tests/golden/ref_* were produced by scripts/make_reference_fixtures.py, which runs the reference's own
`import_mns_from_bedpe`, `import_bed` (utils.py:425-547, 220-347), `build_init_mmcif`, `write_mmcif_chrom`,
`compute_init_struct` (initial_structure_tools.py:292-358, 417-458, 256-289) in the build container; the inputs are the
reference's own test fixture `tests/fixtures/ENCFF045MJY_simple.bedpe` (a data file) and a synthetic `.bed` written by
that script.  Nothing here reads /root/reference.  The force / minimizer rows (a4-a12) are NOT pinned by this: their
arithmetic lives in OpenMM, which the image lacks."""
import json
import os

import numpy as np
import pytest

from multimm_amd import cif
from multimm_amd.ingest import import_bed, import_mns_from_bedpe
from multimm_amd.initial_structure import compute_init_struct

GOLD = os.path.join(os.path.dirname(__file__), "golden")
MANIFEST = json.load(open(os.path.join(GOLD, "ref_manifest.json")))
INGEST = np.load(os.path.join(GOLD, "ref_ingest.npz"))
CURVES = np.load(os.path.join(GOLD, "ref_curves.npz"))
BEDPE = os.path.join(GOLD, "ref_inputs", "ENCFF045MJY_simple.bedpe")
BED = os.path.join(GOLD, "ref_inputs", "synthetic_subcompartments.bed")

_RENAME = {"N_beads": "n_beads"}


def _kwargs(kw):
    return {_RENAME.get(k, k): v for k, v in kw.items()}


def _ref(kind, case, name):
    return INGEST[f"{kind}__{case}__{name}"]


@pytest.mark.parametrize("case", sorted(MANIFEST["bedpe"]))
def test_bedpe_parser_equals_the_reference(case, tmp_path):
    kw = _kwargs(MANIFEST["bedpe"][case]["kwargs"])
    n = kw.pop("n_beads")
    ms, ns, ds, ends, idxs = import_mns_from_bedpe(BEDPE, n, path=str(tmp_path), **kw)
    assert len(ms) == MANIFEST["bedpe"][case]["n_loops"]
    assert np.array_equal(ms, _ref("bedpe", case, "ms"))
    assert np.array_equal(ns, _ref("bedpe", case, "ns"))
    assert np.array_equal(np.asarray(ds, np.float64), _ref("bedpe", case, "ds"))       # bit for bit, float64
    assert np.array_equal(ends, _ref("bedpe", case, "chr_ends"))
    assert np.array_equal(idxs, _ref("bedpe", case, "chrom_idxs"))
    # the metadata files of utils.py:477,536-539
    for f in ("chrom_lengths", "chrom_idxs", "ms", "ns", "ds"):
        assert os.path.exists(tmp_path / "metadata" / (f + ".npy"))


@pytest.mark.parametrize("case", sorted(MANIFEST["bed"]))
def test_bed_parser_equals_the_reference(case, tmp_path):
    kw = _kwargs(MANIFEST["bed"][case]["kwargs"])
    n = kw.pop("n_beads")
    cs, ends, idxs = import_bed(BED, n, path=str(tmp_path), **kw)
    assert np.array_equal(cs, _ref("bed", case, "Cs").astype(np.int64))
    assert np.array_equal(ends, _ref("bed", case, "chr_ends"))
    assert np.array_equal(idxs, _ref("bed", case, "chrom_idxs"))
    hist = {int(k): v for k, v in MANIFEST["bed"][case]["histogram"].items()}
    assert {v: int((cs == v).sum()) for v in (-2, -1, 0, 1, 2)} == hist


@pytest.mark.parametrize("entry", MANIFEST["cif"], ids=lambda e: f"{e['curve']}_{e['n']}")
def test_start_curves_and_mmcif_writers_equal_the_reference(entry, tmp_path):
    curve, n, ends = entry["curve"], entry["n"], entry["chrom_ends"]
    pts = compute_init_struct(n, curve)
    assert pts.dtype == np.float64 and np.array_equal(pts, CURVES[f"{curve}_{n}"])       # the formulas, bit for bit
    out = tmp_path / "init.cif"
    cif.write_structure_angstrom(str(out), pts, np.array(ends))
    ref = open(os.path.join(GOLD, "ref_cif", entry["init"]), "rb").read()
    assert out.read_bytes() == ref                                                         # build_init_mmcif, byte for byte
    lo, hi = entry["chrom_slice"]
    outc = tmp_path / "chrom.cif"
    cif.write_chromosome_angstrom(str(outc), pts[lo:hi])
    refc = open(os.path.join(GOLD, "ref_cif", entry["chrom"]), "rb").read()
    assert outc.read_bytes() == refc                                                       # write_mmcif_chrom
    # the reader: the reference's get_coordinates_cif (utils.py:168-205; Angstrom, ATOM rows only) on its own writer's file
    # against cif.read_positions (nm; every ATOM / HETATM row: SURVEY.md appendix A.12) on the rows both keep
    path = os.path.join(GOLD, "ref_cif", entry["init"])
    is_atom = np.array([ln.startswith("ATOM") for ln in open(path) if ln.startswith(("ATOM", "HETATM"))])
    ours = cif.read_positions(path)
    assert len(ours) == n and np.array_equal(ours[is_atom], CURVES[f"read_{curve}_{n}"] * 0.1)


def test_multimm_writes_the_init_structure_in_the_files_unit(tmp_path):
    """MultiMM.initialize_simulation hands the curve to the writer in the file's own unit: the x0.1 x10 round trip that
    round 3 had flipped %.3f ties (helix, 257 beads, row 233: 465.812 in the reference, 465.813 here)."""
    from multimm_amd.config import load_config
    from multimm_amd.model import MultiMM
    cfg = load_config({"PLATFORM": "MI355X", "N_BEADS": 257, "OUT_PATH": str(tmp_path), "CHROM": "chr1",
                       "INITIAL_STRUCTURE_TYPE": "helix", "NB_CUTOFF": 0.6})
    m = MultiMM(cfg, ms=np.array([3]), ns=np.array([40]), ds=np.array([0.1]), chr_ends=np.array([0, 257]),
                Cs=np.zeros(257, np.int8))
    m.set_radiuses()
    m.initialize_simulation()
    got = open(tmp_path / "metadata" / "MultiMM_init.cif", "rb").read()
    ref = open(os.path.join(GOLD, "ref_cif", "init_helix_257.cif"), "rb").read()
    assert got == ref
    assert b" 465.812\n" in got


def test_config_defaults_equal_the_reference_schema():
    """Every ini key this engine reads has the reference's default (`config.py:94-312`, read as text by
    scripts/make_reference_config_fixture.py: name, annotation, literal default of every `Field`): quantities such as
    '300000.0 kilojoules_per_mole/nanometer**2' or '1 femtosecond' reduce to the engine's nm / kJ/mol / rad / ps floats through
    the same `parse_quantity` the ini reader uses.  Keys that differ on purpose are listed with the reason; keys of the
    reference that this path does not read (plots, nucleosomes, ATAC, force-field xml ...) are listed too, so that a new key in
    either table fails the test until someone decides where it belongs."""
    import dataclasses
    from multimm_amd.config import SimulationConfig, parse_quantity
    from multimm_amd.system import ForceFieldParams
    ref = {f["name"]: f for f in json.load(open(os.path.join(GOLD, "ref_config_defaults.json")))["fields"]}
    ours = {f.name: f.default for f in dataclasses.fields(SimulationConfig) if f.name != "ff"}
    ours.update({f.name: f.default for f in dataclasses.fields(ForceFieldParams)})
    on_purpose = {
        "PLATFORM": "the new value this engine adds (reference: CPU)",
        "DEVICE": "an int device index here, OpenMM's DeviceIndex string there",
        "INITIAL_STRUCTURE_TYPE": "an enum member there (InitialStructureType.HILBERT), its value 'hilbert' here",
        "INITIAL_STRUCTURE_PATH": "'' there, None here: both mean 'not given'",
        "GENE_TSV": "the reference ships a default table (a path inside its package); none travels here",
        "GENE_NAME": "'' there, None here", "GENE_ID": "'' there, None here",
    }
    engine_only = {"DETERMINISTIC_FORCES", "NB_CUTOFF_AUTO", "NB_CUTOFF", "MIN_TOLERANCE", "MIN_MAX_ITERATIONS"}
    not_read = {"CPU_THREADS", "FORCEFIELD_PATH", "ATACSEQ_PATH", "SAVE_PLOTS", "SC_RADIUS1", "SC_RADIUS2", "COB_DISTANCE",
                "SCB_DISTANCE", "NUC_DO_INTERPOLATION", "MAX_NUCS_PER_BEAD", "NUC_RADIUS", "POINTS_PER_NUC", "PHI_NORM",
                "SIM_ERROR_TOLERANCE", "SIM_SET_INITIAL_VELOCITIES"}
    assert set(ours) - set(ref) == engine_only
    assert set(ref) - set(ours) == not_read
    checked = 0
    for name, f in ref.items():
        if name in not_read or name in on_purpose:
            continue
        want = f["default"]
        if "OpenMMQuantity" in f["annotation"] and want is not None:
            want = parse_quantity(want)
        got = ours[name]
        if isinstance(want, float) or isinstance(got, float):
            assert float(got) == float(want), (name, got, want)
        else:
            assert got == want, (name, got, want)
        checked += 1
    assert checked >= 60
    assert ours["INITIAL_STRUCTURE_TYPE"] == "hilbert" and ref["INITIAL_STRUCTURE_TYPE"]["default"].endswith(".HILBERT")


def test_initial_structure_types_are_the_reference_enum():
    """INITIAL_STRUCTURE_TYPE takes exactly the values of the reference's `enums.py:InitialStructureType` (read as text): each
    builds a structure, anything else raises with the same list of choices (initial_structure_tools.py:256-289)."""
    kinds = json.load(open(os.path.join(GOLD, "ref_config_defaults.json")))["initial_structure_types"]
    assert len(kinds) == 9 and "hilbert" in kinds
    for k in kinds:
        pts = compute_init_struct(64, k)
        assert pts.shape == (64, 3) and np.isfinite(pts).all(), k
    with pytest.raises(ValueError) as exc:
        compute_init_struct(64, "lattice")
    assert all(k in str(exc.value) for k in kinds)


def test_functional_form_names_are_the_reference_branches():
    """Every `*_FORCE_TYPE` key accepts exactly the names the reference's force builders branch on (`mode == "<name>"` in
    model.py's add_* methods, read as text), the default being the getattr default there: `system.FORM_NAMES` lists them in
    the order of `mmx_set_functional_form`'s form index, default first."""
    from multimm_amd.system import FORM_NAMES
    ref = json.load(open(os.path.join(GOLD, "ref_config_defaults.json")))["force_types"]
    assert set(ref) == set(FORM_NAMES)
    for key, entry in ref.items():
        assert FORM_NAMES[key][0] == entry["default"], key
        assert sorted(FORM_NAMES[key]) == sorted(entry["names"]), key


@pytest.mark.parametrize("has_comp", [False, True])
def test_modelling_level_presets_equal_the_reference(has_comp):
    """MODELLING_LEVEL (run.py:128-213, `ArgumentChanger.convenient_argument_changer`, read as text): every name of every branch
    sets the same keys to the same values here -- literals, or `bool(COMPARTMENT_PATH)` where the reference says so -- starting
    from a configuration in which each of those keys holds the OPPOSITE value.  (LOC_START / LOC_END of the chromosome level are
    expressions on the chromosome-size table there; this engine leaves them unset and the ingest takes the whole chromosome.)"""
    import dataclasses
    from multimm_amd.config import load_config
    ref = json.load(open(os.path.join(GOLD, "ref_config_defaults.json")))["modelling_levels"]
    assert sorted(n for lv in ref for n in lv["names"]) == sorted(["gene", "region", "loc", "chromosome", "chrom", "gw", "genome"])
    for lv in ref:
        for name in lv["names"]:
            start = {"MODELLING_LEVEL": name, "COMPARTMENT_PATH": "comps.bed" if has_comp else None, "PLATFORM": "MI355X"}
            want = {}
            for key, val in lv["sets"].items():
                if isinstance(val, dict) or key.startswith("LOC_"):
                    continue
                want[key] = has_comp if val == "has_compartments" else val
                start[key] = (not want[key]) if isinstance(want[key], bool) else (want[key] + 7)
            cfg = load_config(start)
            got = {**{f.name: getattr(cfg, f.name) for f in dataclasses.fields(cfg) if f.name != "ff"},
                   **{f.name: getattr(cfg.ff, f.name) for f in dataclasses.fields(cfg.ff)}}
            for key, val in want.items():
                assert got[key] == val, (name, key, got[key], val)


def test_chromosome_tables_equal_the_reference():
    """`utils.chrom_lengths_array[1:]` (hg38 lengths, utils.py:67-95) and `utils.chrom_strength` (utils.py:137: the weights the
    central force multiplies by, model.py:158-162, 621) as the reference's module holds them: the table and the per-bead weights
    here are bit-equal."""
    from multimm_amd.system import CHROM_LENGTHS, chrom_strength_per_bead, gw_chr_ends
    assert np.array_equal(CHROM_LENGTHS, CURVES["chrom_lengths_array"][1:])
    ends = gw_chr_ends(20000, 22)
    w = chrom_strength_per_bead(ends, 20000)
    for i in range(22):
        assert np.all(w[ends[i]:ends[i + 1]] == CURVES["chrom_strength"][i]), i
