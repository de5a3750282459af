"""Host-side logic: ini config + presets, mmCIF layout, synthetic tensor contracts, MultiMM plumbing."""
import os

import numpy as np
import pytest

from multimm_amd import synthetic_system
from multimm_amd import cif
from multimm_amd.config import load_config, parse_bool, parse_quantity


def test_quantity_and_bool_parsing():
    assert parse_quantity("0.1 nanometer") == pytest.approx(0.1)
    assert parse_quantity("300000.0 kilojoules_per_mole/nanometer**2") == 300000.0
    assert parse_quantity("3.141592653589793 radian") == pytest.approx(np.pi)
    assert parse_quantity("1 angstrom") == pytest.approx(0.1)
    assert parse_quantity(2) == 2.0
    assert parse_bool("True") and not parse_bool("false") and not parse_bool("")
    with pytest.raises(ValueError):
        parse_quantity("nanometer")


def test_gw_preset_matches_reference_argument_changer(tmp_path):
    ini = tmp_path / "c.ini"
    ini.write_text("[Main]\nplatform = MI355X\nmodelling_level = GW\ncompartment_path = comps.bed\n"
                   "n_beads = 123\nsc_use_spherical_container = False\ncf_use_central_force = True\n")
    c = load_config(str(ini))
    # run.py:202-212: the preset overrides the file
    assert c.N_BEADS == 200000 and c.ff.SC_USE_SPHERICAL_CONTAINER and c.ff.COB_USE_COMPARTMENT_BLOCKS
    assert c.ff.IBL_USE_B_LAMINA_INTERACTION and not c.ff.CF_USE_CENTRAL_FORCE and not c.SIM_RUN_MD
    c2 = load_config(dict(MODELLING_LEVEL="region", N_BEADS=77))
    assert c2.N_BEADS == 5000 and not c2.ff.COB_USE_COMPARTMENT_BLOCKS and not c2.ff.SC_USE_SPHERICAL_CONTAINER
    assert load_config(dict(CHB_USE_CHROMOSOMAL_BLOCKS=True, CHB_KC=0.3)).ff.CHB_USE_CHROMOSOMAL_BLOCKS
    with pytest.raises(ValueError):
        load_config(dict(MODELLING_LEVEL="nonsense"))


def test_region_ini_like_reference_example(tmp_path):
    ini = tmp_path / "r.ini"
    ini.write_text("[Main]\nPLATFORM = MI355X\nINITIAL_STRUCTURE_TYPE = circle\nN_BEADS = 500\n"
                   "SC_USE_SPHERICAL_CONTAINER = False\nEV_POWER = 3.0\nLE_HARMONIC_BOND_K = 30000.0 kilojoules_per_mole/nanometer**2\n"
                   "SAVE_PLOTS = True\nNUC_DO_INTERPOLATION = True\n")
    c = load_config(str(ini))
    assert c.N_BEADS == 500 and c.INITIAL_STRUCTURE_TYPE == "circle" and c.ff.EV_POWER == 3.0


def test_mmcif_round_trip_and_layout(tmp_path):
    rng = np.random.default_rng(0)
    x = rng.normal(size=(50, 3))
    p = tmp_path / "s.cif"
    cif.write_structure(str(p), x, [0, 20, 50])
    y = cif.read_positions(str(p))
    assert y.shape == (50, 3) and np.abs(x - y).max() <= 5.1e-5        # %.3f Angstrom
    rows = [l.split() for l in p.read_text().splitlines() if l.startswith(("ATOM", "HETATM"))]
    assert all(len(r) == 13 for r in rows)
    assert rows[0][0] == "HETATM" and rows[0][5] == "ALB" and rows[5][0] == "ATOM" and rows[5][5] == "ALA"
    assert rows[19][0] == "HETATM" and rows[20][0] == "HETATM" and rows[20][6] != rows[18][6]   # new chain letter
    pc = tmp_path / "c.cif"
    cif.write_chromosome(str(pc), x[:20])
    assert np.abs(cif.read_positions(str(pc)) - x[:20]).max() <= 5.1e-5


def test_synthetic_contracts():
    s = synthetic_system("gw_200k", n_beads=20000, seed=3)
    assert len(s.chr_ends) == 23 and s.chr_ends[-1] == 20000
    assert np.all(s.loop_n > s.loop_m + 2)                                     # utils.py:515-519
    assert len(set(zip(s.loop_m.tolist(), s.loop_n.tolist()))) == s.n_loops   # unique pairs, utils.py:507
    assert s.loop_r0.min() >= 0.1 and s.loop_r0.max() <= 0.2
    k = np.searchsorted(s.chr_ends, s.loop_m, side="right")
    assert np.all(s.loop_n < s.chr_ends[k])                                    # loops stay inside a chromosome
    assert set(np.unique(s.labels)) <= {-2, -1, 0, 1, 2}
    assert np.allclose(np.linalg.norm(np.diff(s.positions, axis=0), axis=1), 0.1)
    s2 = synthetic_system("gw_200k", n_beads=20000, seed=4)
    assert not np.array_equal(s.labels, s2.labels)                             # ensemble seeds differ
    c = synthetic_system("region_5k", n_beads=500, start="circle")
    assert np.allclose(np.hypot(c.positions[:, 0], c.positions[:, 1]), 0.5)    # radius 5 Angstrom


def test_multimm_plumbing_until_the_engine(tmp_path):
    """run() order of model.py:1216-1248; without a GPU add_forcefield must raise (no CPU platform here)."""
    import torch
    from multimm_amd.engine import MMXError
    from multimm_amd.model import MultiMM
    cfg = load_config(dict(PLATFORM="MI355X", N_BEADS=300, OUT_PATH=str(tmp_path / "out"), NB_CUTOFF=0.6))
    m = MultiMM(cfg)
    m.set_radiuses()
    assert m.radius2 == pytest.approx(0.1 * 300 ** (1 / 3)) and m.r_comp == pytest.approx(0.15)
    m.initialize_simulation()
    assert os.path.exists(tmp_path / "out" / "metadata" / "MultiMM_init.cif")
    assert m.system.n_beads == 300 and np.allclose(m.mass_center, m.system.centre)
    if not torch.cuda.is_available():
        with pytest.raises(MMXError):
            m.add_forcefield()
    with pytest.raises(ValueError):
        MultiMM(load_config(dict(PLATFORM="CUDA", N_BEADS=10, OUT_PATH=str(tmp_path / "o2"))))


def test_md_config_keys_and_presets():
    """SIM_* keys of config.py:252-267 (Quantity strings reduce to ps / K), REGION preset turns MD on (run.py:171)."""
    from multimm_amd.config import load_config
    c = load_config({"MODELLING_LEVEL": "region", "SIM_INTEGRATOR_STEP": "2 femtosecond", "SIM_TEMPERATURE": "300 kelvin",
                     "SIM_N_STEPS": "500", "SIM_FRICTION_COEFF": "0.1", "TRJ_FRAMES": "50",
                     "SIM_INTEGRATOR_TYPE": "brownian"})
    assert c.SIM_RUN_MD and abs(c.SIM_INTEGRATOR_STEP - 0.002) < 1e-15 and c.SIM_TEMPERATURE == 300.0
    # (the preset overwrites the step count, whatever the ini says: run.py:172 -- pinned by tests/test_reference_fixtures.py)
    assert (c.SIM_N_STEPS, c.TRJ_FRAMES, c.SIM_FRICTION_COEFF, c.SIM_INTEGRATOR_TYPE) == (10000, 50, 0.1, "brownian")
    assert load_config({"SIM_N_STEPS": "500"}).SIM_N_STEPS == 500
    g = load_config({"GENERATE_ENSEMBLE": "True", "N_ENSEMBLE": "4"})
    assert g.GENERATE_ENSEMBLE is True and g.N_ENSEMBLE == 4 and not load_config({}).GENERATE_ENSEMBLE
    e = load_config({"SIM_INTEGRATOR_TYPE": "amd", "SIM_AMD_ALPHA": "250", "SIM_AMD_E": "5e3"})
    assert (e.SIM_INTEGRATOR_TYPE, e.SIM_AMD_ALPHA, e.SIM_AMD_E) == ("amd", 250.0, 5000.0)
    d = load_config({"MODELLING_LEVEL": "gw"})
    assert not d.SIM_RUN_MD
    assert (d.SIM_AMD_ALPHA, d.SIM_AMD_E) == (100.0, 1000.0)   # config.py:255-256
    assert (d.SIM_N_STEPS, d.SIM_SAMPLING_STEP, d.SIM_INTEGRATOR_TYPE, d.SIM_INTEGRATOR_STEP, d.SIM_FRICTION_COEFF,
            d.SIM_TEMPERATURE, d.TRJ_FRAMES) == (10000, 100, "langevin", 0.001, 0.5, 310.0, 2000)


def test_dcd_layout_and_round_trip(tmp_path):
    """CHARMM DCD as OpenMM's DCDReporter writes it for a non-periodic system: 84-byte CORD record with the
    frame count at offset 8 and the last step at offset 20 (patched per frame), two 80-char titles, atom count,
    then x / y / z float32 records in Angstrom."""
    import struct
    from multimm_amd.dcd import DCDWriter, read_dcd
    p = str(tmp_path / "t.dcd")
    x = np.random.default_rng(0).random((3, 7, 3)) * 5.0
    with DCDWriter(p, 7, 0.001, first_step=5, interval=5) as w:
        for f in x:
            w.write_frame(f)
    raw = open(p, "rb").read()
    assert len(raw) == 92 + 172 + 12 + 3 * 3 * (8 + 4 * 7) == 276 + 324
    assert struct.unpack_from("<i4s3i", raw, 0) == (84, b"CORD", 3, 5, 5)
    assert struct.unpack_from("<i", raw, 20)[0] == 20
    assert struct.unpack_from("<3i", raw, 84) == (24, 84, 164) and struct.unpack_from("<i", raw, 96)[0] == 2
    assert struct.unpack_from("<3i", raw, 264) == (4, 7, 4)
    first_x = np.frombuffer(raw, "<f4", 7, 280)
    assert np.allclose(first_x, x[0][:, 0] * 10.0, rtol=1e-6)
    r = read_dcd(p)
    assert (r["n_frames"], r["n_atoms"], r["first_step"], r["interval"], r["last_step"]) == (3, 7, 5, 5, 20)
    assert abs(r["dt_ps"] - 0.001) < 1e-9 and np.abs(r["frames_nm"] - x).max() < 1e-6


def test_initial_structure_types():
    """INITIAL_STRUCTURE_TYPE (config.py:138-141): the nine generators of initial_structure_tools.py:169-289, 614-640.
    Hand-checkable values for the curves, defining properties for the random ones."""
    from multimm_amd.initial_structure import MODES, compute_init_struct
    assert set(MODES) == {"rw", "confined_rw", "knot", "self_avoiding_rw", "circle", "helix", "spiral", "sphere", "hilbert"}
    n = 101
    h = compute_init_struct(n, "helix")
    assert np.allclose(h[0], [1, 0, 0]) and np.allclose(h[-1], [1, 0, 2 * n]) and np.allclose(h[25], [-1, 0, 50.5], atol=1e-12)
    sp = compute_init_struct(n, "spiral")
    assert np.allclose(sp[-1], [1 + 0.05 * 100, 0, n]) and np.allclose(np.hypot(sp[:, 0], sp[:, 1]), 1 + 0.05 * np.arange(n))
    k = compute_init_struct(n, "knot")
    assert np.allclose(k[0], [0, -5, 0]) and np.allclose(k[-1], [0, -5, 0], atol=1e-12)   # closed curve
    assert np.allclose(k[25], [5 * (1 + 0), 5 * (0 + 2), 5.0], atol=1e-12)                # t = pi/2
    c = compute_init_struct(4, "circle")
    assert np.allclose(c, [[5, 0, 12.5], [0, 5, 25], [-5, 0, 37.5], [0, -5, 50]], atol=1e-12)
    assert np.abs(np.diff(compute_init_struct(64, "hilbert"), axis=0)).sum(axis=1).tolist() == [1] * 63   # lattice walk
    rw = compute_init_struct(500, "rw", seed=3)
    assert np.allclose(np.linalg.norm(np.diff(rw, axis=0), axis=1), 1.0) and np.all(rw[0] == 0)
    assert np.array_equal(rw, compute_init_struct(500, "rw", seed=3)) and not np.array_equal(rw, compute_init_struct(500, "rw", seed=4))
    cw = compute_init_struct(2000, "confined_rw", seed=1)
    d = np.abs(np.diff(cw, axis=0))
    assert np.abs(cw).max() == 5.0 and set(np.unique(d)) <= {0.0, 1.0} and (d == 0).any()   # sticks to the walls
    saw = compute_init_struct(120, "self_avoiding_rw", seed=2)
    dist = np.linalg.norm(saw[:, None] - saw[None], axis=2) + 10 * np.eye(120)
    assert np.allclose(np.linalg.norm(np.diff(saw, axis=0), axis=1), 1.0) and dist.min() >= 1.0 - 1e-3 - 1e-12
    ball = compute_init_struct(20000, "sphere", seed=0)
    r = np.linalg.norm(ball, axis=1)
    assert r.max() <= 1.0 and abs(r.mean() - 0.75) < 0.01 and np.abs(ball.mean(axis=0)).max() < 0.02  # uniform in the ball
    with pytest.raises(ValueError, match="Invalid option for initial structure: 'zigzag'"):
        compute_init_struct(10, "zigzag")


def test_ini_without_nb_cutoff_follows_the_reference(tmp_path):
    """An ini written for the reference (no NB_CUTOFF key) runs NoCutoff up to 50 000 beads; above, the truncation is
    chosen and logged."""
    from multimm_amd.config import load_config
    assert load_config(dict(PLATFORM="MI355X", N_BEADS=5000)).ff.NB_CUTOFF == 0.0
    assert load_config(dict(PLATFORM="MI355X", N_BEADS=50000)).ff.NB_CUTOFF == 0.0
    big = load_config(dict(PLATFORM="MI355X", N_BEADS=200000))
    assert big.ff.NB_CUTOFF == 0.6 and big.NB_CUTOFF_AUTO
    assert load_config(dict(PLATFORM="MI355X", N_BEADS=200000, NB_CUTOFF=0)).ff.NB_CUTOFF == 0.0
    assert load_config(dict(PLATFORM="MI355X", MODELLING_LEVEL="GW")).ff.NB_CUTOFF == 0.6
