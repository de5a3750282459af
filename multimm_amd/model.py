"""``MultiMM`` for ``PLATFORM = MI355X``: the reference's simulation entry point with OpenMM replaced.

Mirrors the part of ``src/multimm/model.py`` that sits on the minimizer path -- same method names,
same order, same outputs:

    run() = set_radiuses -> initialize_simulation -> add_forcefield -> min_energy -> save_chromosomes

(model.py:1216-1248), followed by run_md() when SIM_RUN_MD is set (model.py:907-995).  What is NOT rebuilt:
plots, nucleosome interpolation.  Inputs come from the reference's own .bedpe/.bed
files (``multimm_amd/ingest.py``), as arrays (``ms, ns, ds, chr_ends, Cs``), or are generated synthetically
with the same tensor contracts.
"""
from __future__ import annotations

import logging
import os
import time
from typing import Optional

import numpy as np

from . import cif
from .config import NOCUTOFF_MAX_BEADS, SimulationConfig, load_config
from .engine import Engine, MMXError
from .hilbert import hilbert_points
from .system import ChromatinSystem, chrom_strength_per_bead, set_radiuses, synthetic_system

logger = logging.getLogger("multimm_amd")


class MultiMM:
    def __init__(self, args: SimulationConfig | str | dict, ms=None, ns=None, ds=None, chr_ends=None, Cs=None,
                 chrom_strength=None):
        """``args``: SimulationConfig, ini path or flat dict.  ``ms, ns, ds, chr_ends, Cs`` are the tensors
        ``MultiMM.__init__`` (model.py:105-132) obtains from import_bed / import_mns_from_bedpe; when
        omitted they are generated synthetically (seed = SHUFFLING_SEED)."""
        self.args = args if isinstance(args, SimulationConfig) else load_config(args)
        if str(self.args.PLATFORM).upper() not in ("MI355X", "GFX950"):
            raise ValueError(f"PLATFORM={self.args.PLATFORM!r}: this package only provides the MI355X platform")
        n = int(self.args.N_BEADS)
        if ms is None and str(self.args.LOOPS_PATH or "").lower().endswith(".bedpe"):
            # the reference's own inputs (model.py:105-132): compartments first, then loops, whose chr_ends
            # overwrite the ones of import_bed
            from .ingest import CHROM_SIZES, get_gene_region, import_bed, import_mns_from_bedpe
            a = self.args
            chrom = a.CHROM or None
            coords = [a.LOC_START, a.LOC_END] if (a.LOC_START is not None and a.LOC_END is not None) else None
            if chrom is not None and coords is None and chrom in CHROM_SIZES:
                coords = [0, CHROM_SIZES[chrom]]                       # whole chromosome, model.py:61-63
            if a.GENE_TSV and str(a.MODELLING_LEVEL).lower() == "gene":  # model.py:65-98
                gid = None if str(a.GENE_ID or "").lower() in ("", "none") else a.GENE_ID
                gname = None if str(a.GENE_NAME or "").lower() in ("", "none") else a.GENE_NAME
                if gid is None and gname is None:
                    raise ValueError("You did not provide gene name or ID.")
                chrom, coords, gene = get_gene_region(a.GENE_TSV, gene_id=gid, gene_name=None if gid else gname,
                                                      window_size=int(a.GENE_WINDOW))
                span = coords[1] - coords[0]
                self.gene_start, self.gene_end = ((gene[0] - coords[0]) * n) // span, ((gene[1] - coords[0]) * n) // span
                logger.info("We model the region %d-%d of chrom %s of the gene %s.", coords[0], coords[1], chrom,
                            gid or gname)
            common = dict(coords=coords, chrom=chrom, shuffle=bool(a.SHUFFLE_CHROMS), seed=int(a.SHUFFLING_SEED),
                          path=a.OUT_PATH)
            if a.COMPARTMENT_PATH and str(a.COMPARTMENT_PATH).lower().endswith(".bed"):
                Cs, chr_ends, self.chrom_idxs = import_bed(a.COMPARTMENT_PATH, n, flip_prob=float(a.COMPARTMENT_FLIP_PROB),
                                             noise_strength=float(a.COMPARTMENT_NOISE_STD), **common)
            ms, ns, ds, chr_ends, self.chrom_idxs = import_mns_from_bedpe(a.LOOPS_PATH, n, down_prob=float(a.DOWNSAMPLING_PROB),
                                                             **common)
        if ms is None:
            # no CHROM = genome-wide layout (22 chromosome intervals), as the parsers produce for chrom=None
            preset = "gw_200k" if (str(self.args.MODELLING_LEVEL).lower() in ("gw", "genome") or not self.args.CHROM) \
                else "chr1_50k"
            syn = synthetic_system(preset, seed=int(self.args.SHUFFLING_SEED), n_beads=n)
            ms, ns, ds = syn.loop_m, syn.loop_n, syn.loop_r0
            chr_ends = syn.chr_ends if chr_ends is None else chr_ends
            Cs = syn.labels if Cs is None else Cs
        self.ms, self.ns, self.ds = np.asarray(ms), np.asarray(ns), np.asarray(ds)
        if not hasattr(self, "chrom_idxs"):                 # which chromosome each interval is (model.py:107,123)
            self.chrom_idxs = np.arange(max(len(chr_ends) - 1, 1) if chr_ends is not None else 1)
        self.chr_ends = np.asarray(chr_ends if chr_ends is not None else [0, n], dtype=np.int32)
        self.Cs = np.zeros(n, np.int8) if Cs is None else np.asarray(Cs, dtype=np.int8)
        # model.py:158-162: per-bead central-force weights by chromosome INTERVAL position (utils.py:137); a single
        # region (CHROM set) keeps zeros, as the reference does
        if chrom_strength is None:
            chrom_strength = (chrom_strength_per_bead(self.chr_ends, n) if not self.args.CHROM
                              else np.zeros(n, dtype=np.float64))
        self.chrom_strength = np.asarray(chrom_strength, dtype=np.float64)
        self.save_path = os.path.join(self.args.OUT_PATH, "")
        for sub in ("metadata", "model", os.path.join("model", "chromosomes"), "md_frames"):  # model.py:44-48
            os.makedirs(os.path.join(self.args.OUT_PATH, sub), exist_ok=True)
        self.engine: Optional[Engine] = None
        self.system: Optional[ChromatinSystem] = None
        self.stats = None
        self.md_history = {"step": [], "potential": [], "kinetic": [], "total": [], "temperature": []}  # model.py:31

    # --- model.py:1016-1067 -------------------------------------------------------------------------
    def set_radiuses(self):
        self.radius1, self.radius2, self.r_comp = set_radiuses(int(self.args.N_BEADS), self.args.ff.POL_HARMONIC_BOND_R0)
        logger.info("[Radiuses] b0=%.4f nm | N=%d | R1=%.4f nm | R2=%.4f nm | r_comp=%.4f nm",
                    self.args.ff.POL_HARMONIC_BOND_R0, self.args.N_BEADS, self.radius1, self.radius2, self.r_comp)

    # --- model.py:722-810 (structure + particles; the integrator is configured in run_md) ---------------
    def initialize_simulation(self):
        n = int(self.args.N_BEADS)
        init_cif = os.path.join(self.args.OUT_PATH, "metadata", "MultiMM_init.cif")
        if self.args.BUILD_INITIAL_STRUCTURE or not self.args.INITIAL_STRUCTURE_PATH:
            from .initial_structure import compute_init_struct
            pts_angstrom = compute_init_struct(n, self.args.INITIAL_STRUCTURE_TYPE, seed=int(self.args.SHUFFLING_SEED))
            cif.write_structure_angstrom(init_cif, pts_angstrom, self.chr_ends)   # the file's unit, no nm round trip
            positions = cif.read_positions(init_cif)      # %.3f Angstrom round trip, as the reference does
        else:
            positions = cif.read_positions(self.args.INITIAL_STRUCTURE_PATH)
        if len(positions) != n:
            raise ValueError(f"structure has {len(positions)} beads, N_BEADS={n}")
        self.positions = positions
        self.mass_center = positions.mean(axis=0)
        self.system = ChromatinSystem(n_beads=n, positions=positions, chr_ends=self.chr_ends, labels=self.Cs,
                                      loop_m=self.ms, loop_n=self.ns, loop_r0=self.ds, ff=self.args.ff,
                                      chrom_strength=self.chrom_strength, seed=int(self.args.SHUFFLING_SEED))

    # --- model.py:812-857 -----------------------------------------------------------------------------
    def add_forcefield(self):
        """Installs the enabled terms in the reference's order on the device engine."""
        try:
            self.engine = Engine(self.system.n_beads, int(self.args.DEVICE))
        except MMXError as e:
            # the analogue of model.py:862-871's platform fallback, except that there is no CPU platform here
            raise MMXError(e.code, f"MI355X platform unavailable ({e}); choose another PLATFORM in the reference "
                                   "MultiMM to run on OpenMM") from e
        self.engine.set_option("deterministic", 1.0 if self.args.DETERMINISTIC_FORCES else 0.0)
        rc = float(self.system.ff.NB_CUTOFF)
        if rc > 0.0:
            # measured on one MI355X, scripts/cutoff_tolerance.py / tests/test_gpu_cutoff.py
            logger.warning("Pair terms are truncated at NB_CUTOFF = %.3g nm%s; the reference evaluates every pair (OpenMM "
                           "NoCutoff).  Measured against NoCutoff at 50 000 beads: total energy -0.4 kJ/mol per bead "
                           "(2.5e-3 of it), forces within 2.3 kJ/mol/nm per component (6e-4 relative L2; the convergence "
                           "tolerance is 10); converged structures are different local minima with the same statistics: R_g within 5 %% "
                           "(1-4 %% measured), energy within 3 %% (asserted by tests/test_gpu_cutoff.py).  "
                           "Set NB_CUTOFF = 0 in the ini for the exact all-pairs kernel.", rc,
                           " (chosen automatically above %d beads)" % NOCUTOFF_MAX_BEADS if self.args.NB_CUTOFF_AUTO else "")
        else:
            logger.info("Pair terms: every pair, no cutoff (the reference's OpenMM NoCutoff semantics)")
        self.engine.load_system(self.system)

    # --- model.py:859-897 -----------------------------------------------------------------------------
    def min_energy(self):
        logger.info("Energy minimization...")
        t0 = time.time()
        self.stats = self.engine.minimize(tolerance=float(self.args.MIN_TOLERANCE),
                                          max_iters=int(self.args.MIN_MAX_ITERATIONS))
        self.state_positions = self.engine.get_positions().astype(np.float64)
        cif.write_structure(os.path.join(self.args.OUT_PATH, "model", "MultiMM_minimized.cif"), self.state_positions,
                            self.chr_ends)
        dt = time.time() - t0
        logger.info("--- Energy minimization done!! %d iterations, %d evaluations, E %.6g -> %.6g kJ/mol, "
                    "RMS force %.3g kJ/mol/nm in %.2f s ---", self.stats.iterations, self.stats.evaluations,
                    self.stats.e_initial, self.stats.e_final, self.stats.rms_force, dt)

    # --- model.py:899-905 -----------------------------------------------------------------------------
    def save_chromosomes(self):
        for i in range(len(self.chr_ends) - 1):
            seg = self.state_positions[self.chr_ends[i]:self.chr_ends[i + 1]]
            if len(seg):
                # named after the chromosome the interval holds (model.py:904: chrs[self.chrom_idxs[i]]), which is not
                # chr{i+1} once SHUFFLE_CHROMS has permuted them
                k = int(self.chrom_idxs[i]) if i < len(self.chrom_idxs) else i
                name = "chrX" if k == 22 else "chrY" if k == 23 else f"chr{k + 1}"
                cif.write_chromosome(os.path.join(self.args.OUT_PATH, "model", "chromosomes",
                                                  f"MultiMM_minimized_{name}.cif"), seg)

    # --- model.py:907-995 -----------------------------------------------------------------------------
    def run_md(self):
        """Relaxation MD after the minimization: StateDataReporter rows every SIM_SAMPLING_STEP steps into
        md_history + md_frames/frame_<i>.cif, a DCD frame every SIM_N_STEPS // TRJ_FRAMES steps, and
        model/MultiMM_afterMD.cif at the end."""
        from .dcd import DCDWriter
        a = self.args
        n_steps, sampling = int(a.SIM_N_STEPS), max(1, int(a.SIM_SAMPLING_STEP))
        dcd_every = max(1, n_steps // max(1, int(a.TRJ_FRAMES)))
        kind = str(a.SIM_INTEGRATOR_TYPE).lower()
        eng = self.engine
        eng.md_configure(kind, dt_ps=float(a.SIM_INTEGRATOR_STEP), temperature_K=float(a.SIM_TEMPERATURE),
                         friction_per_ps=float(a.SIM_FRICTION_COEFF), seed=int(a.SHUFFLING_SEED),
                         amd_alpha=float(a.SIM_AMD_ALPHA), amd_e=float(a.SIM_AMD_E))
        # context.setVelocitiesToTemperature(SIM_TEMPERATURE, SHUFFLING_SEED), model.py:878 (the minimizer
        # leaves velocities alone, so setting them here is equivalent)
        eng.set_velocities_to_temperature(float(a.SIM_TEMPERATURE), int(a.SHUFFLING_SEED))
        logger.info("Running relaxation...")
        t0 = time.time()
        print("#\"Step\"\t\"Potential Energy (kJ/mole)\"\t\"Kinetic Energy (kJ/mole)\"\t\"Total Energy (kJ/mole)\"\t"
              "\"Temperature (K)\"")
        done = 0
        with DCDWriter(os.path.join(a.OUT_PATH, "metadata", "MultiMM_annealing.dcd"), self.system.n_beads,
                       float(a.SIM_INTEGRATOR_STEP), first_step=dcd_every, interval=dcd_every) as dcd:
            for i in range(n_steps // sampling):
                target = (i + 1) * sampling
                while done < target:  # stop at every DCD frame boundary inside the sampling window
                    nxt = min(target, (done // dcd_every + 1) * dcd_every)
                    st = eng.md_step(nxt - done)
                    done = nxt
                    if done % dcd_every == 0:
                        dcd.write_frame(eng.get_positions())
                kinetic_T = st.temperature
                print(f"{done}\t{st.potential}\t{st.kinetic}\t{st.potential + st.kinetic}\t{kinetic_T}")
                self.md_history["step"].append(int(st.step_count))
                self.md_history["potential"].append(st.potential)
                self.md_history["kinetic"].append(st.kinetic)
                self.md_history["total"].append(st.potential + st.kinetic)
                # model.py:959-972: integrator.getTemperature() (the bath set point) when the integrator has
                # one, else 2K / (3 N kB)
                self.md_history["temperature"].append(kinetic_T if kind in ("verlet", "amd")
                                                      else float(a.SIM_TEMPERATURE))
                self.state_positions = eng.get_positions().astype(np.float64)
                cif.write_structure(os.path.join(a.OUT_PATH, "md_frames", f"frame_{i + 1}.cif"),
                                    self.state_positions, self.chr_ends)
        self.state_positions = eng.get_positions().astype(np.float64)
        cif.write_structure(os.path.join(a.OUT_PATH, "model", "MultiMM_afterMD.cif"), self.state_positions,
                            self.chr_ends)
        dt = time.time() - t0
        logger.info("MD finished: %d steps in %.2f s (%.0f steps/s)", done, dt, done / max(dt, 1e-9))

    # --- model.py:1216-1248 ---------------------------------------------------------------------------
    def run(self, observer=None):
        """``observer(stage, self)``, when given, is called after the force field is installed ("forcefield") and after
        the minimization ("minimized"): what the parity tests use to look at a replica from inside the ensemble loop."""
        self.set_radiuses()
        self.initialize_simulation()
        self.add_forcefield()
        if observer is not None:
            observer("forcefield", self)
        self.min_energy()
        if observer is not None:
            observer("minimized", self)
        if self.args.LOC_START is None:
            self.save_chromosomes()
        if self.args.SIM_RUN_MD:
            self.run_md()
        self.save_parameters()
        return self.stats

    # --- utils.py:733-742 (save_args_to_txt), called at model.py:1246 -------------------------------------------
    def save_parameters(self):
        """metadata/parameters.txt: one ``NAME = value`` line per configuration key (None -> empty value)."""
        import dataclasses
        items = {k: v for k, v in dataclasses.asdict(self.args).items() if k != "ff"}
        items.update(dataclasses.asdict(self.args.ff))
        with open(os.path.join(self.args.OUT_PATH, "metadata", "parameters.txt"), "w") as f:
            for name, value in items.items():
                f.write(f"{name} = \n" if value is None else f"{name} = {value}\n")
