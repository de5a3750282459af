"""mmCIF I/O in the 13-column layout MultiMM writes and reads.

Format source: ``initial_structure_tools.py:487-528`` (atom / connection loop headers) and the row
layout of ``build_init_mmcif`` (:292-358) / ``write_mmcif_chrom`` (:417-458).  Coordinates are in
Angstrom with ``%.3f`` (positions inside the engine are nm: x10 on write, x0.1 on read, as
OpenMM's PDBxFile does for the reference, model.py:753-757, 889-905).
"""
from __future__ import annotations

import numpy as np

_ATOM_COLUMNS = ("group_PDB", "id", "type_symbol", "label_atom_id", "label_alt_id", "label_comp_id",
                 "label_asym_id", "label_entity_id", "label_seq_id", "pdbx_PDB_ins_code", "Cartn_x", "Cartn_y",
                 "Cartn_z")
_CONN_COLUMNS = ("id", "conn_type_id", "ptnr1_label_comp_id", "ptnr1_label_asym_id", "ptnr1_label_seq_id",
                 "ptnr1_label_atom_id", "ptnr2_label_comp_id", "ptnr2_label_asym_id", "ptnr2_label_seq_id",
                 "ptnr2_label_atom_id")


def _atom_header() -> str:
    lines = ["data_MultiMM", "# ", "_entry.id MultiMM", "# ",
             "_audit_conform.dict_name       mmcif_pdbx.dic ",
             "_audit_conform.dict_version    5.296 ",
             "_audit_conform.dict_location   http://mmcif.pdb.org/dictionaries/ascii/mmcif_pdbx.dic ",
             "# ----------- ATOMS ----------------", "loop_"]
    lines += [f"_atom_site.{c} " for c in _ATOM_COLUMNS[:-1]] + [f"_atom_site.{_ATOM_COLUMNS[-1]}"]
    return "\n".join(lines) + "\n"


def _conn_header() -> str:
    return "#\nloop_\n" + "".join(f"_struct_conn.{c}\n" for c in _CONN_COLUMNS)


def _chain_of(i: int, chr_ends: np.ndarray) -> int:
    k = int(np.searchsorted(chr_ends, i))
    return k + 1 if i in chr_ends else k


def write_structure(path: str, positions_nm: np.ndarray, chr_ends) -> None:
    """Whole-structure file from nm positions (model/MultiMM_minimized.cif): x10 in float64, then as below."""
    write_structure_angstrom(path, np.asarray(positions_nm, dtype=np.float64) * 10.0, chr_ends)


def write_structure_angstrom(path: str, xyz_angstrom: np.ndarray, chr_ends) -> None:
    """Whole-structure file (metadata/MultiMM_init.cif): the numbers are written as they are handed over, ``%.3f`` --
    ``build_init_mmcif`` formats what ``compute_init_struct`` returns, no unit conversion in between
    (initial_structure_tools.py:292-358; a x0.1 x10 round trip flips ``%.3f`` ties).  Chain letter per chromosome,
    chromosome-boundary beads as HETATM/ALB/CB (:299-309).  Byte-identical to the reference's output for the
    deterministic curves: tests/test_reference_fixtures.py."""
    xyz = np.asarray(xyz_angstrom, dtype=np.float64)
    ce = np.asarray(chr_ends, dtype=np.int64)
    n = len(xyz)
    ends = set(int(e) for e in ce)
    ends_m1 = set(int(e) - 1 for e in ce)
    rows = []
    for i in range(n):
        chain = _chain_of(i, ce)
        if i in ends or i in ends_m1:
            grp, res, atom = "HETATM", "ALB", "CB"
        else:
            grp, res, atom = "ATOM", "ALA", "CA"
        rows.append(f"{grp} {i + 1} D {atom} . {res} {chr(65 + chain)} {chain} {i + 1} ? "
                    f"{xyz[i, 0]:.3f} {xyz[i, 1]:.3f} {xyz[i, 2]:.3f}\n")
    conns = []
    for i in range(n - 1):
        if i in ends_m1:
            continue
        chain = _chain_of(i, ce)
        r1, a1 = ("ALB", "CB") if i in ends else ("ALA", "CA")
        r2, a2 = ("ALB", "CB") if (i + 1) in ends_m1 else ("ALA", "CA")
        cl = chr(65 + chain)
        conns.append(f"D{i + 1} covale {r1} {cl} {i + 1} {a1} {r2} {cl} {i + 2} {a2}\n")
    with open(path, "w") as f:
        f.write(_atom_header() + "".join(rows) + "\n" + _conn_header() + "".join(conns))


def write_chromosome(path: str, positions_nm: np.ndarray) -> None:
    """Per-chromosome file from nm positions: ``10 * V`` in float64 as save_chromosomes does (model.py:899-905)."""
    write_chromosome_angstrom(path, np.asarray(positions_nm, dtype=np.float64) * 10.0)


def write_chromosome_angstrom(path: str, xyz_angstrom: np.ndarray) -> None:
    """Per-chromosome file (model/chromosomes/*.cif, initial_structure_tools.py:417-458)."""
    xyz = np.asarray(xyz_angstrom, dtype=np.float64)
    n = len(xyz)
    rows, conns = [], []
    for i in range(n):
        res = "ALA" if (i != 0 and i != n - 1) else "ALB"
        rows.append(f"ATOM {i + 1} D CA . {res} A 1 {i + 1} ? {xyz[i, 0]:.3f} {xyz[i, 1]:.3f} {xyz[i, 2]:.3f}\n")
    for i in range(n - 1):
        r1 = "ALA" if (i != 0 and i != n - 1) else "ALB"
        r2 = "ALA" if ((i + 1) != 0 and (i + 1) != n - 1) else "ALB"
        conns.append(f"D{i + 1} covale {r1} A {i + 1} CA {r2} A {i + 2} CA\n")
    with open(path, "w") as f:
        f.write(_atom_header() + "".join(rows) + _conn_header() + "".join(conns))


def read_positions(path: str) -> np.ndarray:
    """All ATOM/HETATM rows -> [N,3] nm (unlike the reference's own readers, which keep only ATOM rows,
    utils.py:184: SURVEY.md appendix A.12)."""
    out = []
    with open(path) as f:
        for line in f:
            if line.startswith("ATOM") or line.startswith("HETATM"):
                c = line.split()
                out.append((float(c[10]), float(c[11]), float(c[12])))
    return np.asarray(out, dtype=np.float64).reshape(-1, 3) * 0.1
