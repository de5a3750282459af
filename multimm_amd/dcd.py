"""DCD trajectory writer for ``metadata/MultiMM_annealing.dcd`` (``DCDReporter``, model.py:920-925).

Layout = the CHARMM/NAMD DCD that OpenMM's ``DCDFile`` emits for a non-periodic system [upstream: openmm 8.5.1
``app/dcdfile.py``; written from the published format, little endian, Fortran record markers]:

    rec 1 (84 B)  'CORD', n_frames, first_step, interval, 6x0, delta (float32, time step in AKMA units), has_box,
                  8x0, 24 (CHARMM version)
    rec 2 (164 B) 2 title lines of 80 chars
    rec 3 (4 B)   n_atoms
    per frame     three records of n_atoms float32: x, y, z in Angstrom

``n_frames`` (offset 8) and the last-step field (offset 20) are patched after every frame, as DCDFile does.
"""
from __future__ import annotations

import struct
import time

import numpy as np

AKMA_PS = 0.04888821  # one AKMA time unit in ps


class DCDWriter:
    def __init__(self, path: str, n_atoms: int, dt_ps: float, first_step: int = 0, interval: int = 1):
        self.n_atoms, self.first_step, self.interval = int(n_atoms), int(first_step), int(interval)
        self.n_frames = 0
        self._f = open(path, "w+b")
        hdr = struct.pack("<i4s9if", 84, b"CORD", 0, self.first_step, self.interval, 0, 0, 0, 0, 0, 0,
                          float(dt_ps) / AKMA_PS)
        hdr += struct.pack("<13i", 0, 0, 0, 0, 0, 0, 0, 0, 0, 24, 84, 164, 2)
        hdr += struct.pack("<80s", b"Created by multimm_amd (MI355X engine)".ljust(80))
        hdr += struct.pack("<80s", ("Created " + time.asctime()).encode().ljust(80))
        hdr += struct.pack("<4i", 164, 4, self.n_atoms, 4)
        self._f.write(hdr)

    def write_frame(self, positions_nm) -> None:
        p = np.asarray(positions_nm, dtype=np.float64).reshape(self.n_atoms, 3) * 10.0  # nm -> Angstrom
        self.n_frames += 1
        f = self._f
        f.seek(8)
        f.write(struct.pack("<i", self.n_frames))
        f.seek(20)
        f.write(struct.pack("<i", self.first_step + self.n_frames * self.interval))
        f.seek(0, 2)
        nbytes = 4 * self.n_atoms
        for k in range(3):
            f.write(struct.pack("<i", nbytes))
            f.write(np.ascontiguousarray(p[:, k], dtype="<f4").tobytes())
            f.write(struct.pack("<i", nbytes))
        f.flush()

    def close(self) -> None:
        if self._f:
            self._f.close()
            self._f = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def read_dcd(path: str) -> dict:
    """Minimal reader (tests, round trips): returns header fields and frames [F,N,3] in nm."""
    with open(path, "rb") as f:
        raw = f.read()
    if struct.unpack_from("<i4s", raw, 0) != (84, b"CORD"):
        raise ValueError("not a little-endian CORD DCD file")
    n_frames, first, interval = struct.unpack_from("<3i", raw, 8)
    last_step = struct.unpack_from("<i", raw, 20)[0]
    delta = struct.unpack_from("<f", raw, 44)[0]
    has_box = struct.unpack_from("<i", raw, 48)[0]
    assert struct.unpack_from("<i", raw, 88)[0] == 84
    off = 92
    tlen = struct.unpack_from("<i", raw, off)[0]
    off += 4 + tlen + 4
    assert struct.unpack_from("<i", raw, off)[0] == 4
    n_atoms = struct.unpack_from("<i", raw, off + 4)[0]
    off += 12
    frames = np.zeros((n_frames, n_atoms, 3))
    for fr in range(n_frames):
        if has_box:
            off += 4 + 48 + 4
        for k in range(3):
            nb = struct.unpack_from("<i", raw, off)[0]
            assert nb == 4 * n_atoms
            frames[fr, :, k] = np.frombuffer(raw, dtype="<f4", count=n_atoms, offset=off + 4)
            off += 4 + nb + 4
    assert off == len(raw), "trailing bytes"
    return dict(n_frames=n_frames, first_step=first, interval=interval, last_step=last_step, dt_ps=delta * AKMA_PS,
                n_atoms=n_atoms, frames_nm=frames * 0.1)
