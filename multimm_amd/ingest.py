"""Input tensors of the minimizer path from MultiMM's data files.

Restates the integer / RNG behaviour of ``import_mns_from_bedpe`` (utils.py:425-547) and ``import_bed``
(utils.py:220-347): the same floor divisions, the same chromosome offset shifts (only the first
``n_chroms`` = 22 autosomes are shifted: chrX/chrY rows stay where they are), the
same ``np.unique`` ordering of loops, the same legacy ``np.random.seed`` stream for shuffling /
down-sampling / label noise.  Files are parsed with the standard library (no pandas needed on the path).
"""
from __future__ import annotations

import numpy as np

import os

from .system import CHROM_LENGTHS

CHRS = {i: f"chr{i + 1}" for i in range(22)}
CHRS[22] = "chrX"
CHRS[23] = "chrY"
CHROM_SIZES = {CHRS[i]: int(CHROM_LENGTHS[i]) for i in range(24)}


def _read_tsv(path, ncols):
    rows = []
    with open(path) as f:
        for line in f:
            if not line.strip():
                continue
            c = line.rstrip("\n").split("\t")
            if len(c) < ncols:
                raise ValueError(f"{path}: expected >= {ncols} tab-separated columns, got {len(c)}")
            rows.append(c)
    return rows


def _chrom_layout(chrom, rng, shuffle, n_chroms):
    if chrom is not None:
        idx = next((k for k, v in CHRS.items() if v == chrom or v == f"chr{chrom}"), 0)
        return np.array([idx]), np.array([0, CHROM_SIZES[CHRS[idx]]], dtype=np.int64)
    chrom_idxs = np.arange(n_chroms).astype(int)
    if shuffle:
        rng.shuffle(chrom_idxs)
    ends = np.cumsum(np.insert(CHROM_LENGTHS[chrom_idxs], 0, 0)).astype(np.int64)
    return chrom_idxs, ends


def get_gene_region(gene_tsv, gene_id=None, gene_name=None, window_size=200000):
    """-> (chrom, [start - window, end + window] clipped at 0, [start, end]) of the first row of the gene table whose
    ``gene_id`` (preferred) or ``gene_name`` matches (utils.py:688-710; columns gene_id, gene_name, chromosome, start,
    end, tab separated with a header line)."""
    import csv
    with open(gene_tsv, newline="") as fh:
        rows = list(csv.DictReader(fh, delimiter="\t"))
    if gene_id is not None:
        key, val = "gene_id", gene_id
    elif gene_name is not None:
        key, val = "gene_name", gene_name
    else:
        raise ValueError("Either 'gene_id' or 'gene_name' must be provided.")
    hit = next((r for r in rows if r.get(key) == val), None)
    if hit is None:
        what = "Gene ID" if key == "gene_id" else "Gene name"
        raise ValueError(f"{what} '{val}' not found in the provided TSV file.")
    start, end = int(float(hit["start"])), int(float(hit["end"]))
    return hit["chromosome"], [max(0, int(start - window_size)), int(end + window_size)], [start, end]


def _save_meta(path, **arrays):
    """metadata/<name>.npy under the run directory, what plots.py:456-458 and viz_chroms read back."""
    if path is None:
        return
    d = os.path.join(path, "metadata")
    os.makedirs(d, exist_ok=True)
    for name, a in arrays.items():
        np.save(os.path.join(d, name + ".npy"), a)


def import_mns_from_bedpe(bedpe_file, n_beads, coords=None, chrom=None, threshold=0, min_loop_dist=2, down_prob=1.0,
                          shuffle=False, seed=0, n_chroms=22, path=None):
    """-> (ms, ns, ds, chr_ends, chrom_idxs) as MultiMM.__init__ receives them (model.py:122-132).
    ``path``: run directory; writes metadata/{chrom_lengths,chrom_idxs,ms,ns,ds}.npy (utils.py:477,536-539)."""
    rng = np.random.RandomState(seed)          # same stream as np.random.seed(seed) in the reference
    rows = _read_tsv(bedpe_file, 7)
    c0 = np.array([r[0] for r in rows])
    c3 = np.array([r[3] for r in rows])
    num = np.array([[int(float(r[1])), int(float(r[2])), int(float(r[4])), int(float(r[5]))] for r in rows],
                   dtype=np.int64).reshape(-1, 4)
    cnt = np.array([float(r[6]) for r in rows])
    chrom_idxs, chrom_ends = _chrom_layout(chrom, rng, shuffle, n_chroms)
    if chrom is not None:
        keep = (c0 == chrom) & (num[:, 0] > coords[0]) & (num[:, 1] < coords[1]) & (num[:, 2] > coords[0]) & \
               (num[:, 3] < coords[1])
        c0, c3, num, cnt = c0[keep], c3[keep], num[keep], cnt[keep]
    else:
        for count, i in enumerate(chrom_idxs):
            m0, m3 = c0 == CHRS[int(i)], c3 == CHRS[int(i)]
            num[m0, 0] += chrom_ends[count]
            num[m0, 1] += chrom_ends[count]
            num[m3, 2] += chrom_ends[count]
            num[m3, 3] += chrom_ends[count]
    if len(num) == 0:
        raise ValueError("The region of interest does not include loops.")
    resolution = int(num[:, 3].max()) // n_beads if chrom is None else (coords[1] - coords[0]) // n_beads
    chrom_ends = chrom_ends // resolution
    chrom_ends[-1] = n_beads
    if chrom is not None:
        num = num - coords[0]
    num = num // resolution
    ms_all = (num[:, 0] + num[:, 1]) // 2
    ns_all = (num[:, 2] + num[:, 3]) // 2
    # "Total Count" = mean count of the rows that share (ms, ns)
    key = ms_all.astype(np.int64) * (1 << 32) + ns_all.astype(np.int64)
    _, inv = np.unique(key, return_inverse=True)
    mean = np.bincount(inv, weights=cnt) / np.bincount(inv)
    counts = mean[inv]
    sel = counts > threshold
    mns = np.vstack((ms_all[sel], ns_all[sel]))
    cs = counts[sel]
    mns, idxs = np.unique(mns, axis=1, return_index=True)
    cs = cs[idxs]
    if cs.size == 0:
        raise ValueError("The region of interest does not include loops.")
    ms, ns = mns[0, :].copy(), mns[1, :].copy()
    ms[ms >= n_beads] = n_beads - 1
    ns[ns >= n_beads] = n_beads - 1
    far = ns > ms + min_loop_dist
    ms, ns, cs = ms[far], ns[far], cs[far]
    if len(cs) and not np.all(cs == cs[0]):
        w = 1.0 / cs ** (2.0 / 3.0)
        ds = 0.1 + 0.1 * ((w - w.min()) / (w.max() - w.min()))     # min_max_trans first (utils.py:396-397,520): bit-equal
    else:
        ds = np.ones(len(ms))
    nz = (ns - ms) != 0
    ms, ns, ds, cs = ms[nz], ns[nz], ds[nz], cs[nz]
    if down_prob < 1.0:
        keep = np.where(rng.rand(len(ms)) < down_prob)[0]
        ms, ns, ds = ms[keep], ns[keep], ds[keep]
    _save_meta(path, chrom_lengths=chrom_ends, chrom_idxs=chrom_idxs, ms=ms, ns=ns, ds=ds)
    return ms.astype(int), ns.astype(int), ds, chrom_ends.astype(int), chrom_idxs.astype(int)


def _label_value(label: str):
    if label.startswith("A.1") or label.startswith("A1"):
        return 2
    if label.startswith("A.2") or label.startswith("A2") or label.startswith("A"):
        return 1
    if label.startswith("B.2") or label.startswith("B2"):
        return -2
    if label.startswith("B.1") or label.startswith("B1") or label.startswith("B"):
        return -1
    return None


def import_bed(bed_file, n_beads, coords=None, chrom=None, shuffle=False, seed=0, n_chroms=22, flip_prob=0.0,
               noise_strength=0.0, path=None):
    """-> (Cs, chr_ends, chrom_idxs) as MultiMM.__init__ receives them (model.py:105-117).
    ``path``: run directory; writes metadata/{chrom_lengths,compartments,chrom_idxs}.npy (utils.py:274,343-344)."""
    rng = np.random.RandomState(seed)
    rows = _read_tsv(bed_file, 4)
    c0 = np.array([r[0] for r in rows])
    se = np.array([[int(float(r[1])), int(float(r[2]))] for r in rows], dtype=np.int64).reshape(-1, 2)
    lab = [r[3] for r in rows]
    chrom_idxs, chrom_ends = _chrom_layout(chrom, rng, shuffle, n_chroms)
    if chrom is not None:
        keep = (c0 == chrom) & (se[:, 0] > coords[0]) & (se[:, 1] < coords[1])
        c0, se, lab = c0[keep], se[keep], [l for l, k in zip(lab, keep) if k]
    else:
        for count, i in enumerate(chrom_idxs):
            m = c0 == CHRS[int(i)]
            se[m] += chrom_ends[count]
    resolution = int(chrom_ends[-1]) // n_beads if chrom is None else (coords[1] - coords[0]) // n_beads
    chrom_ends = chrom_ends // resolution
    chrom_ends[-1] = n_beads
    if chrom is not None:
        se = se - coords[0]
    se = se // resolution
    comps = np.zeros(n_beads, dtype=float)
    for (s, e), l in zip(se, lab):
        v = _label_value(l)
        if v is not None:
            comps[s:e] = v
    if noise_strength > 0:
        noise = rng.normal(0.0, noise_strength, size=n_beads)
        try:
            from scipy.ndimage import gaussian_filter1d
            noise = gaussian_filter1d(noise, sigma=8)
        except ImportError:
            pass
        comps = comps + noise
    if flip_prob > 0:
        mask = rng.rand(n_beads) < flip_prob
        mask &= comps != 0
        step = rng.choice([-1, 1], size=n_beads)
        comps[mask] += step[mask]
        comps = np.clip(comps, -2, 2)
    cs = np.where(comps > 1.5, 2, np.where(comps > 0.2, 1, np.where(comps < -1.5, -2, np.where(comps < -0.2, -1, 0))))
    _save_meta(path, chrom_lengths=chrom_ends, compartments=cs.astype(int), chrom_idxs=chrom_idxs)
    return cs.astype(int), chrom_ends.astype(int), chrom_idxs.astype(int)
