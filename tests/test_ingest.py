"""Known-answer tests of the .bedpe / .bed ingest.  Every input row is written here and
every expected number is derived by hand from the rules of utils.py:425-547 / :220-347 (floor divisions,
offset shifts, np.unique ordering); no reference data file is used."""
import numpy as np
import pytest

from multimm_amd.ingest import CHROM_SIZES, import_bed, import_mns_from_bedpe


def _write(path, rows):
    path.write_text("".join("\t".join(str(c) for c in r) + "\n" for r in rows))
    return str(path)


def test_bedpe_region_mode(tmp_path):
    # region chr1:0-1000, 100 beads -> resolution 10
    f = _write(tmp_path / "a.bedpe", [
        ("chr1", 100, 120, "chr1", 300, 320, 8.0),     # ms=(10+12)//2=11, ns=(30+32)//2=31
        ("chr1", 100, 120, "chr1", 300, 329, 27.0),    # same (11,31): (30+32)//2 = 31 -> mean count 17.5
        ("chr1", 500, 510, "chr1", 520, 530, 64.0),    # ms=50, ns=52 : ns > ms+2 fails -> dropped
        ("chr1", 700, 720, "chr1", 900, 940, 1.0),     # ms=71, ns=92
        ("chr2", 100, 120, "chr2", 300, 320, 5.0),     # other chromosome -> dropped
        ("chr1", 0, 20, "chr1", 300, 320, 5.0),        # start == coords[0] (not >) -> dropped
    ])
    ms, ns, ds, ends, idxs = import_mns_from_bedpe(f, 100, coords=(0, 1000), chrom="chr1")
    assert ms.tolist() == [11, 71] and ns.tolist() == [31, 92]
    assert ends.tolist() == [0, 100] and idxs.tolist() == [0]
    # ds = 0.1 + 0.1*minmax(count^(-2/3)), counts (17.5, 1.0): the weakest loop gets 0.2, the strongest 0.1
    assert ds.tolist() == pytest.approx([0.1, 0.2])


def test_bedpe_equal_counts_collapse_to_one(tmp_path):
    f = _write(tmp_path / "b.bedpe", [("chr1", 100, 120, "chr1", 300, 320, 1), ("chr1", 400, 420, "chr1", 800, 820, 1)])
    ms, ns, ds, _, _ = import_mns_from_bedpe(f, 100, coords=(0, 1000), chrom="chr1")
    assert ds.tolist() == [1.0, 1.0]                      # utils.py:520 (single-cell inputs)


def test_bedpe_genome_wide_offsets_and_resolution(tmp_path):
    L1, L2 = CHROM_SIZES["chr1"], CHROM_SIZES["chr2"]
    f = _write(tmp_path / "c.bedpe", [
        ("chr2", 1000000, 1000000, "chr2", 50000000, 50000000, 3.0),
        ("chr1", 2000000, 2000000, "chr1", 9000000, 9000000, 9.0),
        ("chrX", 5000000, 5000000, "chrX", 7000000, 7000000, 2.0),   # not among the 22 shifted autosomes
    ])
    n = 1000
    ms, ns, ds, ends, idxs = import_mns_from_bedpe(f, n)
    res = (L1 + 50000000) // n                                        # max shifted end coordinate // N
    assert idxs.tolist() == list(range(22)) and len(ends) == 23 and ends[-1] == n
    assert ends[1] == L1 // res and ends[2] == (L1 + L2) // res
    exp = sorted([((L1 + 1000000) // res, (L1 + 50000000) // res), (2000000 // res, 9000000 // res),
                  (5000000 // res, 7000000 // res)])
    exp = [(m, min(q, n - 1)) for m, q in exp if min(q, n - 1) > m + 2]
    assert list(zip(ms.tolist(), ns.tolist())) == exp                  # np.unique order = sorted by (ms, ns)


def test_bedpe_downsampling_uses_the_seeded_legacy_stream(tmp_path):
    rows = [("chr1", 10 * i, 10 * i, "chr1", 10 * i + 400, 10 * i + 400, float(i + 1)) for i in range(1, 50)]
    f = _write(tmp_path / "d.bedpe", rows)
    full = import_mns_from_bedpe(f, 100, coords=(0, 1000), chrom="chr1")[0]
    a = import_mns_from_bedpe(f, 100, coords=(0, 1000), chrom="chr1", down_prob=0.5, seed=3)[0]
    b = import_mns_from_bedpe(f, 100, coords=(0, 1000), chrom="chr1", down_prob=0.5, seed=3)[0]
    keep = np.random.RandomState(3).rand(len(full)) < 0.5
    assert a.tolist() == b.tolist() == full[keep].tolist()


def test_bed_labels_and_discretisation(tmp_path):
    f = _write(tmp_path / "e.bed", [
        ("chr1", 100, 300, "A.1.1"), ("chr1", 300, 500, "A.2.2"), ("chr1", 500, 700, "B.1.1"),
        ("chr1", 700, 900, "B.2.2"), ("chr1", 900, 990, "unknown"), ("chr3", 0, 10, "A.1"),
    ])
    cs, ends, idxs = import_bed(f, 100, coords=(0, 1000), chrom="chr1")
    assert cs[10:30].tolist() == [2] * 20 and cs[30:50].tolist() == [1] * 20
    assert cs[50:70].tolist() == [-1] * 20 and cs[70:90].tolist() == [-2] * 20
    assert cs[:10].tolist() == [0] * 10 and cs[90:].tolist() == [0] * 10
    assert ends.tolist() == [0, 100]


def test_bed_genome_wide_shift(tmp_path):
    L1 = CHROM_SIZES["chr1"]
    total = sum(CHROM_SIZES[f"chr{i}"] for i in range(1, 23))
    n = 2000
    res = total // n
    f = _write(tmp_path / "g.bed", [("chr1", 0, 10 * res, "A1"), ("chr2", 0, 5 * res, "B2")])
    cs, ends, idxs = import_bed(f, n)
    assert len(ends) == 23 and ends[-1] == n and ends[1] == L1 // res
    assert cs[:10].tolist() == [2] * 10
    lo = L1 // res
    assert set(cs[lo:lo + 5].tolist()) == {-2} and cs[lo + 6] == 0


def test_ingest_feeds_the_model(tmp_path):
    """The parsed tensors plug into MultiMM exactly where the reference's parsers do (model.py:105-132)."""
    from multimm_amd.config import load_config
    from multimm_amd.model import MultiMM
    f = _write(tmp_path / "l.bedpe", [("chr1", 100 * i, 100 * i + 10, "chr1", 100 * i + 900, 100 * i + 910, float(i + 1))
                                        for i in range(1, 40)])
    b = _write(tmp_path / "c.bed", [("chr1", 100, 3000, "A1"), ("chr1", 3000, 4900, "B1")])
    ms, ns, ds, ends, _ = import_mns_from_bedpe(f, 500, coords=(0, 5000), chrom="chr1")
    cs, ends2, _ = import_bed(b, 500, coords=(0, 5000), chrom="chr1")
    cfg = load_config(dict(PLATFORM="MI355X", N_BEADS=500, OUT_PATH=str(tmp_path / "o"), LOC_START=0, LOC_END=5000,
                           COB_USE_COMPARTMENT_BLOCKS=True))
    m = MultiMM(cfg, ms=ms, ns=ns, ds=ds, chr_ends=ends, Cs=cs)
    m.set_radiuses()
    m.initialize_simulation()
    assert m.system.n_loops == len(ms) and set(np.unique(m.system.labels)) <= {-1, 0, 2}


def test_metadata_npy_files_like_the_reference(tmp_path):
    """The parsers leave metadata/*.npy behind (utils.py:274,343-344,477,536-539): plots.py:456-458 reads them."""
    b = _write(tmp_path / "a.bedpe", [("chr1", 100, 120, "chr1", 300, 320, 8.0), ("chr1", 700, 720, "chr1", 900, 940, 1.0)])
    e = _write(tmp_path / "e.bed", [("chr1", 100, 300, "A.1.1"), ("chr1", 500, 700, "B.1.1")])
    out = tmp_path / "run"
    cs, ends_b, idx_b = import_bed(e, 100, coords=(0, 1000), chrom="chr1", path=str(out))
    ms, ns, ds, ends, idxs = import_mns_from_bedpe(b, 100, coords=(0, 1000), chrom="chr1", path=str(out))
    meta = out / "metadata"
    assert np.array_equal(np.load(meta / "compartments.npy"), cs)
    assert np.array_equal(np.load(meta / "chrom_lengths.npy"), ends) and np.array_equal(np.load(meta / "chrom_idxs.npy"), idxs)
    assert np.array_equal(np.load(meta / "ms.npy"), ms) and np.array_equal(np.load(meta / "ns.npy"), ns)
    assert np.allclose(np.load(meta / "ds.npy"), ds)


def test_gene_region_lookup_and_gene_level_run(tmp_path):
    """utils.py:688-710 + model.py:65-98: MODELLING_LEVEL = gene takes chromosome and region from the gene table
    (gene id preferred over gene name), widened by GENE_WINDOW on both sides and clipped at 0; the gene's own bead
    range is kept as gene_start / gene_end."""
    from multimm_amd.config import load_config
    from multimm_amd.ingest import get_gene_region
    from multimm_amd.model import MultiMM
    tsv = tmp_path / "genes.tsv"
    tsv.write_text("gene_id\tgene_name\tchromosome\tstart\tend\n"
                   "ENSG01\tAAA\tchr2\t50000\t80000\n"
                   "ENSG02\tBBB\tchr1\t20000\t30000\n")
    assert get_gene_region(str(tsv), gene_id="ENSG02", window_size=5000) == ("chr1", [15000, 35000], [20000, 30000])
    assert get_gene_region(str(tsv), gene_name="AAA", window_size=100000) == ("chr2", [0, 180000], [50000, 80000])
    with pytest.raises(ValueError, match="Gene ID 'nope' not found"):
        get_gene_region(str(tsv), gene_id="nope")
    with pytest.raises(ValueError, match="Either"):
        get_gene_region(str(tsv))
    f = _write(tmp_path / "l.bedpe", [("chr1", 15000 + 100 * i, 15010 + 100 * i, "chr1", 15900 + 100 * i, 15910 + 100 * i,
                                         float(i + 1)) for i in range(1, 100)])
    cfg = load_config(dict(PLATFORM="MI355X", MODELLING_LEVEL="gene", GENE_TSV=str(tsv), GENE_ID="ENSG02",
                           GENE_WINDOW="5000", LOOPS_PATH=f, OUT_PATH=str(tmp_path / "o"), SHUFFLE_CHROMS="true"))
    assert cfg.N_BEADS == 1000 and cfg.SHUFFLE_CHROMS is False and cfg.GENE_WINDOW == 5000   # gene preset, run.py:137-154
    m = MultiMM(cfg)
    assert (m.gene_start, m.gene_end) == ((20000 - 15000) * 1000 // 20000, (30000 - 15000) * 1000 // 20000)
    ms, ns, _, ends, _ = import_mns_from_bedpe(f, 1000, coords=[15000, 35000], chrom="chr1")
    assert np.array_equal(m.ms, ms) and np.array_equal(m.ns, ns) and list(m.chr_ends) == list(ends)
    with pytest.raises(ValueError, match="gene name or ID"):
        MultiMM(load_config(dict(PLATFORM="MI355X", MODELLING_LEVEL="gene", GENE_TSV=str(tsv), LOOPS_PATH=f,
                                 OUT_PATH=str(tmp_path / "o2"))))


def test_ingest_options_come_from_the_config(tmp_path):
    """model.py:105-132: SHUFFLE_CHROMS, DOWNSAMPLING_PROB, COMPARTMENT_FLIP_PROB and COMPARTMENT_NOISE_STD reach the
    parsers with SHUFFLING_SEED; CHROM without LOC_START/LOC_END means the whole chromosome (model.py:61-63)."""
    from multimm_amd.config import load_config
    from multimm_amd.ingest import CHROM_SIZES
    from multimm_amd.model import MultiMM
    f = _write(tmp_path / "l.bedpe", [("chr1", 500000 * i, 500000 * i + 10, "chr1", 500000 * i + 5000000,
                                         500000 * i + 5000010, float(i + 1)) for i in range(1, 400)])
    b = _write(tmp_path / "c.bed", [("chr1", 0, 100000000, "A1"), ("chr1", 100000000, CHROM_SIZES["chr1"], "B1")])
    base = dict(PLATFORM="MI355X", N_BEADS=2000, CHROM="chr1", LOOPS_PATH=f, COMPARTMENT_PATH=b, SHUFFLING_SEED=3)
    m0 = MultiMM(load_config(dict(base, OUT_PATH=str(tmp_path / "a"))))
    ms, ns, _, _, _ = import_mns_from_bedpe(f, 2000, coords=[0, CHROM_SIZES["chr1"]], chrom="chr1", seed=3)
    assert len(ms) > 100 and np.array_equal(m0.ms, ms) and np.array_equal(m0.ns, ns)   # whole chromosome
    m1 = MultiMM(load_config(dict(base, OUT_PATH=str(tmp_path / "b"), DOWNSAMPLING_PROB="0.5",
                                  COMPARTMENT_FLIP_PROB="0.25", COMPARTMENT_NOISE_STD="0.1")))
    ms1, _, _, _, _ = import_mns_from_bedpe(f, 2000, coords=[0, CHROM_SIZES["chr1"]], chrom="chr1", seed=3, down_prob=0.5)
    cs1, _, _ = import_bed(b, 2000, coords=[0, CHROM_SIZES["chr1"]], chrom="chr1", seed=3, flip_prob=0.25,
                           noise_strength=0.1)
    assert 0 < len(m1.ms) < len(m0.ms) and np.array_equal(m1.ms, ms1)
    assert np.array_equal(m1.Cs, cs1) and not np.array_equal(m1.Cs, m0.Cs)
