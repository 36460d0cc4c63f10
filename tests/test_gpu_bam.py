"""The BAM pipeline end to end on the GPU: process::bam::process (this build's BGZF/BAM reader, the reference's UMI
grouping, many UMI groups per device call, the gzip TSV of src/process/bam.rs:92-121) against the independent model of
tests/bam_util.py with the CPU oracle's call_umi as the aligner.  The reference's own BAM fixtures are git-LFS pointers in
this checkout: the composition is checked against the restatement only (parity unpinned)."""
import gzip
import importlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import bam_util
from oracle import oracle as ora
from test_bam_cpu import make_records

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
nim = importlib.import_module("nimble-aligner_amd")
synth = importlib.import_module("nimble-aligner_amd.synth")
HEADERS = ["reference_genome", "sequence_name", "nt_length", "sequence"]


def build(tmp_path, seed, strand="unstranded", **cfg_over):
    rng = np.random.default_rng(seed)
    names, seqs = synth.make_library(24)
    obj = synth.library_json(names, seqs)
    obj[0].update(score_percent=0.2, score_threshold=25, **cfg_over)
    path = str(tmp_path / "lib.json")
    json.dump(obj, open(path, "w"))

    last = {}

    def seq_of(L):
        # every other read is cut where the one before it was: mates of a pair then often carry the same class (a valid pair)
        if last and last["L"] == L and rng.random() < 0.6:
            s, st = last["s"], last["st"]
            last.clear()
        else:
            s = seqs[int(rng.integers(0, len(seqs)))]
            st = int(rng.integers(0, len(s) - L))
            last.update(s=s, st=st, L=L)
        r = list(s[st:st + L])
        if rng.random() < 0.3:
            r[int(rng.integers(0, L))] = "ACGT"[int(rng.integers(0, 4))]
        if rng.random() < 0.1:
            r = list("ACGT"[i] for i in rng.integers(0, 4, size=L))      # a read from nowhere
        return "".join(r)

    # (good qualities where a valid pair is wanted: mates trimmed to different lengths rarely end in one class)
    recs = make_records(rng, n_umis=150, seq_of=seq_of, good_quals=bool(cfg_over.get("require_valid_pair")))
    # copies of a pair inside one UMI group with OTHER qualities: one key, last one decides (align.rs:591-600,685)
    extra = []
    for r in recs:
        extra.append(r)
    bam = str(tmp_path / "in.bam")
    bam_util.write_bam(bam, extra, block=20000)
    cols = [["s"] * len(names), names, [str(len(s)) for s in seqs], seqs]
    ref = ora.Reference.from_columns(HEADERS, cols, "")
    cfg = ora.config_from_json(obj[0], len(names), strand)
    return path, bam, extra, ref, cfg


@pytest.mark.parametrize("force,over", [(False, {}), (True, {}), (False, {"require_valid_pair": True, "num_mismatches": 1})])
def test_bam_pipeline_equals_the_model(tmp_path, force, over, monkeypatch):
    # (a valid pair = both mates in ONE class, i.e. on one strand of the library: the strand filter must be off for it)
    strand = "none" if over else "unstranded"
    path, bam, recs, ref, cfg = build(tmp_path, 11 + int(force), strand=strand, **over)
    monkeypatch.setenv("NIMBLE_BAM_BATCH", "64")                 # several device calls, groups never split
    lib = nim.Library(path, strand).build_index()
    out = str(tmp_path / "out.tsv.gz")
    nim.bam_process(bam, [lib], [out], cores=2, force_bam_paired=force)
    text = gzip.open(out, "rb").read().decode("latin-1")
    groups = bam_util.model_groups(recs, force)
    inp = bam_util.call_inputs(groups)
    # (with -p the input often ends early: a UMI whose reads are all unpaired leaves nothing, which the reference takes
    # for the end of the file -- see tests/bam_util.py)
    assert len(inp["seg"]) > (10 if force else 100)
    exp = ora.call_umi(ora.Index.from_reference(ref), ref, cfg, inp["r1"], inp["o1"], inp["r2"], inp["o2"], q1=inp["q1"],
                       q2=inp["q2"], skip1=inp["skip1"], skip2=inp["skip2"], segment=inp["seg"], keep_per_read=True)
    n = bam_util.check_output(text, groups, inp, exp)
    assert n > (5 if force else 20 if over else 50)
    if not force:
        assert int(inp["skip1"].sum()) > 5                       # dummies of unpaired reads went through SKIP_ALIGN
    if over:
        assert "Required Valid Pair Not Matching" in text


def test_cli_takes_a_bam_file(tmp_path):
    path, bam, recs, ref, cfg = build(tmp_path, 21)
    exe = os.path.join(ROOT, "nimble-aligner_amd", "lib", "nimble")
    out = str(tmp_path / "cli.tsv.gz")
    cp = subprocess.run([exe, "-r", path, "-o", out, "-i", bam, "-c", "4", "-f", "unstranded"], capture_output=True, text=True,
                        timeout=300)
    assert cp.returncode == 0, cp.stderr
    assert "Processing as BAM file" in cp.stdout and "Validation successful" in cp.stdout
    text = gzip.open(out, "rb").read().decode("latin-1")
    groups = bam_util.model_groups(recs, False)
    inp = bam_util.call_inputs(groups)
    exp = ora.call_umi(ora.Index.from_reference(ref), ref, cfg, inp["r1"], inp["o1"], inp["r2"], inp["o2"], q1=inp["q1"],
                       q2=inp["q2"], skip1=inp["skip1"], skip2=inp["skip2"], segment=inp["seg"], keep_per_read=True)
    assert bam_util.check_output(text, groups, inp, exp) > 50


def test_large_output_is_one_gzip_member_and_batches_do_not_matter(tmp_path, monkeypatch):
    # an output of several deflate blocks (written by a pool, pigz fashion) must still be ONE valid gzip member, as
    # flate2's GzEncoder writes it, and cutting the input into device calls anywhere must not change a byte of the text
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gzip
    import subprocess
    import e2e_bam
    names, seqs = synth.make_library(64)
    lib = str(tmp_path / "lib.json")
    synth.write_library(lib, names, seqs)
    r1, r2 = synth.make_reads(seqs, 40000, paired=True, seed=5)
    bam = str(tmp_path / "in.bam")
    e2e_bam.write_bam(bam, r1, r2, 8, np.random.default_rng(3))
    exe = os.path.join(ROOT, "nimble-aligner_amd", "lib", "nimble")
    texts = []
    for batch in ("3000", "100000000"):
        out = str(tmp_path / ("out_%s.tsv.gz" % batch))
        cp = subprocess.run([exe, "-r", lib, "-o", out, "-i", bam], capture_output=True, text=True,
                            env=dict(os.environ, NIMBLE_BAM_BATCH=batch))
        assert cp.returncode == 0, cp.stderr[-500:]
        assert "Validation successful" in cp.stdout
        raw = open(out, "rb").read()
        assert raw.count(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\xff") == 1
        assert subprocess.run(["gzip", "-t", out]).returncode == 0
        texts.append(gzip.open(out).read())
    assert len(texts[0]) > (4 << 20)          # several 1 MiB blocks
    assert texts[0] == texts[1]
    assert texts[0].split(b"\n", 1)[0].count(b"\t") + 1 == 84
