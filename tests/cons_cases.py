"""Inputs and option sets shared by the consensus tests (CPU harness: test_consensus_emul.py; device: test_gpu_consensus.py)."""
import os

from mdtag import add_md_tags
from synth import write_synth_sam
from synth_rich import write_rich_sam

QCAL = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "consensus_extra", "qcal.txt")
LARGE_POS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "large_pos", "longref.sam")      # positions beyond 2^31

# every caller mode, every writer, the options that change what the device computes
OPTION_SETS = [
    ["-m", "simple"],
    ["-m", "simple", "-f", "pileup"],
    ["-m", "simple", "-f", "fastq", "-q", "-A", "-H", "0.3", "-c", "0.6", "--min-BQ", "8"],
    ["-f", "fastq"],
    ["-f", "pileup"],
    ["-f", "pileup", "-A", "-a"],
    ["-m", "bayesian_m", "-f", "pileup"],
    ["-m", "bayesian_p", "-f", "fastq", "-aa", "-T", "{fa}", "--ref-qual", "7"],
    ["-m", "bayesian_116", "-f", "pileup", "--show-del", "yes", "--show-ins", "no"],
    ["-f", "pileup", "-p"],
    ["-f", "fastq", "--homopoly-score", "0.3", "--homopoly-redux", "0.02", "--low-MQ", "5", "--scale-MQ", "1.5", "--het-scale", "0.37", "--mark-ins"],
    ["-f", "fastq", "--no-use-MQ", "-q", "--min-BQ", "10", "-d", "3"],
    ["-f", "pileup", "--no-adj-qual", "--no-adj-MQ", "--min-MQ", "20", "--ff", "0x704", "--NM-halo", "20", "--SC-cost", "30"],
    ["-f", "fasta", "-l", "60", "-C", "25", "--P-het", "0.01", "--P-indel", "0.001"],
    ["-f", "pileup", "-t", QCAL],                       # --qual-calibration file (bam_consensus.c:674-738)
    ["-m", "bayesian_m", "-f", "fastq", "-t", QCAL],
    # round 5: the machine profiles (-X / --config, bam_consensus.c:3366-3421) and the named calibration tables (:672-686).  The
    # reference holds no expected file for them: engine == oracle on the same tables (scripts/gen_qcal_tables.py)
    ["-f", "pileup", "-X", "hifi"],
    ["-f", "fastq", "-X", "hiseq"],
    ["-f", "pileup", "--config", "r10.4_sup"],
    ["-f", "fastq", "-X", "r10.4_dup"],
    ["-f", "pileup", "-X", "ultima"],
    ["-f", "pileup", "-t", ":hiseq"],
    ["-m", "bayesian_m", "-f", "fastq", "-t", ":ultima"],
]


def write_homopolymer_sam(outdir, seed=3):
    """reads over long single-base runs (130 A, 20 C, 14 G, 13 T): the homopolymer length in nm_init is capped at 100 and the
    device searches the first 12 bases either side without branches before it falls back to a loop"""
    import random
    rng = random.Random(seed)
    rnd = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    ref = rnd(200) + "A" * 130 + rnd(150) + "C" * 20 + rnd(90) + "G" * 14 + rnd(70) + "T" * 13 + rnd(200)
    fa = os.path.join(outdir, "homo.fa")
    with open(fa, "w") as fh:
        fh.write(">h1\n" + "\n".join(ref[k:k + 60] for k in range(0, len(ref), 60)) + "\n")
    sam = os.path.join(outdir, "homo.sam")
    with open(sam, "w") as fh:
        fh.write("@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:h1\tLN:%d\n" % len(ref))
        k = 0
        for pos in range(0, len(ref) - 150, 3):
            L = rng.choice((150, 150, 101, 37))
            seq = list(ref[pos:pos + L])
            for _ in range(rng.randint(0, 2)):
                j = rng.randrange(L); seq[j] = rng.choice("ACGT")
            qual = "".join(chr(33 + rng.choice((2, 11, 25, 37, 40))) for _ in range(L))
            fh.write("r%d\t%d\th1\t%d\t%d\t%dM\t*\t0\t0\t%s\t%s\n" % (k, rng.choice((0, 16)), pos + 1, rng.choice((60, 60, 30, 3)), L, "".join(seq), qual))
            k += 1
    return add_md_tags(sam, fa, os.path.join(outdir, "homo_md.sam")), fa


def make_inputs(tmpdir):
    """[(sam, fasta)]: 30x pairs with many indels (no MD), the same with MD tags on two records out of three, and the messy
    multi-contig set (clips, pads, ref skips, SEQ-less reads) with MD tags, reads over long homopolymers"""
    d = str(tmpdir)
    sam1, fa1 = write_synth_sam(d, n_ref=20000, depth=30, read_len=150, seed=5, paired=True, indel_rate=0.3, max_indel=7)
    sam1md = add_md_tags(sam1, fa1, os.path.join(d, "pairs_md.sam"), every=1)
    os.makedirs(os.path.join(d, "rich"), exist_ok=True)
    sam2, fa2 = write_rich_sam(os.path.join(d, "rich"), seed=11, n_templates=3000)
    sam2md = add_md_tags(sam2, fa2, os.path.join(d, "rich", "rich_md.sam"), every=3)
    os.makedirs(os.path.join(d, "homo"), exist_ok=True)
    sam3, fa3 = write_homopolymer_sam(os.path.join(d, "homo"))
    return [(sam1, fa1), (sam1md, fa1), (sam2md, fa2), (sam3, fa3)]
